"""ctypes loader for libclsr_hip.so (the C ABI declared in include/clsr_hip.h).

The prototypes are parsed from the header itself so the Python bindings can never drift from
the C declarations.  There is NO fallback: if the shared library is missing the product path
raises (the CPU oracle under oracle/ is test infrastructure and is never used here).
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "clsr_hip.h")
LIB_PATH = os.environ.get("CLSR_LIB") or os.path.join(HERE, "libclsr_hip.so")   # CLSR_LIB: A/B builds of the kernels

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
}


def parse_header(path=HEADER):
    """Return {function name: (restype, [argtypes])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(clsr_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, kinds = [], []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
                base = a.replace("const", "").split("*")[0].strip()
                kinds.append({"float": "float32", "int": "int32", "double": "float64",
                              "unsigned char": "uint8"}.get(base, "void"))
            else:
                base = a.replace("const", "").replace("unsigned", "").split()
                argtypes.append(_CT[base[0]])
                kinds.append(None)
        protos[name] = (_CT[ret], argtypes, kinds)
    return protos


class ClsrLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the ctypes library with typed prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be the first HIP runtime mapped into the process
    # (same SONAME as /opt/rocm's) so that this library and torch share ONE runtime / device context.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise ClsrLibraryError(
            "%s not found: build the gfx950 kernels first (python -m clsr_amd.build). "
            "There is no CPU fallback for the CLSR step." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, argtypes, kinds) in parse_header().items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = ret
        fn.argtypes = argtypes
        fn.ptr_kinds = kinds
    _lib = lib
    return lib


def last_error():
    lib = load()
    buf = ctypes.create_string_buffer(512)
    lib.clsr_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc, name=""):
    if rc != 0:
        raise ClsrLibraryError("%s failed (rc=%d): %s" % (name, rc, last_error()))

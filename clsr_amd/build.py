"""Build clsr_amd/libclsr_hip.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m clsr_amd.build [--force]

The shared library has no torch / python dependency; the host loads it with ctypes
(clsr_amd/_lib.py).  Objects go to build/ (git-ignored); the .so is git-ignored but travels
to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC_DIR = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(HERE, "libclsr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wno-unused-result", "-I" + os.path.join(ROOT, "include")]


def _sources():
    return sorted(f for f in os.listdir(SRC_DIR) if f.endswith((".hip", ".cpp")))


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR) if f.endswith(".h")]
    hdr_dir = os.path.join(ROOT, "include")
    if os.path.isdir(hdr_dir):
        headers += [os.path.join(hdr_dir, f) for f in os.listdir(hdr_dir) if f.endswith(".h")]
    objs, procs = [], []
    for src in _sources():
        s = os.path.join(SRC_DIR, src)
        o = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
            if verbose:
                print("[clsr_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode())
        if p.returncode != 0:
            failed = True
            print("[clsr_amd.build] FAILED:", src)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[clsr_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

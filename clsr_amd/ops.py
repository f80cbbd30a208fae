"""Thin torch-tensor front end of the C ABI (include/clsr_hip.h).

``call("clsr_pgemm", X, ldx, ...)`` converts torch tensors to raw device pointers, checks
their dtype against the C prototype, appends nothing: the trailing ``stream`` argument of every
kernel entry point is filled with torch's current HIP stream.  PyTorch is used only as the
owner of device memory and streams.
"""
import ctypes

import torch

from clsr_amd import _lib


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args, stream=None):
    """Invoke an ``int clsr_*(..., void* stream)`` entry point; raises on a non-zero return."""
    lib = _lib.load()
    fn = getattr(lib, name)
    kinds = fn.ptr_kinds
    if len(args) != len(kinds) - 1:
        raise TypeError("%s expects %d arguments (+stream), got %d" % (name, len(kinds) - 1, len(args)))
    conv = []
    for i, (a, kind) in enumerate(zip(args, kinds)):
        if kind is not None:
            if a is None:
                conv.append(None)
            elif isinstance(a, torch.Tensor):
                if not a.is_cuda:
                    raise TypeError("%s arg %d: tensor must live on the GPU" % (name, i))
                if kind != "void" and str(a.dtype) != "torch." + kind:
                    raise TypeError("%s arg %d: expected %s tensor, got %s" % (name, i, kind, a.dtype))
                conv.append(a.data_ptr())
            elif isinstance(a, int):
                conv.append(a)
            else:
                raise TypeError("%s arg %d: expected tensor/None/address, got %r" % (name, i, type(a)))
        else:
            conv.append(a)
    conv.append(stream if stream is not None else stream_ptr())
    rc = fn(*conv)
    _lib.check(rc, name)


def query(name, *args):
    """Invoke a pure host-side size query (``clsr_*_parts`` / ``*_workspace_floats``)."""
    return getattr(_lib.load(), name)(*args)


def graph_begin(stream=None):
    lib = _lib.load()
    _lib.check(lib.clsr_graph_begin(stream if stream is not None else stream_ptr()), "clsr_graph_begin")


def graph_end(stream=None):
    lib = _lib.load()
    out = ctypes.c_void_p()
    _lib.check(lib.clsr_graph_end(stream if stream is not None else stream_ptr(), ctypes.byref(out)),
               "clsr_graph_end")
    return out.value


def graph_launch(graph_exec, stream=None):
    lib = _lib.load()
    _lib.check(lib.clsr_graph_launch(graph_exec, stream if stream is not None else stream_ptr()),
               "clsr_graph_launch")


def graph_destroy(graph_exec):
    _lib.check(_lib.load().clsr_graph_destroy(graph_exec), "clsr_graph_destroy")


def kp_for(K):
    """Row stride of a packed transposed weight for an input width K (see clsr_pack_weight)."""
    return 16 * ((K + 15) // 16) + 4


def pack_weight(W, out_features, in_features, transposed=False, W2=None, s1=1.0, s2=1.0, ld=None,
                ld2=None, out=None):
    """Pack a [in, out] weight block (or its transpose) into the MFMA A-operand layout
    Wt[16*ceil(out/16)][Kp], zero padded.  ``W``/``W2`` may be views into larger matrices
    (pass their row stride ``ld``)."""
    Kp = kp_for(in_features)
    opad = 16 * ((out_features + 15) // 16)
    if out is None:
        out = torch.empty(opad * Kp, dtype=torch.float32, device=W.device)
    ld = W.stride(0) if ld is None else ld
    ld2 = (W2.stride(0) if W2 is not None else 0) if ld2 is None else ld2
    call("clsr_pack_weight", W, ld, float(s1), W2, ld2, float(s2), 1 if transposed else 0,
         out_features, in_features, Kp, out)
    return out, Kp

"""Thin torch-tensor front end of the C ABI (include/clsr_hip.h).

``call("clsr_pgemm", X, ldx, ...)`` converts torch tensors to raw device pointers, checks
their dtype against the C prototype, appends nothing: the trailing ``stream`` argument of every
kernel entry point is filled with torch's current HIP stream.  PyTorch is used only as the
owner of device memory and streams.
"""
import ctypes

import torch

from clsr_amd import _lib


_scope = [None]   # raw stream of the innermost ``stream_scope`` (None: ask torch on every launch)
_scope_obj = [None]


def current_stream():
    """torch's current stream object (cached by the innermost ``stream_scope``)."""
    s = _scope_obj[0]
    return s if s is not None else torch.cuda.current_stream()


def stream_ptr():
    s = _scope[0]
    return s if s is not None else torch.cuda.current_stream().cuda_stream


class stream_scope(object):
    """``with stream_scope(stream=None):`` -- every ``call`` inside launches on ``stream`` (default: torch's current
    stream, looked up ONCE instead of per launch: ``torch.cuda.current_stream()`` costs ~8 us, a step has ~190
    launches).  Nests; code that switches torch's current stream inside a scope opens an inner scope."""

    def __init__(self, stream=None):
        self.stream = stream

    def __enter__(self):
        self.prev = (_scope[0], _scope_obj[0])
        obj = self.stream if self.stream is not None else torch.cuda.current_stream()
        _scope[0], _scope_obj[0] = obj.cuda_stream, obj
        return self

    def __exit__(self, *exc):
        _scope[0], _scope_obj[0] = self.prev
        return False


# ---- launch plans: the host side of a step whose shapes, buffers and scalars do not change is a fixed sequence of
# C-ABI calls and stream-event operations.  Recorded once (while it executes), it is replayed with the arguments
# already converted -- ~2 us per launch instead of ~10-20 us of tensor / dtype checks, descriptor packing, buffer
# look-ups and torch stream queries (clsr_amd/net.py: CLSRNet.train_step / forward).
class LaunchPlan(object):
    __slots__ = ("ops", "keep", "result")

    def __init__(self):
        self.ops, self.keep, self.result = [], [], None


_rec = [None]


def record_begin():
    _rec[0] = LaunchPlan()
    return _rec[0]


def record_end():
    p, _rec[0] = _rec[0], None
    return p


def recording():
    return _rec[0] is not None


def replay(plan):
    for fn, args in plan.ops:
        rc = fn(*args)
        if rc:                      # ctypes entry points return 0 / negative; torch stream methods return None
            _lib.check(rc, getattr(fn, "__name__", "replayed launch"))


def keep_alive(*objs):
    """Descriptor arrays whose ADDRESS is passed to a recorded call must outlive the plan."""
    if _rec[0] is not None:
        _rec[0].keep.extend(objs)


def event_record(stream):
    """A new event recorded on ``stream`` (a torch stream object)."""
    ev = torch.cuda.Event()
    ev.record(stream)
    if _rec[0] is not None:
        _rec[0].ops.append((ev.record, (stream,)))
    return ev


def stream_wait(stream, ev):
    stream.wait_event(ev)
    if _rec[0] is not None:
        _rec[0].ops.append((stream.wait_event, (ev,)))


def host_call(fn, *args):
    """Run a host-side callback now and, while a launch plan is being recorded, make it part of the plan (the
    data-parallel collectives issued from inside a step: clsr_amd/dp.py).  The callback must return None."""
    rec, _rec[0] = _rec[0], None      # launches made by the callback itself belong to the callback, not to the plan
    try:
        fn(*args)
    finally:
        _rec[0] = rec
    if rec is not None:
        rec.ops.append((fn, args))


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
    if _rec[0] is not None:
        _rec[0].ops.append((fn, tuple(conv)))


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


# hidden-to-hidden products of the recurrences (clsr_gru_desc.products): process default / fp32-input MFMA / split-bf16
RNN_PRODUCTS = {None: 0, "default": 0, "fp32": 1, "x3": 2, "x6": 3}


class GruDesc(ctypes.Structure):
    """ctypes mirror of clsr_gru_desc (include/clsr_hip.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("Pin", "Wgh", "Wch", "h0", "hT", "out_seq", "hprev", "gates",
                                               "dhT", "dout_seq", "dPin", "dh0")] + \
               [("h0_stride", ctypes.c_long), ("ldp", ctypes.c_int), ("ldg", ctypes.c_int), ("ldc", ctypes.c_int),
                ("n", ctypes.c_int), ("lddp", ctypes.c_int), ("in_div", ctypes.c_int),
                ("att", ctypes.c_void_p), ("datt", ctypes.c_void_p), ("dpin_bf16", ctypes.c_int), ("products", ctypes.c_int)] + \
               [(n, ctypes.c_void_p) for n in ("X", "Wgx", "Wcx", "bg", "bc")] + [("ldx", ctypes.c_int), ("Dx", ctypes.c_int)]


class T4Desc(ctypes.Structure):
    """ctypes mirror of clsr_t4_desc (include/clsr_hip.h)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("Pin", "Wm", "out_seq", "act", "cst", "mprev", "dout_seq", "dPin")] + \
               [("ldp", ctypes.c_int), ("ldm", ctypes.c_int), ("n", ctypes.c_int), ("lddp", ctypes.c_int),
                ("dpin_bf16", ctypes.c_int), ("products", ctypes.c_int)] + \
               [(n, ctypes.c_void_p) for n in ("st_in", "st_out", "dst_in", "dst_out")] + \
               [("act_tiled", ctypes.c_int), ("pad_", ctypes.c_int)] + \
               [(n, ctypes.c_void_p) for n in ("X", "Wkx", "bk")] + [("ldx", ctypes.c_int), ("Dx", ctypes.c_int)]


def _ptr(t):
    return None if t is None else t.data_ptr()


def gru_desc(n, Pin=None, ldp=0, Wgh=None, ldg=0, Wch=None, ldc=0, h0=None, h0_stride=0, hT=None, out_seq=None,
             hprev=None, gates=None, dhT=None, dout_seq=None, dPin=None, dh0=None, lddp=0, att=None, datt=None,
             in_div=1, products=0, X=None, ldx=0, Dx=0, Wgx=None, Wcx=None, bg=None, bc=None):
    d = GruDesc()
    d.X, d.Wgx, d.Wcx, d.bg, d.bc, d.ldx, d.Dx = _ptr(X), _ptr(Wgx), _ptr(Wcx), _ptr(bg), _ptr(bc), ldx, Dx
    d.products = RNN_PRODUCTS.get(products, products)
    for k, v in dict(Pin=Pin, Wgh=Wgh, Wch=Wch, h0=h0, hT=hT, out_seq=out_seq, hprev=hprev, gates=gates, dhT=dhT,
                     dout_seq=dout_seq, dPin=dPin, dh0=dh0, att=att, datt=datt).items():
        setattr(d, k, _ptr(v))
    d.h0_stride, d.ldp, d.ldg, d.ldc, d.n, d.lddp, d.in_div = h0_stride, ldp, ldg, ldc, n, lddp, in_div
    d.dpin_bf16 = 1 if (dPin is not None and dPin.dtype == torch.bfloat16) else 0
    return d


def t4_desc(n, Pin=None, ldp=0, Wm=None, ldm=0, out_seq=None, act=None, cst=None, mprev=None, dout_seq=None,
            dPin=None, lddp=0, st_in=None, st_out=None, dst_in=None, dst_out=None, products=0, act_tiled=False,
            X=None, ldx=0, Dx=0, Wkx=None, bk=None):
    d = T4Desc()
    d.X, d.Wkx, d.bk, d.ldx, d.Dx = _ptr(X), _ptr(Wkx), _ptr(bk), ldx, Dx
    d.products = RNN_PRODUCTS.get(products, products)
    for k, v in dict(Pin=Pin, Wm=Wm, out_seq=out_seq, act=act, cst=cst, mprev=mprev, dout_seq=dout_seq,
                     dPin=dPin, st_in=st_in, st_out=st_out, dst_in=dst_in, dst_out=dst_out).items():
        setattr(d, k, _ptr(v))
    d.ldp, d.ldm, d.n, d.lddp = ldp, ldm, n, lddp
    d.dpin_bf16 = 1 if (dPin is not None and dPin.dtype == torch.bfloat16) else 0
    d.act_tiled = 1 if act_tiled else 0
    return d


_rnn_descs_checked = False


def _check_rnn_descs():
    """The recurrence descriptors travel by address: a layout that differs from the C header would corrupt them silently."""
    global _rnn_descs_checked
    if not _rnn_descs_checked:
        assert ctypes.sizeof(GruDesc) == query("clsr_sizeof_gru_desc"), "GruDesc does not match clsr_gru_desc"
        assert ctypes.sizeof(T4Desc) == query("clsr_sizeof_t4_desc"), "T4Desc does not match clsr_t4_desc"
        _rnn_descs_checked = True


def rnn_multi(name, grus, t4, seq_len, len_stride, Hn, T, t_range=None):
    """clsr_rnn_fwd_multi / clsr_rnn_bwd_multi with python lists of descriptors; ``t_range=(t0, t1)``: the
    ``*_range`` entry points (one time range of a recurrence that runs as a chain of launches)."""
    _check_rnn_descs()
    arr = (GruDesc * max(len(grus), 1))(*grus)
    t4p = ctypes.addressof(t4) if t4 is not None else None
    keep_alive(arr, t4, grus)
    if t_range is None:
        call(name, ctypes.addressof(arr) if grus else None, len(grus), t4p, seq_len, len_stride, Hn, T)
    else:
        call(name + "_range", ctypes.addressof(arr) if grus else None, len(grus), t4p, seq_len, len_stride, Hn, T,
             int(t_range[0]), int(t_range[1]))


class PackDesc(ctypes.Structure):
    """ctypes mirror of clsr_pack_desc (include/clsr_hip.h)."""
    _fields_ = [("src1", ctypes.c_void_p), ("src2", ctypes.c_void_p), ("dst", ctypes.c_void_p),
                ("s1", ctypes.c_float), ("s2", ctypes.c_float)] + \
               [(n, ctypes.c_int) for n in ("ld1", "ld2", "transposed", "O", "I", "Kp", "o0", "i0")]


def pack_desc(src1, O, I, dst, Kp, ld1=None, transposed=False, o0=0, i0=0, src2=None, s1=1.0, s2=1.0, ld2=None):
    d = PackDesc()
    d.src1, d.src2, d.dst = src1.data_ptr(), _ptr(src2), dst.data_ptr()
    d.s1, d.s2 = float(s1), float(s2)
    d.ld1 = src1.stride(0) if ld1 is None else ld1
    d.ld2 = (src2.stride(0) if src2 is not None else 0) if ld2 is None else ld2
    d.transposed, d.O, d.I, d.Kp, d.o0, d.i0 = (1 if transposed else 0), O, I, Kp, o0, i0
    return d


def pack_table(descs, device):
    """Upload a list of PackDesc to device memory; returns (tensor, n, max_elems)."""
    import torch as _t

    arr = (PackDesc * len(descs))(*descs)
    raw = bytes(memoryview(arr))
    t = _t.frombuffer(bytearray(raw), dtype=_t.uint8).to(device)
    return t, len(descs), max(d.O * d.I for d in descs)


class DwDesc(ctypes.Structure):
    """ctypes mirror of clsr_dw_desc (include/clsr_hip.h)."""
    _fields_ = [("partial", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p),
                ("scale", ctypes.c_float)] + \
               [(n, ctypes.c_int) for n in ("nparts", "K", "N", "ldw", "accumulate")] + \
               [("partial2", ctypes.c_void_p), ("scale2", ctypes.c_float), ("nparts2", ctypes.c_int)]


def dw_table(sig, device):
    """Upload dW-reduction descriptors (tuples partial, dW, db, scale, nparts, K, N, ldw, accumulate) to the
    device; returns (tensor, n, max_outputs)."""
    import torch as _t

    assert ctypes.sizeof(DwDesc) == query("clsr_sizeof_dw_desc")
    arr = (DwDesc * len(sig))()
    for d, row in zip(arr, sig):
        partial, dW, db, scale, nparts, K, N, ldw, acc = row[:9]
        d.partial, d.dW, d.db, d.scale = partial, dW, db or None, scale
        d.nparts, d.K, d.N, d.ldw, d.accumulate = nparts, K, N, ldw, acc
        if len(row) > 9:      # + (partial2, scale2, nparts2): a second product summed into the same output
            d.partial2, d.scale2, d.nparts2 = row[9], row[10], row[11]
    t = _t.frombuffer(bytearray(bytes(memoryview(arr))), dtype=_t.uint8).to(device)
    # max_outputs / 64 = blocks per descriptor: one per quarter of a 16 x 16 tile (+ 64-wide bias slices)
    cd = lambda a, b: -(-a // b)
    return t, len(sig), 64 * max(cd(s[5], 16) * cd(s[6], 16) * 4 + (cd(s[6], 64) if s[2] else 0) for s in sig)


# ---- multi-launch descriptors (include/clsr_hip.h: clsr_mark_desc, clsr_gather_desc, clsr_rp_desc, clsr_table_desc)
_P, _L, _I, _F = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float


class MarkDesc(ctypes.Structure):
    _fields_ = [("idx", _P), ("flags", _P), ("nrows", _L), ("row_stride", _L), ("ncols", _I), ("pad_", _I)]


class ZeroDesc(ctypes.Structure):
    _fields_ = [("p", _P), ("nbytes", _L)]


class ScatterDesc(ctypes.Structure):
    _fields_ = [("src", _P), ("idx", _P), ("tbl_grad", _P), ("sumsq", _P), ("idx_stride", _L), ("ld_src", _I),
                ("col0", _I), ("N", _I), ("C", _I)]


class GatherDesc(ctypes.Structure):
    _fields_ = [("table", _P), ("idx", _P), ("out", _P), ("idx_stride", _L), ("N", _I), ("C", _I), ("ldo", _I),
                ("col0", _I)]


class RpDesc(ctypes.Structure):
    _fields_ = [("partial", _P), ("out", _P), ("scale", _F), ("nparts", _I), ("stride", _I), ("n", _I),
                ("accumulate", _I), ("pad_", _I)]


class TableDesc(ctypes.Structure):
    _fields_ = [("table", _P), ("partner", _P), ("grad", _P), ("m", _P), ("v", _P), ("flags", _P),
                ("sumsq_reg", _P), ("disc_loss", _P), ("sumsq_adam", _P), ("V", _L), ("C", _I), ("nsum", _I),
                ("sumsq_stride", _I), ("disc_scale", _F), ("disc_loss_scale", _F), ("pad_", _I)]


class SortIdsDesc(ctypes.Structure):
    """ctypes mirror of clsr_sortids_desc (include/clsr_hip.h)."""
    _fields_ = [("ids", _P), ("keys_out", _P), ("perm_out", _P), ("counts", _P), ("nrows", _L), ("row_stride", _L),
                ("ncols", _I), ("bits", _I), ("ids2", _P), ("nrows2", _L), ("row_stride2", _L)]


def sort_ids_multi(rows):
    """clsr_sort_ids_multi on a list of (ids, keys_out, perm_out, counts, nrows, row_stride, ncols, bits) tuples."""
    assert ctypes.sizeof(SortIdsDesc) == query("clsr_sizeof_sortids_desc")
    arr = (SortIdsDesc * len(rows))()
    keep_alive(arr)
    for d, row in zip(arr, rows):
        for (fname, _), val in zip(SortIdsDesc._fields_, row):
            setattr(d, fname, val)
    call("clsr_sort_ids_multi", ctypes.addressof(arr), len(rows))


def sort_ids_stable_multi(rows, workspace):
    """clsr_sort_ids_stable_multi on a list of (ids, keys_out, perm_out, nrows, row_stride, ncols, bits[, ids2, nrows2,
    row_stride2]) tuples (pointers as integers); ``workspace``: a uint8 tensor of
    clsr_sort_ids_stable_workspace_bytes(total entries, tables)."""
    assert ctypes.sizeof(SortIdsDesc) == query("clsr_sizeof_sortids_desc")
    arr = (SortIdsDesc * len(rows))()
    keep_alive(arr)
    for d, row in zip(arr, rows):
        ids, ko, po, nrows, stride, ncols, bits = row[:7]
        d.ids, d.keys_out, d.perm_out, d.counts = ids, ko, po, None
        d.nrows, d.row_stride, d.ncols, d.bits = nrows, stride, ncols, bits
        d.ids2, d.nrows2, d.row_stride2 = row[7:10] if len(row) > 7 else (None, 0, 0)
    call("clsr_sort_ids_stable_multi", ctypes.addressof(arr), len(rows), workspace, workspace.numel())


class SegsumDesc(ctypes.Structure):
    """ctypes mirror of clsr_segsum_desc (include/clsr_hip.h)."""
    _fields_ = [("src", _P), ("src2", _P), ("dmean", _P), ("drecent", _P), ("keys", _P), ("perm", _P), ("seq_len", _P),
                ("grad", _P), ("sumsq", _P), ("n", _L), ("src_bf16", _I), ("len_stride", _I), ("T", _I), ("D", _I),
                ("col0", _I), ("C", _I), ("recent_k", _I), ("ldg", _I), ("gcol0", _I), ("assign", _I),
                ("src_b", _P), ("sumsq_b", _P), ("n1", _L), ("ldb", _I), ("colb", _I), ("border_wch", _I), ("pad_", _I)]


def segsum_descs(rows):
    """ctypes array of clsr_segsum_desc from field tuples in the order of SegsumDesc._fields_ (pointers as integers)."""
    assert ctypes.sizeof(SegsumDesc) == query("clsr_sizeof_segsum_desc")
    arr = (SegsumDesc * len(rows))()
    for d, row in zip(arr, rows):        # (short tuples: the trailing fields -- assign, second source -- stay zero)
        for (fname, _), val in zip(SegsumDesc._fields_, row):
            setattr(d, fname, val)
    return arr


def segsum_workspace_bytes(rows):
    arr = segsum_descs(rows)
    return query("clsr_segsum_workspace_bytes", ctypes.addressof(arr), len(rows))


def segsum_multi(rows, workspace):
    """clsr_segsum_multi: deterministic segmented sums of the lookup sites ``rows`` into their gradient tables."""
    arr = segsum_descs(rows)
    keep_alive(arr)
    call("clsr_segsum_multi", ctypes.addressof(arr), len(rows), workspace, workspace.numel())


class BnPtrs(ctypes.Structure):
    """ctypes mirror of clsr_bn_ptrs (include/clsr_hip.h)."""
    _fields_ = [(n, _P) for n in ("gamma", "beta", "moving_mean", "moving_var", "scale", "shift", "mean", "invstd", "coef",
                                  "dgamma", "dbeta")]


class HeadsDesc(ctypes.Structure):
    """ctypes mirror of clsr_heads_desc (include/clsr_hip.h)."""
    _fields_ = ([(n, _P) for n in ("fs", "target", "att_long", "att_short", "tnow", "labels")]
                + [("tnow_stride", _L), ("tnow_col", _I), ("tnow_group", _I), ("B", _L)]
                + [(n, _I) for n in ("G", "D", "nfs", "a_in", "ld", "A0", "A1", "L0", "L1")]
                + [(n, _P) for n in ("al_w0", "al_w1", "lg_w0", "lg_w1", "al_w0T", "al_w1T", "lg_w0T", "lg_w1T")]
                + [(n, _I) for n in ("kp_al_w0", "kp_al_w1", "kp_lg_w0", "kp_lg_w1", "kp_al_w0T", "kp_al_w1T", "kp_lg_w0T",
                                     "kp_lg_w1T")]
                + [(n, _P) for n in ("al_b0", "al_b1", "al_wout", "al_bout", "lg_b0", "lg_b1", "lg_wout", "lg_bout")]
                + [("bn", BnPtrs * 4), ("momentum", _F), ("eps", _F), ("lscale", _F), ("pad_", _I)]
                + [(n, _P) for n in ("ain", "al_z0", "al_z1", "alpha", "mo", "lg_z0", "lg_z1", "logit", "dlogit", "loss",
                                     "lg_dz1", "lg_dz0", "dmo", "al_dz1", "al_dz0", "lg_wp", "al_wp", "dL", "dS", "dtarget",
                                     "dfs", "workspace")]
                + [("workspace_bytes", _L), ("comm", _P), ("abort_flag", _P)])


def heads_desc(**kw):
    """clsr_heads_desc from keyword values: tensors become their device pointers, ``bn`` is a list of four dicts."""
    assert ctypes.sizeof(HeadsDesc) == query("clsr_sizeof_heads_desc")
    d = HeadsDesc()
    ptr = lambda v: v.data_ptr() if hasattr(v, "data_ptr") else (0 if v is None else v)
    for i, layer in enumerate(kw.pop("bn")):
        for name, v in layer.items():
            setattr(d.bn[i], name, ptr(v))
    for name, v in kw.items():
        setattr(d, name, ptr(v))
    return d


def heads_fused(step, desc):
    """clsr_heads_fused_step1 / _step2 on a HeadsDesc."""
    keep_alive(desc)
    call("clsr_heads_fused_step%d" % step, ctypes.addressof(desc))


class DwJob(ctypes.Structure):
    """ctypes mirror of clsr_dwjob (include/clsr_hip.h)."""
    _fields_ = [("X", _P), ("Xmul", _P), ("in_scale", _P), ("in_shift", _P), ("dY", _P), ("workspace", _P),
                ("x_bf16", _I), ("ldx", _I), ("T", _I), ("G", _I), ("ldmul", _I), ("in_relu", _I), ("dy_bf16", _I),
                ("ldy", _I), ("M", _I), ("K", _I), ("N", _I), ("pad_", _I),
                ("rm_tc", _I), ("rm_T", _I), ("rm_t0", _I), ("pgx", _I), ("pstride", _I), ("poff", _I)]


DW_MULTI_MAX = 12


def dw_multi(name, rows, stream=None):
    """clsr_pgemm_dw_partial_multi / clsr_hdw_partial_multi on a list of DwJob field tuples (pointers as integers)."""
    assert ctypes.sizeof(DwJob) == query("clsr_sizeof_dwjob")
    for i in range(0, len(rows), DW_MULTI_MAX):
        chunk = rows[i:i + DW_MULTI_MAX]
        arr = (DwJob * len(chunk))()
        keep_alive(arr)
        for d, row in zip(arr, chunk):
            for (fname, _), val in zip(DwJob._fields_, row):
                setattr(d, fname, val)
        call(name, ctypes.addressof(arr), len(chunk), stream=stream)


_multi_checked = False


def _check_multi_sizes():
    global _multi_checked
    if _multi_checked:
        return
    sz = [ctypes.c_int() for _ in range(4)]
    _lib.load().clsr_sizeof_multi_descs(*[ctypes.byref(x) for x in sz])
    got = [ctypes.sizeof(c) for c in (MarkDesc, GatherDesc, RpDesc, TableDesc)]
    got += [ctypes.sizeof(ZeroDesc), ctypes.sizeof(ScatterDesc)]
    sz += [ctypes.c_int(query("clsr_sizeof_zero_desc")), ctypes.c_int(query("clsr_sizeof_scatter_desc"))]
    if got != [x.value for x in sz]:
        raise RuntimeError("ctypes mirrors of the multi-launch descriptors are out of date: %r vs %r"
                           % (got, [x.value for x in sz]))
    _multi_checked = True


def multi(name, cls, rows, *extra):
    """Launch ``name`` (a clsr_*_multi entry point) on a list of descriptor field tuples (chunks of 16)."""
    _check_multi_sizes()
    lim = 4 if cls is TableDesc else 16
    for i in range(0, len(rows), lim):
        chunk = rows[i:i + lim]
        arr = (cls * len(chunk))()
        keep_alive(arr)
        for d, row in zip(arr, chunk):
            for (fname, _), val in zip(cls._fields_, row):
                setattr(d, fname, val)
        call(name, ctypes.addressof(arr), len(chunk), *extra)


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

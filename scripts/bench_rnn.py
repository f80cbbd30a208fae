"""Isolated timing of the fused recurrence launches at configs[1] shapes.  usage: python scripts/bench_rnn.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from clsr_amd import ops
dev = "cuda:0"
def timeit(fn, iters=10, warm=3):
    s = torch.cuda.current_stream()
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    Hn, T, n = 4096, 50, 40
    NX = 3 * n * 2 + 6 * n
    lens = torch.full((Hn,), T, dtype=torch.int32, device=dev)
    Pin = torch.randn(Hn * T, NX, device=dev) * 0.3
    Wg, Wc, Wm = torch.randn(n, 2 * n, device=dev) * 0.2, torch.randn(n, n, device=dev) * 0.2, torch.randn(n, 4 * n, device=dev) * 0.2
    FUSED = bool(os.environ.get("RNN_FUSED"))
    X = torch.randn(Hn * T, 40, device=dev) * 0.5
    Wgf, Wcf, Wkf = torch.randn(40 + n, 2 * n, device=dev) * 0.2, torch.randn(40 + n, n, device=dev) * 0.2, torch.randn(40 + n, 4 * n, device=dev) * 0.2
    bgf, bcf, bkf = torch.zeros(2 * n, device=dev), torch.zeros(n, device=dev), torch.zeros(4 * n, device=dev)
    P3 = torch.randn(Hn * T, 3 * n, device=dev) * 0.3
    fg = dict(X=X, ldx=40, Dx=40, Wgx=Wgf, Wcx=Wcf, bg=bgf, bc=bcf) if FUSED else {}
    ft = dict(X=X, ldx=40, Dx=40, Wkx=Wkf, bk=bkf) if FUSED else {}
    def gru(off, train=True):
        return ops.gru_desc(n, Pin=None if FUSED else Pin[:, off:], ldp=NX, Wgh=Wg, ldg=2 * n, Wch=Wc, ldc=n, **fg,
                            hT=torch.zeros(Hn, n, device=dev), hprev=torch.zeros(Hn, T, n, device=dev) if train else None,
                            gates=torch.zeros(Hn, T, 3 * n, device=dev) if train else None)
    TILED = bool(os.environ.get("RNN_TILED"))
    act_t = torch.zeros(ops.query("clsr_t4_act_tiled_floats", Hn, T, n), device=dev) if TILED else None
    def t4(train=True):
        return ops.t4_desc(n, Pin=P3 if FUSED else Pin[:, 6 * n:], ldp=3 * n if FUSED else NX, **ft, Wm=Wm, ldm=4 * n, out_seq=torch.zeros(Hn, T, n, device=dev),
                           act=(act_t if TILED else torch.zeros(Hn, T, 6 * n, device=dev)) if train else None, act_tiled=TILED and train,
                           cst=torch.zeros(Hn, T, n, device=dev) if train else None,
                           mprev=torch.zeros(Hn, T, n, device=dev) if train else None)
    g1, g2, t = gru(0), gru(3 * n), t4()
    only = sys.argv[1] if len(sys.argv) > 1 else None      # "t4" | "gru" | "all" | "bwd": ONE configuration (counter runs)
    if only and only != "bwd":
        fn = {"t4": lambda: ops.rnn_multi("clsr_rnn_fwd_multi", [], t, lens, 1, Hn, T),
              "gru": lambda: ops.rnn_multi("clsr_rnn_fwd_multi", [g1], None, lens, 1, Hn, T),
              "all": lambda: ops.rnn_multi("clsr_rnn_fwd_multi", [g1, g2], t, lens, 1, Hn, T)}[only]
        print("fwd  %-4s                    : %6.1f us" % (only, timeit(fn)))
        sys.exit(0)
    if only != "bwd":
      print("fwd  t4 only (training)     : %6.1f us" % timeit(lambda: ops.rnn_multi("clsr_rnn_fwd_multi", [], t, lens, 1, Hn, T)))
    te = t4(False)
    if only != "bwd":
      print("fwd  t4 only (scoring)      : %6.1f us" % timeit(lambda: ops.rnn_multi("clsr_rnn_fwd_multi", [], te, lens, 1, Hn, T)))
    if only != "bwd":
      print("fwd  one gru (training)     : %6.1f us" % timeit(lambda: ops.rnn_multi("clsr_rnn_fwd_multi", [g1], None, lens, 1, Hn, T)))
    if only != "bwd":
      print("fwd  gru + gru + t4         : %6.1f us" % timeit(lambda: ops.rnn_multi("clsr_rnn_fwd_multi", [g1, g2], t, lens, 1, Hn, T)))
    # backward-through-time of the same three encoders (saved activations of the launch above)
    dP = torch.zeros(Hn * T, NX, device=dev)
    dseq, dh = torch.randn(Hn, T, n, device=dev) * 0.1, torch.randn(Hn, n, device=dev) * 0.1
    def gru_b(d, off):
        hp = torch.rand(Hn, T, n, device=dev); ga = torch.rand(Hn, T, 3 * n, device=dev)
        return ops.gru_desc(n, Wgh=Wg, ldg=2 * n, Wch=Wc, ldc=n, hprev=hp, gates=ga, dhT=dh, dPin=dP[:, off:], lddp=NX)
    tb = ops.t4_desc(n, Wm=Wm, ldm=4 * n, act=torch.rand_like(act_t) if TILED else torch.rand(Hn, T, 6 * n, device=dev), act_tiled=TILED, cst=torch.randn(Hn, T, n, device=dev) * 0.3,
                     dout_seq=dseq, dPin=dP[:, 6 * n:], lddp=NX)
    b1, b2 = gru_b(g1, 0), gru_b(g2, 3 * n)
    if only != "bwd":
      print("bwd  t4 only                : %6.1f us" % timeit(lambda: ops.rnn_multi("clsr_rnn_bwd_multi", [], tb, lens, 1, Hn, T)))
    if only != "bwd":
      print("bwd  one gru                : %6.1f us" % timeit(lambda: ops.rnn_multi("clsr_rnn_bwd_multi", [b1], None, lens, 1, Hn, T)))
    print("bwd  gru + gru + t4         : %6.1f us" % timeit(lambda: ops.rnn_multi("clsr_rnn_bwd_multi", [b1, b2], tb, lens, 1, Hn, T)))

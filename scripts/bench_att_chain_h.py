"""Isolated timing of the speed-mode (bf16) chain kernels at configs[1] shapes (short-term attention, M = 1 024 000
positions): clsr_att_l0_fwd_x1_h, clsr_att_l1_fwd_x1_h, clsr_att_l1_bwd_x1_h (both passes), clsr_att_l0_bwd_x1_h, with the
algorithmic bytes of each.   usage: python scripts/bench_att_chain_h.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clsr_amd import ops
from clsr_amd.ops import call, query
dev, BF = "cuda:0", torch.bfloat16
def timeit(fn, iters=20, warm=3):
    s = torch.cuda.current_stream()
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters): fn()
    e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
st_ = torch.cuda.Stream()
with torch.cuda.stream(st_):
    Hn, G, T, Q, A0, A1 = 4096, 5, 50, 40, 80, 40
    R, M = Hn * G, Hn * G * T
    a, q = torch.randn(Hn * T, Q, device=dev), torch.randn(R, Q, device=dev)
    U, V = torch.randn(Hn * T, A0, device=dev), torch.randn(R, A0, device=dev)
    Wp = torch.randn(Q, A0, device=dev) * 0.2
    Wf, Kf = ops.pack_weight(Wp, A0, Q)
    WtT, KpT = ops.pack_weight(Wp, Q, A0, transposed=True)
    W1 = torch.randn(A0, A1, device=dev) * 0.3
    W1f, K1f = ops.pack_weight(W1, A1, A0)
    W1T, K1T = ops.pack_weight(W1, A0, A1, transposed=True)
    z0, z1 = torch.randn(M, A0, device=dev).to(BF), torch.randn(M, A1, device=dev).to(BF)
    dz0 = torch.randn(M, A0, device=dev).to(BF)
    st = torch.zeros(1024, 2, A0, dtype=torch.float64, device=dev)
    ds = torch.randn(M, device=dev)
    v = lambda n: torch.rand(n, device=dev) + 0.5
    sc1, sh1, wo, c1 = v(A1), torch.randn(A1, device=dev) * 0.3, torch.randn(A1, device=dev), torch.randn(3 * A1, device=dev)
    sc0, sh0, mu0, is0, c0 = v(A0), torch.randn(A0, device=dev) * 0.3, torch.randn(A0, device=dev) * 0.1, v(A0), torch.randn(3 * A0, device=dev)
    b1 = torch.randn(A1, device=dev)
    C = query("clsr_dw_chunk_floats")
    ws = torch.zeros(1024 * C, device=dev)
    da, dq = torch.zeros(Hn * T, Q, device=dev), torch.zeros(R, Q, device=dev)
    dU, dV = torch.zeros(Hn * T, A0, device=dev), torch.zeros(R, A0, device=dev)
    hist = Hn * T * (Q + A0) * 4 / 1e6
    def line(name, mb, t):
        print("%-58s %6.1f MB  %6.1f us  %5.2f TB/s" % (name, mb, t, mb / t))
    t = timeit(lambda: call("clsr_att_l0_fwd_x1_h", a, Q, q, Q, Wf, Kf, U, A0, V, A0, z0, A0, st, Hn, G, T, Q, A0))
    line("x1_h l0 fwd: z0 (bf16) = U + V + (a*q).Wp + stats", M * A0 * 2 / 1e6 + hist, t)
    t = timeit(lambda: call("clsr_att_l1_fwd_x1_h", z0, A0, sc0, sh0, W1f, K1f, b1, z1, A1, st, M, A0, A1))
    line("x1_h l1 fwd: z1 = relu(bn z0).W1 + b1 + stats", M * (A0 + A1) * 2 / 1e6, t)
    t = timeit(lambda: call("clsr_att_l1_bwd_x1_h", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, mu0, is0, None, None, 0, None, st, M, A1, A0))
    line("x1_h l1 bwd pass 1 (stats)", M * ((A0 + A1) * 2 + 4) / 1e6, t)
    t = timeit(lambda: call("clsr_att_l1_bwd_x1_h", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, None, None, c0, dz0, A0, ws, None, M, A1, A0))
    line("x1_h l1 bwd pass 2 (dz0 + dW1, db1)", M * ((2 * A0 + A1) * 2 + 4) / 1e6, t)
    t = timeit(lambda: call("clsr_att_l0_bwd_x1_h", dz0, A0, WtT, KpT, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0, ws))
    line("x1_h l0 bwd (da, dq, dU, dV + dWp)", M * A0 * 2 / 1e6 + Hn * T * (2 * Q + A0) * 4 / 1e6, t)

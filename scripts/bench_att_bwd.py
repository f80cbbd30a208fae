"""Isolated timing of the attention-MLP backward kernels at configs[1] shapes: the fp32-MFMA kernels + their weight-gradient
launches against the split-bf16 kernels with the weight gradients folded in (csrc/attbwdx3.hip).
usage: python scripts/bench_att_bwd.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from clsr_amd import ops
from clsr_amd.ops import call, query
dev = "cuda:0"
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
    Wp = torch.randn(Q, A0, device=dev) * 0.2
    WtT, KpT = ops.pack_weight(Wp, Q, A0, transposed=True)
    z0, z1 = torch.randn(M, A0, device=dev), torch.randn(M, A1, device=dev)
    dz0, dz1 = torch.randn(M, A0, device=dev), torch.zeros(M, A1, device=dev)
    st = torch.zeros(1024, 2, A0, dtype=torch.float64, device=dev)
    W1 = torch.randn(A0, A1, device=dev) * 0.3
    W1T, K1T = ops.pack_weight(W1, A0, A1, transposed=True)
    ds = torch.randn(M, device=dev)
    v = lambda n: torch.rand(n, device=dev) + 0.5
    sc1, sh1, wo, c1 = v(A1), torch.randn(A1, device=dev) * 0.3, torch.randn(A1, device=dev), torch.randn(3 * A1, device=dev)
    sc0, sh0, mu0, is0, c0 = v(A0), torch.randn(A0, device=dev) * 0.3, torch.randn(A0, device=dev) * 0.1, v(A0), torch.randn(3 * A0, device=dev)
    C = query("clsr_dw_chunk_floats")
    ws = torch.zeros(1024 * C, device=dev)
    wsd = torch.zeros(query("clsr_pgemm_dw_workspace_floats", M, A0, A0), device=dev)
    da, dq = torch.zeros(Hn * T, Q, device=dev), torch.zeros(R, Q, device=dev)
    dU, dV = torch.zeros(Hn * T, A0, device=dev), torch.zeros(R, A0, device=dev)
    if os.environ.get("ONLY_L0X3"):
        t = timeit(lambda: call("clsr_att_l0_bwd_x3", dz0, A0, WtT, KpT, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0, ws))
        print("x3    l0 (da, dq, dU, dV + dWp)  %-10s %6.1f us" % (os.environ["ONLY_L0X3"], t))
        sys.exit(0)
    t = timeit(lambda: call("clsr_att_l1_bwd", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, mu0, is0, None, None, 0, None, 0, st, M, A1, A0))
    print("fp32  l1 pass 1 (stats)               R 492 MB            %6.1f us  %5.2f TB/s" % (t, 492 / t))
    t = timeit(lambda: call("clsr_att_l1_bwd_x3", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, mu0, is0, None, None, 0, None, st, M, A1, A0))
    print("x3    l1 pass 1 (stats)               R 492 MB            %6.1f us  %5.2f TB/s" % (t, 492 / t))
    t = timeit(lambda: call("clsr_att_l1_bwd", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, None, None, c0, dz1, A1, dz0, A0, None, M, A1, A0))
    print("fp32  l1 pass 2 (dz0, dz1)            R 492 + W 492 MB    %6.1f us  %5.2f TB/s" % (t, 984 / t))
    t2 = timeit(lambda: call("clsr_pgemm_dw_partial", z0, A0, 0, 0, None, 0, sc0, sh0, 1, dz1, A1, M, A0, A1, wsd))
    print("fp32  dW1 = relu(bn z0)^T dz1         R 492 MB            %6.1f us  %5.2f TB/s" % (t2, 492 / t2))
    t = timeit(lambda: call("clsr_att_l1_bwd_x3", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, None, None, c0, dz0, A0, ws, None, M, A1, A0))
    print("x3    l1 pass 2 (dz0 + dW1, db1)      R 492 + W 328 MB    %6.1f us  %5.2f TB/s" % (t, 820 / t))
    t = timeit(lambda: call("clsr_att_l0_bwd", dz0, A0, WtT, KpT, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0))
    print("fp32  l0 (da, dq, dU, dV)             R 328 + 33, W 98 MB %6.1f us  %5.2f TB/s" % (t, 459 / t))
    t2 = timeit(lambda: call("clsr_pgemm_dw_partial", a, Q, T, G, q, Q, None, None, 1, dz0, A0, M, Q, A0, wsd))
    print("fp32  dWp = (a*q)^T dz0               R 328 MB + L2       %6.1f us  %5.2f TB/s" % (t2, 328 / t2))
    t = timeit(lambda: call("clsr_att_l0_bwd_x3", dz0, A0, WtT, KpT, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0, ws))
    print("x3    l0 (da, dq, dU, dV + dWp)       R 328 + 33, W 98 MB %6.1f us  %5.2f TB/s" % (t, 459 / t))
    U, V = torch.randn(Hn * T, A0, device=dev), torch.randn(R, A0, device=dev)
    Wf, Kf = ops.pack_weight(Wp, A0, Q)
    for name in ("clsr_att_l0_fwd", "clsr_att_l0_fwd_x3", "clsr_att_l0_fwd_x6"):
        t = timeit(lambda: call(name, a, Q, q, Q, Wf, Kf, U, A0, V, A0, z0, A0, st, Hn, G, T, Q, A0))
        print("%-20s z0 = U + V + (a*q).Wp + stats   W 328 MB     %6.1f us  %5.2f TB/s" % (name, t, 328 / t))
    # history-level prologue of the short-term attention: keys [Hn*T, 40] -> a [.., 80], U [.., 80] (+ the qh = 40 product term)
    Dk, Qs, qh = 40, 80, 40
    keys = torch.randn(Hn * T, Dk, device=dev)
    Am, Wu, Wp1 = torch.randn(Dk, Qs, device=dev) * 0.3, torch.randn(Qs, A0, device=dev) * 0.3, torch.randn(qh, A0, device=dev) * 0.3
    qhist = torch.randn(Hn, qh, device=dev)
    At, Kpa = ops.pack_weight(Am, Qs, Dk); Wut, Kpu = ops.pack_weight(Wu, A0, Qs); Wpt, Kpp = ops.pack_weight(Wp1, A0, qh)
    a2, U2 = torch.zeros(Hn * T, Qs, device=dev), torch.zeros(Hn * T, A0, device=dev)
    zV = torch.zeros(Hn, A0, device=dev)
    def three():
        call("clsr_pgemm", keys, Dk, 0, 0, None, 0, None, None, 1, At, Kpa, None, None, 0, None, 0, a2, Qs, 0, None, Hn * T, Dk, Qs)
        call("clsr_pgemm", a2, Qs, 0, 0, None, 0, None, None, 1, Wut, Kpu, None, None, 0, None, 0, U2, A0, 0, None, Hn * T, Qs, A0)
        call("clsr_pgemm", a2, Qs, T, 1, qhist, qh, None, None, 1, Wpt, Kpp, None, U2, A0, zV, A0, U2, A0, 0, None, Hn * T, qh, A0)
    t = timeit(three)
    print("fp32  history-level prologue, 3 clsr_pgemm launches   R 33 + W 131 MB  %6.1f us" % t)
    for pc in (2, 3):
        t = timeit(lambda: call("clsr_att_hist_fwd_x3", keys, Dk, At, Kpa, Wut, Kpu, Wpt, Kpp, qhist, qh, Hn, T, Dk, Qs, A0, qh, pc, a2, Qs, U2, A0))
        print("x%d    history-level prologue, one launch              R 33 + W 131 MB  %6.1f us" % (3 * (pc - 1), t))
    W1f, K1f = ops.pack_weight(W1, A1, A0)
    b1 = torch.randn(A1, device=dev)
    t = timeit(lambda: call("clsr_pgemm", z0, A0, 0, 0, None, 0, sc0, sh0, 1, W1f, K1f, b1, None, 0, None, 0, z1, A1, 0, st, M, A0, A1))
    print("fp32  l1 forward z1 = relu(bn z0).W1 + stats   R 328 + W 164 MB   %6.1f us  %5.2f TB/s" % (t, 492 / t))
    t = timeit(lambda: call("clsr_att_l1_fwd", z0, A0, sc0, sh0, W1f, K1f, b1, z1, A1, st, M, A0, A1))
    print("x6    l1 forward z1 = relu(bn z0).W1 + stats   R 328 + W 164 MB   %6.1f us  %5.2f TB/s" % (t, 492 / t))
    if os.environ.get("ONLY_ST"):      # (counter runs: one shape per kernel)
        torch.cuda.synchronize(); sys.exit(0)
    # long-term attention shapes (G = 1, history level)
    Hn, G = 4096, 1
    R, M = Hn, Hn * T
    t = timeit(lambda: call("clsr_att_l0_bwd", dz0, A0, WtT, KpT, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, None, 0, dV, A0))
    print("fp32  l0 long-term (G = 1)            R 66 MB             %6.1f us" % t)
    t = timeit(lambda: call("clsr_att_l0_bwd_x3", dz0, A0, WtT, KpT, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, None, 0, dV, A0, ws))
    print("x3    l0 long-term (G = 1, + dWp)     R 66 MB             %6.1f us" % t)
    t = timeit(lambda: call("clsr_att_l1_bwd", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, None, None, c0, dz1, A1, dz0, A0, None, M, A1, A0))
    print("fp32  l1 pass 2 long-term             R 98 + W 98 MB      %6.1f us" % t)
    t = timeit(lambda: call("clsr_att_l1_bwd_x3", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, None, None, c0, dz0, A0, ws, None, M, A1, A0))
    print("x3    l1 pass 2 long-term (+ dW1)     R 98 + W 66 MB      %6.1f us" % t)

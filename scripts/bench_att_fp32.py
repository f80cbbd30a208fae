"""Isolated timing of the exact-mode (fp32) kernels of the short-term attention block at configs[1] shapes.
usage: python scripts/bench_att_fp32.py"""
import sys, torch
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
    Hn, G, T, Q, A0, A1 = 4096, 5, 50, 80, 80, 40
    R, M = Hn * G, Hn * G * T
    a, q = torch.randn(Hn * T, Q, device=dev), torch.randn(R, Q, device=dev)
    U, V = torch.randn(Hn * T, A0, device=dev), torch.randn(R, A0, device=dev)
    Wp = torch.randn(Q, A0, device=dev) * 0.2
    Wt, Kp = ops.pack_weight(Wp, A0, Q)
    WtT, KpT = ops.pack_weight(Wp, Q, A0, transposed=True)
    z0, z1 = torch.randn(M, A0, device=dev), torch.randn(M, A1, device=dev)
    dz0, dz1 = torch.randn(M, A0, device=dev), torch.zeros(M, A1, device=dev)
    st = torch.zeros(1024, 2, A0, dtype=torch.float64, device=dev)
    t = timeit(lambda: call("clsr_att_l0_fwd", a, Q, q, Q, Wt, Kp, U, A0, V, A0, z0, A0, st, Hn, G, T, Q, A0))
    print("att_l0_fwd  z0 = U+V+(a*q).Wp + stats  13.1 GFLOP, W 328 MB        %6.1f us  %5.1f TFLOP/s" % (t, 13.1e3 / t))
    t = timeit(lambda: call("clsr_pgemm", a, Q, T, G, q, Q, None, None, 1, Wt, Kp, None, U, A0, V, A0, z0, A0, 0, st, M, Q, A0))
    print("pgemm_fast<MUL,UV> (position-tiled, the kernel it replaced)         %6.1f us  %5.1f TFLOP/s" % (t, 13.1e3 / t))
    W1 = torch.randn(A0, A1, device=dev) * 0.3
    W1T, K1T = ops.pack_weight(W1, A0, A1, transposed=True)
    ds = torch.randn(M, device=dev)
    v = lambda n: torch.rand(n, device=dev) + 0.5
    sc1, sh1, wo, c1 = v(A1), torch.randn(A1, device=dev) * 0.3, torch.randn(A1, device=dev), torch.randn(3 * A1, device=dev)
    sc0, sh0, mu0, is0, c0 = v(A0), torch.randn(A0, device=dev) * 0.3, torch.randn(A0, device=dev) * 0.1, v(A0), torch.randn(3 * A0, device=dev)
    t = timeit(lambda: call("clsr_att_l1_bwd", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, mu0, is0, None, None, 0, None, 0, st, M, A1, A0))
    print("att_l1_bwd pass 1 (stats)   6.5 GFLOP, R 492 MB                     %6.1f us  %5.2f TB/s" % (t, 492 / t))
    t = timeit(lambda: call("clsr_att_l1_bwd", z1, A1, ds, sc1, sh1, wo, c1, W1T, K1T, z0, A0, sc0, sh0, None, None, c0, dz1, A1, dz0, A0, None, M, A1, A0))
    print("att_l1_bwd pass 2 (apply)   6.5 GFLOP, R 492 + W 492 MB             %6.1f us  %5.2f TB/s" % (t, 984 / t))
    da, dq = torch.zeros(Hn * T, Q, device=dev), torch.zeros(R, Q, device=dev)
    dU, dV = torch.zeros(Hn * T, A0, device=dev), torch.zeros(R, A0, device=dev)
    t = timeit(lambda: call("clsr_att_l0_bwd", dz0, A0, WtT, KpT, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV, A0))
    print("att_l0_bwd (dz0 -> da, dq, dU, dV)  17 GFLOP incl. transposes, R 393 MB  %6.1f us  %5.2f TB/s" % (t, 393 / t))

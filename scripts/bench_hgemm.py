#!/usr/bin/env python
"""Isolated timing of the bf16-MFMA attention-block kernels at BASELINE configs[1] shapes (short-term attention:
20480 rows x 50 steps = 1.024 M positions).   python scripts/bench_hgemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsr_amd import ops  # noqa: E402
from clsr_amd.ops import call, query  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def pack(W, out_f, in_f, transposed=False):
    Kp = query("clsr_hgemm_kp", in_f)
    buf = torch.zeros(32 * ((out_f + 31) // 32) * Kp, dtype=BF, device="cuda")
    d = ops.pack_desc(W, out_f, in_f, buf, Kp, transposed=transposed)
    tbl, n, mx = ops.pack_table([d], torch.device("cuda"))
    call("clsr_pack_batch_bf16", tbl, n, mx)
    torch.cuda.synchronize()
    return buf, Kp, (d, tbl)


def main():
    dev = "cuda"
    Hn, G, T, Q, A0, A1 = 4096, 5, 50, 80, 80, 40
    R, M = Hn * G, Hn * G * T
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        a, q = torch.randn(Hn * T, Q, device=dev), torch.randn(R, Q, device=dev)
        U, V = torch.randn(Hn * T, A0, device=dev), torch.randn(R, A0, device=dev)
        Wp, W1 = torch.randn(Q, A0, device=dev) * 0.1, torch.randn(A0, A1, device=dev) * 0.1
        z0, z1 = torch.empty(M, A0, dtype=BF, device=dev), torch.empty(M, A1, dtype=BF, device=dev)
        dz0, dz1, daq = torch.empty(M, A0, dtype=BF, device=dev), torch.empty(M, A1, dtype=BF, device=dev), torch.empty(M, Q, dtype=BF, device=dev)
        parts = query("clsr_hgemm_stats_parts", M)
        st = torch.zeros(parts * 2 * 80, dtype=torch.float64, device=dev)
        sc0, sh0, mu0, is0 = (torch.rand(A0, device=dev) + 0.5 for _ in range(4))
        sc1, sh1 = torch.rand(A1, device=dev) + 0.5, torch.randn(A1, device=dev) * 0.1
        wo, c1, c0 = torch.randn(A1, device=dev), torch.randn(3 * A1, device=dev) * 0.1, torch.randn(3 * A0, device=dev) * 0.1
        ds = torch.randn(M, device=dev)
        Wp_h, Kp, k1 = pack(Wp, A0, Q)
        W1_h, K1, k2 = pack(W1, A1, A0)
        W1T_h, K1T, k3 = pack(W1, A0, A1, transposed=True)
        WpT_h, KpT, k4 = pack(Wp, Q, A0, transposed=True)
        rows = [
            ("z0 = U+V+(a*q).Wp + stats   W 164 MB, R 130 MB (L2 x5)", 294,
             lambda: call("clsr_hgemm_mul_uv", a, Q, T, G, q, Q, Wp_h, Kp, U, A0, V, A0, z0, A0, st, M, Q, A0)),
            ("z0, one wave per history group W 164 MB, R 130 MB (L2 x1)", 294,
             lambda: call("clsr_hgemm_l0_group", a, Q, q, Q, Wp_h, Kp, U, A0, V, A0, z0, A0, st, Hn, G, T, Q, A0)),
            ("z1 = relu(bn z0).W1 + stats  R 164, W 82", 246,
             lambda: call("clsr_hgemm", z0, A0, sc0, sh0, 1, W1_h, K1, sh1, z1, A1, st, M, A0, A1)),
            ("l1 bwd pass 1 (stats)        R 82 + 164", 246,
             lambda: call("clsr_hgemm_att_l1_bwd", z1, A1, ds, sc1, sh1, wo, c1, W1T_h, K1T, z0, A0, sc0, sh0, mu0, is0, None,
                          None, 0, None, 0, st, M, A1, A0)),
            ("l1 bwd pass 2 (apply)        R 246, W 82 + 164", 492,
             lambda: call("clsr_hgemm_att_l1_bwd", z1, A1, ds, sc1, sh1, wo, c1, W1T_h, K1T, z0, A0, sc0, sh0, None, None, c0,
                          dz1, A1, dz0, A0, None, M, A1, A0)),
            ("daq = dz0.Wp^T               R 164, W 164", 328,
             lambda: call("clsr_hgemm", dz0, A0, None, None, 0, WpT_h, KpT, None, daq, Q, None, M, A0, Q)),
        ]
        for name, mb, fn in rows:
            t = timeit(fn)
            print("%-62s %7.1f us  %.2f TB/s" % (name, t, mb / t))
        da, dq = torch.zeros(Hn * T, Q, device=dev), torch.zeros(R, Q, device=dev)
        t = timeit(lambda: call("clsr_att_prod_bwd_h", daq, Q, a, Q, q, Q, Hn, G, T, Q, da, Q, dq, Q, 0))
        print("%-62s %7.1f us  %.2f TB/s" % ("att_prod_bwd (daq -> da, dq)   R 164 + 65", t, 229 / t))
        dU, dV = torch.zeros(Hn * T, A0, device=dev), torch.zeros(R, A0, device=dev)
        t = timeit(lambda: call("clsr_att_z0_bwd_reduce_h", dz0, Hn, G, T, A0, dU, dV))
        print("%-62s %7.1f us  %.2f TB/s" % ("att_z0_bwd_reduce (dz0 -> dU, dV)  R 164, W 65", t, 229 / t))
        t = timeit(lambda: call("clsr_att_l0_bwd_h", dz0, A0, WpT_h, WpT_h, KpT, a, Q, q, Q, Hn, G, T, Q, A0, da, Q, dq, Q, dU, A0, dV,
                                A0))
        print("%-62s %7.1f us  %.2f TB/s" % ("att_l0_bwd (dz0 -> da (+dU.Wu^T), dq, dU, dV)  R 164+65, W 131", t, 360 / t))
        wts, out = torch.zeros(R, T, device=dev), torch.zeros(R, 40, device=dev)
        keys = torch.randn(Hn, T, 40, device=dev)
        ln = torch.full((Hn,), T, dtype=torch.int32, device=dev)
        t = timeit(lambda: call("clsr_att_out_fwd_h", z1, sc1, sh1, wo, wo, ln, 1, keys, Hn, G, T, A1, 40, wts, out))
        print("%-62s %7.1f us  %.2f TB/s" % ("att_out_fwd (z1 -> weights, out)  R 82", t, 82 / t))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Micro-driver: launch the heavy kernels of the step at BASELINE config-2 shapes in isolation
(for rocprofv3 --pmc runs and quick A/B timing).  python scripts/prof_kernels.py [dw|pgemm|all]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clsr_amd import ops  # noqa: E402
from clsr_amd.ops import call, query  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = "cuda"
    Hn, G, T = 4096, 5, 50
    R = Hn * G
    M = R * T
    g = torch.Generator(device="cpu").manual_seed(0)
    a = torch.randn(Hn * T, 80, generator=g).to(dev)
    q = torch.randn(R, 80, generator=g).to(dev)
    z0 = torch.randn(M, 80, device=dev)
    dz0 = torch.randn(M, 80, device=dev)
    dz1 = torch.randn(M, 40, device=dev)
    sc = torch.rand(80, device=dev) + 0.5
    sh = torch.randn(80, device=dev) * 0.1
    W = torch.randn(80, 80, device=dev) * 0.1
    ws = torch.empty(query("clsr_pgemm_dw_workspace_floats", M, 80, 80), device=dev)
    dW = torch.zeros(80, 80, device=dev)
    if which in ("dw", "all"):
        t = timeit(lambda: call("clsr_pgemm_dw", a, 80, T, G, q, 80, None, None, 0, dz0, 80, M, 80, 80, 1.0, dW, 80,
                                None, 0, ws))
        print("dw (a*q)^T dz0   1M x 80 x 80 : %8.1f us  (%.1f TF/s, %.0f GB/s algorithmic)" % (
            t, 2.0 * M * 80 * 80 / t / 1e6, (M * 80 * 4 + Hn * T * 80 * 4) / t / 1e3))
        dW1 = torch.zeros(80, 40, device=dev)
        t = timeit(lambda: call("clsr_pgemm_dw", z0, 80, 0, 0, None, 0, sc, sh, 1, dz1, 40, M, 80, 40, 1.0, dW1, 40,
                                None, 0, ws))
        print("dw relu(bn z0)^T dz1 1M x 80 x 40: %8.1f us  (%.0f GB/s algorithmic)" % (t, M * 120 * 4 / t / 1e3))
        x = torch.randn(Hn * T, 40, device=dev)
        dy = torch.randn(Hn * T, 120, device=dev)
        dWs = torch.zeros(40, 80, device=dev)
        t = timeit(lambda: call("clsr_pgemm_dw", x, 40, 0, 0, None, 0, None, None, 0, dy, 120, Hn * T, 40, 80, 1.0,
                                dWs, 80, None, 0, ws))
        print("dw hist^T dPin   205k x 40 x 80 : %8.1f us" % t)
    if which in ("gather", "all"):
        # embedding-history gather: Taobao-shaped (cache resident) and a 100M-item catalogue (HBM resident)
        for name, Vi, Vc, Di, Dc in (("taobao", 64138, 4096, 32, 8), ("catalogue100m", 100_000_000, 10000, 96, 32)):
            it = torch.empty(Vi, Di, device=dev)
            ct = torch.empty(Vc, Dc, device=dev)
            if name == "taobao":
                it.normal_()
                ct.normal_()
            else:
                it.zero_()   # 38 GB: touch every page once so the gather reads committed HBM
                ct.normal_()
            ii = torch.randint(1, Vi, (Hn, T), device=dev, dtype=torch.int32)
            ci = torch.randint(1, Vc, (Hn, T), device=dev, dtype=torch.int32)
            ln = torch.full((Hn,), T, device=dev, dtype=torch.int32)
            D = Di + Dc
            hist = torch.empty(Hn, T, D, device=dev)
            hm, hr = torch.empty(Hn, D, device=dev), torch.empty(Hn, D, device=dev)
            t = timeit(lambda: call("clsr_gather_hist_fwd", it, ct, ii, ci, T, ln, 1, Hn, T, Di, Dc, 3, hist, hm, hr),
                       iters=20)
            nbytes = Hn * T * (D * 8 + 8)
            print("gather_hist_fwd %-14s rows %dB+%dB: %8.1f us  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
                name, Di * 4, Dc * 4, t, nbytes / t / 1e3, nbytes / t / 1e3 / 80))
            del it, ct
    if which in ("pgemm", "all"):
        Wt, Kp = ops.pack_weight(W, 80, 80)
        U = torch.randn(Hn * T, 80, device=dev)
        V = torch.randn(R, 80, device=dev)
        out = torch.empty(M, 80, device=dev)
        t = timeit(lambda: call("clsr_pgemm", a, 80, T, G, q, 80, None, None, 1, Wt, Kp, None, U, 80, V, 80, out, 80, 0,
                                None, M, 80, 80))
        print("pgemm z0 = U+V+(a*q)Wp 1M x 80 x 80: %8.1f us (%.1f TF/s)" % (t, 2.0 * M * 80 * 80 / t / 1e6))
        st = torch.zeros(query("clsr_pgemm_stats_parts", M), 2, 80, dtype=torch.float64, device=dev)
        t = timeit(lambda: call("clsr_pgemm", a, 80, T, G, q, 80, None, None, 1, Wt, Kp, None, U, 80, V, 80, out, 80, 0,
                                st, M, 80, 80))
        print("  + stats                          : %8.1f us" % t)
        W1t, Kp1 = ops.pack_weight(W[:, :40].contiguous(), 40, 80)
        out1 = torch.empty(M, 40, device=dev)
        st1 = torch.zeros(query("clsr_pgemm_stats_parts", M), 2, 40, dtype=torch.float64, device=dev)
        t = timeit(lambda: call("clsr_pgemm", z0, 80, 0, 0, None, 0, sc, sh, 1, W1t, Kp1, sh[:40].contiguous(), None, 0,
                                None, 0, out1, 40, 0, st1, M, 80, 40))
        print("pgemm z1 = relu(bn z0)W1 + stats 1M x 80 x 40: %8.1f us" % t)
        Wtt, Kpt = ops.pack_weight(W, 80, 80, transposed=True)
        t = timeit(lambda: call("clsr_pgemm", dz0, 80, 0, 0, None, 0, None, None, 1, Wtt, Kpt, None, None, 0, None, 0,
                                out, 80, 0, None, M, 80, 80))
        print("pgemm daq = dz0 Wp^T 1M x 80 x 80 plain: %8.1f us (%.1f TF/s)" % (t, 2.0 * M * 80 * 80 / t / 1e6))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()

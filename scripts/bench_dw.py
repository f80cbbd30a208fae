#!/usr/bin/env python
"""Isolated timing of the weight-gradient kernels (fp32-MFMA pgemm_dw vs bf16-MFMA hdw) and of the wide projection
GEMMs at BASELINE configs[1] shapes.   python scripts/bench_dw.py"""
import os
import sys

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
    dev = "cuda"
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        M = 204800
        hist = torch.randn(M, 40, device=dev)
        dPin = torch.randn(M, 480, device=dev)
        for name, (X, ldx, dY, ldy, K, N) in {
            "xw dW      M=204800 K=40 N=480": (hist, 40, dPin, 480, 40, 480),
            "t4 mprev dW        K=40 N=160": (hist, 40, dPin, 480, 40, 160),
            "TT dW              K=80 N=120": (dPin, 480, dPin[:, 120:], 480, 80, 120),
            "gru hidden dW      K=40 N=80 ": (hist, 40, dPin, 480, 40, 80),
        }.items():
            ws = torch.empty(query("clsr_pgemm_dw_workspace_floats", M, K, N), device=dev)
            t0 = timeit(lambda: call("clsr_pgemm_dw_partial", X, ldx, 0, 0, None, 0, None, None, 1, dY, ldy, M, K, N, ws))
            t1 = timeit(lambda: call("clsr_hdw_partial", X, 0, ldx, 0, 0, None, 0, None, None, 1, dY, 0, ldy, M, K, N, ws))
            t3 = float("nan")      # (the generic split-bf16 weight-gradient kernel was removed in round 5)
            mb = (M * (K + N) * 4) / 1e6
            print("%s  fp32 %.1f us  bf16-mfma %.1f us  split-bf16 %.1f us  (%.0f MB -> %.2f / %.2f / %.2f TB/s)"
                  % (name, t0, t1, t3, mb, mb / t0, mb / t1, mb / t3))
        M2 = 1024000
        z0 = torch.randn(M2, 80, device=dev).to(torch.bfloat16)
        dz1 = torch.randn(M2, 40, device=dev).to(torch.bfloat16)
        dz0 = torch.randn(M2, 80, device=dev).to(torch.bfloat16)
        sc, sh = torch.rand(80, device=dev) + 0.5, torch.randn(80, device=dev)
        a, q = torch.randn(204800, 80, device=dev), torch.randn(20480, 80, device=dev)
        ws = torch.empty(query("clsr_pgemm_dw_workspace_floats", M2, 80, 80), device=dev)
        t0 = timeit(lambda: call("clsr_pgemm_dw_partial_h", z0, 1, 80, 0, 0, None, 0, sc, sh, 1, dz1, 1, 40, M2, 80, 40, ws))
        t1 = timeit(lambda: call("clsr_hdw_partial", z0, 1, 80, 0, 0, None, 0, sc, sh, 1, dz1, 1, 40, M2, 80, 40, ws))
        print("dW1 (z0 bf16 aff, dz1 bf16) M=1M K=80 N=40: fp32 %.1f us  bf16-mfma %.1f us  (246 MB)" % (t0, t1))
        t0 = timeit(lambda: call("clsr_pgemm_dw_partial_h", a, 0, 80, 50, 5, q, 80, None, None, 1, dz0, 1, 80, M2, 80, 80, ws))
        t1 = timeit(lambda: call("clsr_hdw_partial", a, 0, 80, 50, 5, q, 80, None, None, 1, dz0, 1, 80, M2, 80, 80, ws))
        print("dWp (a*q fp32, dz0 bf16)    M=1M K=80 N=80: fp32 %.1f us  bf16-mfma %.1f us  (164 MB + L2)" % (t0, t1))
        # the exact-mode (all fp32) forms of the same two products
        z0f, dz1f, dz0f = z0.float(), dz1.float(), dz0.float()
        t0 = timeit(lambda: call("clsr_pgemm_dw_partial", z0f, 80, 0, 0, None, 0, sc, sh, 1, dz1f, 40, M2, 80, 40, ws))
        t3 = float("nan")      # (the generic split-bf16 weight-gradient kernel was removed in round 5)
        print("dW1 all fp32 (exact mode)   M=1M K=80 N=40: fp32 %.1f us  split-bf16 %.1f us (492 MB, 6.5 GFLOP)" % (t0, t3))
        t0 = timeit(lambda: call("clsr_pgemm_dw_partial", a, 80, 50, 5, q, 80, None, None, 1, dz0f, 80, M2, 80, 80, ws))
        t3 = float("nan")      # (the generic split-bf16 weight-gradient kernel was removed in round 5)
        print("dWp all fp32 (exact mode)   M=1M K=80 N=80: fp32 %.1f us  split-bf16 %.1f us (328 MB + L2, 13.1 GFLOP)" % (t0, t3))
        X80, Y80 = torch.randn(M2, 80, device=dev), torch.randn(M2, 80, device=dev)
        t0 = timeit(lambda: call("clsr_pgemm_dw_partial", X80, 80, 0, 0, None, 0, None, None, 1, Y80, 80, M2, 80, 80, ws))
        t3 = float("nan")      # (the generic split-bf16 weight-gradient kernel was removed in round 5)
        print("plain 80x80 all fp32        M=1M             : fp32 %.1f us  split-bf16 %.1f us (655 MB: %.2f TB/s)" % (t0, t3, 655.4 / t3))
        # projection GEMMs
        W = torch.randn(40, 480, device=dev) * 0.1
        Wt, Kp = ops.pack_weight(W, 480, 40)
        Y = torch.empty(M, 480, device=dev)
        t = timeit(lambda: call("clsr_pgemm", hist, 40, 0, 0, None, 0, None, None, 1, Wt, Kp, None, None, 0, None, 0, Y, 480, 0,
                                None, M, 40, 480))
        print("xw GEMM   [204800,40] x [40,480]  %.1f us (writes 393 MB: %.2f TB/s)" % (t, 393.2 / t))
        Wt2, Kp2 = ops.pack_weight(W, 40, 480, transposed=True)
        dh = torch.zeros(M, 40, device=dev)
        t = timeit(lambda: call("clsr_pgemm", dPin, 480, 0, 0, None, 0, None, None, 1, Wt2, Kp2, None, None, 0, None, 0, dh, 40,
                                1, None, M, 480, 40))
        print("xw^T GEMM [204800,480] x [480,40] %.1f us (reads 393 MB: %.2f TB/s)" % (t, 393.2 / t))


if __name__ == "__main__":
    main()

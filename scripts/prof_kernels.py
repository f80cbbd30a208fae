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
            NS = 3      # id sets the launches rotate through (3 x 210 MB touched at the catalogue: past the 256 MiB Infinity Cache)
            ii = [torch.randint(1, Vi, (Hn, T), device=dev, dtype=torch.int32) for _ in range(NS)]
            ci = [torch.randint(1, Vc, (Hn, T), device=dev, dtype=torch.int32) for _ in range(NS)]
            ln = torch.full((Hn,), T, device=dev, dtype=torch.int32)
            D = Di + Dc
            hist = [torch.empty(Hn, T, D, device=dev) for _ in range(NS)]
            hm, hr = torch.empty(Hn, D, device=dev), torch.empty(Hn, D, device=dev)
            turn = [0]

            def gather():
                j = turn[0] % NS
                turn[0] += 1
                call("clsr_gather_hist_fwd", it, ct, ii[j], ci[j], T, ln, 1, Hn, T, Di, Dc, 3, hist[j], hm, hr)

            t = timeit(gather, iters=21)
            nbytes = Hn * T * (D * 8 + 8)
            print("gather_hist_fwd %-14s rows %dB+%dB: %8.1f us  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
                name, Di * 4, Dc * 4, t, nbytes / t / 1e3, nbytes / t / 1e3 / 80))
            del it, ct
    if which == "embed":
        # the other HBM-bound embedding kernels on the 100M-item catalogue (BASELINE configs[4]: rows 384 B + 128 B, uniform
        # ids): deterministic segmented sums of the history + target slices (gather backward), lazy-Adam row update, and
        # the bf16-table forms -- >= 22 launches each, for rocprofv3 --kernel-trace / --pmc (scripts/prof_embed.sh)
        Vi, Vc, Di, Dc, B = 100_000_000, 10000, 96, 32, Hn * G
        taobao = os.environ.get("EMBED_SHAPE") == "taobao"      # BASELINE configs[1]: small Zipf vocabularies, narrow rows
        if taobao:
            Vi, Vc, Di, Dc = 64138, 4096, 32, 8
        D, n = Di + Dc, Hn * T
        gi, gc = torch.zeros(Vi, Di, device=dev), torch.zeros(Vc, Dc, device=dev)      # gradient tables (touches the pages)
        NS = 3          # sorted lists + gradient tensors the launches rotate through (past the 256 MiB Infinity Cache)
        ln = torch.full((Hn,), T, device=dev, dtype=torch.int32)
        keys, perm, iis, cis = [], [], [], []
        for j in range(NS):
            ii = torch.randint(1, Vi, (Hn, T), device=dev, dtype=torch.int32)
            ci = torch.randint(1, Vc, (Hn, T), device=dev, dtype=torch.int32)
            if taobao:
                ii = (torch.rand(Hn, T, device=dev).pow(8.0) * (Vi - 2)).int() + 1
                ci = (torch.rand(Hn, T, device=dev).pow(8.0) * (Vc - 2)).int() + 1
            it, ct = torch.randint(1, Vi, (B,), device=dev, dtype=torch.int32), torch.randint(1, Vc, (B,), device=dev, dtype=torch.int32)
            k_ = [torch.empty(n + B, dtype=torch.int32, device=dev) for _ in range(2)]
            p_ = [torch.empty(n + B, dtype=torch.int32, device=dev) for _ in range(2)]
            rows = [(ii.data_ptr(), k_[0].data_ptr(), p_[0].data_ptr(), Hn, T, T, 17 if taobao else 27, it.data_ptr(), B, 1),
                    (ci.data_ptr(), k_[1].data_ptr(), p_[1].data_ptr(), Hn, T, T, 12 if taobao else 14, ct.data_ptr(), B, 1)]
            wss = torch.empty(query("clsr_sort_ids_stable_workspace_bytes", 2 * (n + B), 2), dtype=torch.uint8, device=dev)
            t = timeit(lambda: ops.sort_ids_stable_multi(rows, wss), iters=5 if j == 0 else 1, warm=1)
            if j == 0:
                print("stable radix sort of 2 x %d ids (27 / 14 bits): %8.1f us" % (n + B, t))
            keys.append(k_); perm.append(p_); iis.append(ii); cis.append(ci)
        ii, ci = iis[0], cis[0]
        dh32 = [torch.randn(n, D, device=dev) * 1e-3 for _ in range(NS)]
        dtarget = torch.randn(B, D, device=dev) * 1e-3
        for tag, ds, sa in (("fp32 d(hist)", dh32, 4), ("bf16 d(hist)", [d.to(torch.bfloat16) for d in dh32], 2)):
            jobs = []
            for j in range(NS):
                d = ds[j]
                bf = int(d.dtype == torch.bfloat16)
                sites = [(d.data_ptr(), 0, 0, 0, keys[j][0].data_ptr(), perm[j][0].data_ptr(), ln.data_ptr(), gi.data_ptr(), 0, n + B, bf,
                          1, T, D, 0, Di, 3, Di, 0, 1, dtarget.data_ptr(), 0, n, D, 0, 0, 0),
                         (d.data_ptr(), 0, 0, 0, keys[j][1].data_ptr(), perm[j][1].data_ptr(), ln.data_ptr(), gc.data_ptr(), 0, n + B, bf,
                          1, T, D, Di, Dc, 3, Dc, 0, 1, dtarget.data_ptr(), 0, n, D, Di, 0, 0)]
                which_sites = os.environ.get("EMBED_SITES", "both")
                sites = sites[:1] if which_sites == "item" else sites[1:] if which_sites == "cate" else sites
                jobs.append((sites, torch.zeros(ops.segsum_workspace_bytes(sites), dtype=torch.uint8, device=dev)))
            turn = [0]

            def seg():
                sites, wsg = jobs[turn[0] % NS]
                turn[0] += 1
                ops.segsum_multi(sites, wsg)

            t = timeit(seg, iters=24)
            nbytes = n * D * sa + n * D * 4 + 2 * n * 4 + B * D * 8 + 2 * B * 4
            print("segmented sums (item + category, history + target slices, stored once, ONE launch), %s: %8.1f us  %.0f GB/s "
                  "algorithmic (%.1f%% of 8 TB/s)" % (tag, t, nbytes / t / 1e3, nbytes / t / 1e3 / 80))
        keys = keys[0]
        del dh32
        if taobao:
            return
        ids = torch.unique(keys[0].long()).int()
        nrows = ids.numel()
        count = torch.tensor([nrows, 0], dtype=torch.int32, device=dev)
        tb = torch.zeros(Vi, Di, device=dev)
        m_, v_ = torch.zeros(Vi, Di, device=dev), torch.zeros(Vi, Di, device=dev)
        fl = torch.zeros(Vi, dtype=torch.uint8, device=dev)
        ss = torch.ones(1, dtype=torch.float64, device=dev)
        st = torch.tensor([1.0, 0.9, 0.999, 0.0, 0.0], dtype=torch.float64, device=dev)
        t = timeit(lambda: call("clsr_table_adam_rows", tb, gi, m_, v_, fl, ids, count, nrows, Di, ss, 1, 1, 2.0, st, 0.9, 0.999,
                                1e-8), iters=22)
        nbytes = nrows * Di * 8 * 4
        print("lazy-Adam rows, %d rows of %d B (fp32 table): %8.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % (
            nrows, Di * 4, t, nbytes / t / 1e3, nbytes / t / 1e3 / 80))
        del m_, v_
        tbh, cth = tb.to(torch.bfloat16), torch.zeros(Vc, Dc, device=dev, dtype=torch.bfloat16)
        del tb
        m_, v_ = torch.zeros(Vi, Di, device=dev), torch.zeros(Vi, Di, device=dev)
        t = timeit(lambda: call("clsr_table_adam_rows_h", tbh, gi, m_, v_, fl, ids, count, nrows, Di, ss, 1, 1, 2.0, st, 0.9,
                                0.999, 1e-8), iters=22)
        nbytes = nrows * Di * (6 * 4 + 2 * 2)
        print("lazy-Adam rows, bf16 table (fp32 moments)      : %8.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % (
            t, nbytes / t / 1e3, nbytes / t / 1e3 / 80))
        histh = torch.empty(Hn, T, D, device=dev, dtype=torch.bfloat16)
        hm, hr = torch.empty(Hn, D, device=dev), torch.empty(Hn, D, device=dev)
        t = timeit(lambda: call("clsr_gather_hist_fwd_h", tbh, cth, ii, ci, T, ln, 1, Hn, T, Di, Dc, 3, histh, 1, hm, hr), iters=22)
        nbytes = n * (D * 4 + 8)
        print("gather_hist_fwd_h (bf16 tables, bf16 hist)      : %8.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % (
            t, nbytes / t / 1e3, nbytes / t / 1e3 / 80))
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

#!/usr/bin/env python
"""The history gather of BASELINE configs[4] in isolation (uniform ids over a 100M-row item table, three rotating id sets
with an output tensor each, HIP events) -- A/B of kernel variants: CLSR_LIB=<variant .so> python scripts/bench_gather.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel  # noqa: E402
from clsr_amd import ops  # noqa: E402

Vi, Vc, Di, Dc, T, P, G, NSETS = 100_000_000, 10000, 96, 32, 50, 4096, 5, 3
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda"
D = Di + Dc
item, cate = torch.empty(Vi, Di, device=dev).normal_(), torch.randn(Vc, Dc, device=dev)
seq_len = torch.full((P * G,), T, dtype=torch.int32, device=dev)
hm, hr = torch.empty(P, D, device=dev), torch.empty(P, D, device=dev)
sets = []
for j in range(NSETS):
    g = torch.Generator(device=dev).manual_seed(77 + j)
    ih = torch.randint(1, Vi, (P * G, T), generator=g, device=dev, dtype=torch.int32)
    ch = torch.randint(1, Vc, (P * G, T), generator=g, device=dev, dtype=torch.int32)
    sets.append((ih, ch, torch.empty(P, T, D, device=dev)))
turn = [0]


def run():
    ih, ch, out = sets[turn[0] % NSETS]
    turn[0] += 1
    ops.call("clsr_gather_hist_fwd", item, cate, ih, ch, G * T, seq_len, G, P, T, Di, Dc, 3, out, hm, hr)


nbytes = P * T * D * 8 + 2 * P * T * 4
for r in range(reps):
    t = time_kernel(run, iters=21)
    print("gather fwd: %.2f us  %.1f GB/s  frac %.4f" % (t * 1e6, nbytes / t / 1e9, nbytes / t / 8e12), flush=True)
ih, ch, out = sets[0]
ops.call("clsr_gather_hist_fwd", item, cate, ih, ch, G * T, seq_len, G, P, T, Di, Dc, 3, out, hm, hr)
rows = ih.view(P, G, T)[:, 0].long()
assert os.environ.get("CLSR_LIB") or torch.equal(out[:, :, :Di], item[rows]) and torch.equal(out[:, :, Di:], cate[ch.view(P, G, T)[:, 0].long()])
assert os.environ.get("CLSR_LIB") or torch.allclose(hm[:, :Di], item[rows].mean(1), atol=1e-5)
print("ok")

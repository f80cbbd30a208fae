#!/usr/bin/env python
"""The item-site segmented sums of BASELINE configs[4] in isolation (rotating id sets, HIP events) -- A/B of kernel
variants: CLSR_LIB=build/abl/lib_<name>.so python scripts/bench_segsum.py [V] [reps] [bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel  # noqa: E402
from clsr_amd import ops  # noqa: E402

V = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"      # d(hist) stored as bf16 (the target rows' gradients stay fp32)
P, G, T, D, Di, NSETS = 4096, 5, 50, 128, 96, 3
n, B = P * T, P * G
dev = "cuda"
grad = torch.zeros(V, Di, device=dev)
seq_len = torch.full((P * G,), T, dtype=torch.int32, device=dev)
dtarget = torch.randn(B, D, device=dev) * 1e-3
jobs = []
for j in range(NSETS):
    g = torch.Generator(device=dev).manual_seed(4242 + j)
    ids = torch.randint(1, V, (n,), generator=g, device=dev, dtype=torch.int64)
    tgt = torch.randint(1, V, (P,), generator=g, device=dev, dtype=torch.int64)
    ids = torch.cat([ids, tgt[torch.randint(0, P, (B,), generator=g, device=dev)]])
    k_, p_ = torch.sort(ids, stable=True)
    k_, p_ = k_.int(), p_.int()
    d = torch.randn(n, D, device=dev) * 1e-3
    if bf16:
        d = d.to(torch.bfloat16)
    rows = [(d.data_ptr(), 0, 0, 0, k_.data_ptr(), p_.data_ptr(), seq_len.data_ptr(), grad.data_ptr(), 0,
             n + B, int(bf16), G, T, D, 0, Di, 3, Di, 0, 1, dtarget.data_ptr(), 0, n, D, 0, 0, 0)]
    ws = torch.zeros(ops.segsum_workspace_bytes(rows), dtype=torch.uint8, device=dev)
    jobs.append((rows, ws, k_, p_, d))
turn = [0]


def run():
    rows, ws = jobs[turn[0] % NSETS][:2]
    turn[0] += 1
    ops.segsum_multi(rows, ws)


nbytes = n * Di * (6 if bf16 else 8) + 2 * n * 4 + B * Di * 8 + 2 * B * 4
for r in range(reps):
    t = time_kernel(run, iters=21)
    print("segsum item site: %.2f us  %.1f GB/s  frac %.4f" % (t * 1e6, nbytes / t / 1e9, nbytes / t / 8e12), flush=True)
# the sums are right: every touched row against an index_add of the same slices (fp32, order of the sorted list)
rows, ws, k_, p_, d = jobs[0]
grad.index_fill_(0, k_.long(), 0.0)
ops.segsum_multi(rows, ws)
src = torch.cat([d[:, :Di].float(), dtarget[:, :Di]])[p_.long()]
uk, inv = torch.unique(k_.long(), return_inverse=True)
exp = torch.zeros(uk.numel(), Di, device=dev, dtype=torch.float64).index_add_(0, inv, src.double())
err = (grad[uk].double() - exp).abs().max().item()
print("max abs err %.3e (max |exp| %.3e)" % (err, exp.abs().max().item()))
assert err < 1e-7 or os.environ.get("CLSR_LIB")

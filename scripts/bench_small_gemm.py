"""Isolated timing of the small row-level products of the heads (B = 20480 rows).  usage: python scripts/bench_small_gemm.py"""
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
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    M = 20480
    for K, N in ((164, 80), (80, 40), (80, 100), (100, 64), (200, 80)):
        X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) * 0.1
        Wt, Kp = ops.pack_weight(W, N, K)
        Y = torch.zeros(M, N, device=dev)
        stats = torch.zeros(query("clsr_pgemm_stats_parts", M), 2, N, dtype=torch.float64, device=dev)
        bias = torch.zeros(N, device=dev)
        for stn, stt in (("stats", stats), ("no stats", None)):
            t = timeit(lambda: call("clsr_pgemm", X, K, 0, 0, None, 0, None, None, 0, Wt, Kp, bias, None, 0, None, 0, Y, N, 0, stt, M, K, N))
            print("pgemm [%d x %d] x [%d x %d] %-8s: %6.1f us" % (M, K, K, N, stn, t))

"""Isolated timing of the fused encoder-backward kernel (csrc/encbwd.hip) at configs[1] shapes (M = 4096 * 50 positions).
usage: python scripts/bench_encbwd.py      (CLSR_LIB=build/abl/lib_<variant>.so for ablation builds)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from clsr_amd import ops
from clsr_amd.ops import call, query
dev = "cuda:0"
M, n = 4096 * 50, 40
r = lambda *s: torch.randn(*s, device=dev) * 0.3
dPin, hist, hp1, hp2, mp, TT = r(M, 480), r(M, n), r(M, n), r(M, n), r(M, n), r(M, 2 * n)
g1, g2 = torch.rand(M, 3 * n, device=dev), torch.rand(M, 3 * n, device=dev)
Wt, Kp = ops.pack_weight(r(n, 480), n, 480, transposed=True)
dhist = torch.zeros(M, n, device=dev)
wss = [torch.zeros(query("clsr_enc_bwd_fused_workspace_floats", M, i), device=dev) for i in range(7)]
fn = lambda: call("clsr_enc_bwd_fused", dPin, hist, hp1, g1, mp, TT, hp2, g2, Wt, Kp, dhist, *wss, M)
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); 
for _ in range(10): fn()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) * 100
flops = 2.0 * M * (44800 + 480 * 40)
print("enc_bwd_fused: %.1f us  (%.1f TFLOP/s useful fp32, %.0f GB/s of the 0.56 GB it must read)" % (us, flops / us * 1e-6, 0.56e3 / us * 1e3))
wsx = [torch.zeros(query("clsr_enc_bwd_fused_x3_workspace_floats", M, i), device=dev) for i in range(7)]
fx = lambda: call("clsr_enc_bwd_fused_x3", dPin, hist, hp1, g1, mp, TT, hp2, g2, *wsx, M)
for _ in range(3): fx()
e0.record()
for _ in range(10): fx()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) * 100
print("enc_bwd_fused_x3 (weight gradients only, split-bf16): %.1f us  (%.0f GB/s of the 0.49 GB it must read)" % (us, 0.49e3 / us * 1e3))
W2 = r(n, 480)
Wt2, Kp2 = ops.pack_weight(W2, n, 480, transposed=True)
f3 = lambda: call("clsr_proj_x3_wide", dPin, 480, Wt2, Kp2, None, dhist, n, M, 480, n, 3, 1)
f1 = lambda: call("clsr_pgemm", dPin, 480, 0, 0, None, 0, None, None, 0, Wt2, Kp2, None, None, 0, None, 0, dhist, n, 1, None, M, 480, n)
for name, fn_ in (("clsr_proj_x3_wide (three pieces)", f3), ("clsr_pgemm", f1)):
    for _ in range(3): fn_()
    e0.record()
    for _ in range(10): fn_()
    e1.record(); e1.synchronize()
    print("d(hist) += dPin . W_x^T through %s: %.1f us" % (name, e0.elapsed_time(e1) * 100))

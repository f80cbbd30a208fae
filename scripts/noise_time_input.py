"""Run-to-run spread of the Time4LSTM time-input gradients at configs[1] full size: de-duplicated vs replicated net and
the same net twice (nondeterminism of the upstream float atomics).  usage: python scripts/noise_time_input.py [reps]"""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_fullsize_gpu import _net
from clsr_amd.synthetic import CONFIGS, synthetic_feed
cfg = CONFIGS["taobao"]; P, T, G = cfg["P"], cfg["T"], 5
feed = synthetic_feed(P, T, cfg["Vu"], cfg["Vi"], cfg["Vc"], G=G, lengths="lognormal")
names = ["sequential/clsr/short_term/time4lstm/time4lstm_cell/_time_input_w1", "sequential/clsr/short_term/time4lstm/time4lstm_cell/_time_input_w2",
         "sequential/clsr/short_term/time4lstm/time4lstm_cell/_time_input_bias1"]
def grads(dedup):
    hp, net = _net(cfg, P, dedup=dedup, seed=1)
    return net
base = grads(True)
sd = base.state_dict()
def run(dedup):
    _, net = _net(cfg, P, dedup=dedup, seed=1)
    net.load_state_dict(sd)
    net.capture_grads = True
    net.train_step(net.upload(feed, True)); torch.cuda.synchronize()
    return {k: net.captured["dense"][k].double().cpu() for k in names}, max(float(g.abs().max()) for g in net.captured["dense"].values())
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
A = [run(True) for _ in range(reps)]
R = [run(False) for _ in range(reps)]
gs = R[0][1]
for k in names:
    gmax = float(R[0][0][k].abs().max())
    tol = 2e-3 * gmax + 2e-5 * gs
    dd = [float((a[0][k] - r[0][k]).abs().max()) for a in A for r in R]
    aa = [float((A[i][0][k] - A[0][0][k]).abs().max()) for i in range(1, reps)]
    rr = [float((R[i][0][k] - R[0][0][k]).abs().max()) for i in range(1, reps)]
    print("%s: max|g| %.3e tol %.3e | dedup-vs-replicated max %.3e median %.3e | dedup run-to-run max %.3e | replicated run-to-run max %.3e"
          % (k.split("/")[-1], gmax, tol, max(dd), float(np.median(dd)), max(aa), max(rr)))

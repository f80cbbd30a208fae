"""Phase times of the two persistent heads launches (csrc/headsfused.hip built with -DHF_TIMING:
bash scripts/build_variant.sh hft -DHF_TIMING headsfused.hip; CLSR_LIB=$PWD/build/abl/lib_hft.so python scripts/heads_phases.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from clsr_amd.net import CLSRNet  # noqa: E402
from clsr_amd.ops import query  # noqa: E402
from clsr_amd.synthetic import CONFIGS, synthetic_feed  # noqa: E402

cfg = dict(CONFIGS["taobao"])
P = int(os.environ.get("P", cfg["P"]))
feed = synthetic_feed(P, cfg["T"], cfg["Vu"], cfg["Vi"], cfg["Vc"], G=5, lengths="lognormal", seed=5)
net = CLSRNet(bench.build_hparams(cfg, P), dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"]), seed=2)
f = net.upload(feed, True)
for _ in range(6):
    net.train_step(f)
torch.cuda.synchronize()
ws = net._heads_ws()
nb = int(query("clsr_heads_fused_workspace_bytes"))
st = ws.view(torch.uint8)[nb - 1024 - 65536: nb - 65536].view(torch.int64).cpu().tolist()
dbg = ws.view(torch.uint8)[nb - 65536: nb].view(torch.int64).cpu()[: 8 * 256 * 2].view(8, 256, 2)
for name, base in (("launch 1", 0), ("launch 2", 64)):
    v = st[base: base + 64]
    t0 = v[0]
    print(name)
    prev = t0
    for i, t in enumerate(v):
        if t == 0 or i == 0:
            continue
        print("  stamp %2d  +%7.2f us  (total %7.2f)" % (i, (t - prev) / 100.0, (t - t0) / 100.0))
        prev = t
for bi in range(8):
    a, e = dbg[bi, :, 0].double(), dbg[bi, :, 1].double()
    if float(a.min()) == 0:
        continue
    a0 = float(a.min())
    print("barrier %d: arrivals spread %.2f us (median +%.2f), last arrival -> first exit %.2f us, -> last exit %.2f us"
          % (bi, (float(a.max()) - a0) / 100, (float(a.median()) - a0) / 100, (float(e.min()) - float(a.max())) / 100,
             (float(e.max()) - float(a.max())) / 100))
print("error word:", query("clsr_heads_fused_error", ws.data_ptr()))

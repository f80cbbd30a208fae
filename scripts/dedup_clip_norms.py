#!/usr/bin/env python
"""How far do the embedding IndexedSlices norms sit from max_grad_norm, and how much does history de-duplication change
them?  (VERDICT r2 "weak" #2: the de-duplicated step clips item / cate slices by the norm of the SUMMED replica slices,
the reference -- ``tf.clip_by_norm`` on IndexedSlices, base_model.py:289-297 -- by the norm of the un-summed ones.)

Trains two nets from the same initialisation on the same stream of synthetic Taobao-shaped batches (configs[1]: 4096
positives x5 rows, seq_len 50, a fresh batch every step): one replicated (reference-exact norms), one de-duplicated;
after every step the squared norms of the slices (CLSRNet.sumsq_tab) are read back.

    python scripts/dedup_clip_norms.py [steps=200] [lengths=lognormal]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from clsr_amd.net import CLSRNet  # noqa: E402
from clsr_amd.synthetic import CONFIGS, synthetic_feed  # noqa: E402


def norms(ss):
    ss = np.asarray(ss)
    return dict(item=float(np.sqrt(ss[0] + ss[2] + ss[4])), cate=float(np.sqrt(ss[1] + ss[3] + ss[5])),
                user_long=float(np.sqrt(ss[6] + ss[8])), user_short=float(np.sqrt(ss[7] + ss[9])))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    lengths = sys.argv[2] if len(sys.argv) > 2 else "lognormal"
    cfg = CONFIGS["taobao"]
    hp = bench.build_hparams(cfg, cfg["P"])
    dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
    nets = {"replicated (reference)": CLSRNet(hp, dims, seed=0, dedup_histories=False),
            "de-duplicated": CLSRNet(hp, dims, seed=0, dedup_histories=True)}
    clip = float(hp.max_grad_norm)
    hist = {k: [] for k in nets}
    for s in range(steps):
        feed = synthetic_feed(cfg["P"], cfg["T"], cfg["Vu"], cfg["Vi"], cfg["Vc"], G=5, lengths=lengths, seed=1000 + s)
        for k, net in nets.items():
            net.train_step(net.upload(feed, True))
            torch.cuda.synchronize()
            hist[k].append(norms(net.sumsq_tab.cpu().numpy()))
    out = {"steps": steps, "lengths": lengths, "max_grad_norm": clip, "batch": "4096 positives x5 rows, seq_len 50", "tables": {}}
    for tab in ("item", "cate", "user_long", "user_short"):
        a = np.array([h[tab] for h in hist["replicated (reference)"]])
        b = np.array([h[tab] for h in hist["de-duplicated"]])
        out["tables"][tab] = {
            "reference_norm_max": round(float(a.max()), 5), "reference_norm_median": round(float(np.median(a)), 5),
            "reference_norm_first_step": round(float(a[0]), 5),
            "fraction_of_steps_clipped_reference": round(float((a > clip).mean()), 4),
            "dedup_norm_max": round(float(b.max()), 5), "fraction_of_steps_clipped_dedup": round(float((b > clip).mean()), 4),
            "dedup_over_reference_norm_ratio_median": round(float(np.median(b / a)), 4),
            "dedup_over_reference_norm_ratio_max": round(float((b / a).max()), 4)}
    p0 = nets["replicated (reference)"].tables["item"]
    p1 = nets["de-duplicated"].tables["item"]
    out["item_table_max_abs_difference_after_training"] = float((p0 - p1).abs().max())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

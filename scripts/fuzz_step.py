#!/usr/bin/env python
"""Shape fuzzing of the training / scoring step against the CPU oracle (test infrastructure, like tests/).

Random small shapes -- embedding width D (= Du = H), history length T, positives P, rows per positive G, encoder
kind, MLP widths, losses, sequence-length patterns incl. length-1 histories and P = 1 -- one training step and one
scoring pass each, compared with oracle/clsr_oracle.py: logits, the five loss terms, every dense gradient.

    python scripts/fuzz_step.py [n_cases] [seed]
"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import build_hparams  # noqa: E402
from clsr_amd.net import CLSRNet  # noqa: E402
from clsr_amd.synthetic import synthetic_feed  # noqa: E402
from oracle import clsr_oracle as O  # noqa: E402


def close(got, exp, rtol, atol):
    got = torch.as_tensor(got).detach().double().cpu().reshape(-1)
    exp = torch.as_tensor(exp).detach().double().cpu().reshape(-1)
    if got.shape != exp.shape:
        return "shape %s vs %s" % (tuple(got.shape), tuple(exp.shape))
    err = (got - exp).abs()
    if float((err - (atol + rtol * exp.abs())).max()) > 0:
        return "max abs err %.3e (max |exp| %.3e)" % (float(err.max()), float(exp.abs().max()))
    return None


def one_case(rng, idx):
    D = int(rng.choice([8, 12, 16, 20, 24, 32, 40, 48, 52, 64, 96, 128]))
    Dc = int(rng.choice([4, 8])) if D > 8 else 4
    T = int(rng.choice([1, 2, 3, 5, 8, 10, 17, 50]))
    P = int(rng.choice([1, 2, 3, 7, 16, 33, 64]))
    G = int(rng.choice([2, 3, 5, 10]))
    enc = str(rng.choice(["time4lstm", "gru", "lstm"]))
    over = dict(
        sequential_model=enc, train_num_ngs=G - 1,
        contrastive_loss=str(rng.choice(["triplet", "bpr"])),
        att_fcn_layer_sizes=[int(rng.choice([8, 20, 40, 80])), int(rng.choice([4, 12, 40]))],
        layer_sizes=[int(rng.choice([16, 36, 100])), int(rng.choice([8, 64]))],
        contrastive_recent_k=int(rng.choice([1, 3, 5])),
        contrastive_length_threshold=int(rng.choice([0, 2, 5])),
        is_clip_norm=int(rng.choice([0, 1])),
        optimizer=str(rng.choice(["adam", "lazyadam"])),
    )
    if os.environ.get("FUZZ_CASE"):      # "D,Dc,T,P,G,encoder": pin the shape, keep the random rest
        tok = os.environ["FUZZ_CASE"].split(",")
        D, Dc, T, P, G = (int(x) for x in tok[:5])
        over.update(sequential_model=tok[5], train_num_ngs=G - 1)
    cfg = dict(T=T, Di=D - Dc, Dc=Dc, Du=D, H=D, Vu=50, Vi=200, Vc=12)
    desc = "case %d: D=%d Dc=%d T=%d P=%d G=%d %s" % (idx, D, Dc, T, P, G, over)
    one_case.desc = desc
    hp = build_hparams(cfg, P, **over)
    dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
    lengths = str(rng.choice(["full", "uniform", "lognormal"])) if T > 1 else "full"
    feed = synthetic_feed(P, T, dims["Vu"], dims["Vi"], dims["Vc"], G=G, lengths=lengths,
                          ids="uniform", seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.5 and T > 1:     # force some length-1 histories
        k = max(1, P // 3)
        rows = np.arange(k * G)
        for key in ("item_history", "item_cate_history", "mask", "time_diff", "time_from_first_action",
                    "time_to_now"):
            feed[key][rows, 1:] = 0
    problems = []
    for dedup in (True, False):
        params32 = O.init_params(dims, hp, seed=idx, scale_dense=8.0)
        net = CLSRNet(hp, dims, device="cuda:0", seed=0, dedup_histories=dedup)
        sd = dict(params32)
        sd.update(O.init_bn_state(params32))
        net.load_state_dict(sd, strict=True)
        params = type(params32)((k, v.double()) for k, v in params32.items())
        tf = O.to_torch_feed(feed, dtype=torch.float64)
        bn = O.init_bn_state(params)
        # scoring (moving statistics)
        ev = O.forward(params, bn, tf, hp, False)
        got_ev = net.forward(net.upload(feed, False), False)
        torch.cuda.synchronize()
        e = close(got_ev["logit"], ev["logit"], 1e-4, 1e-4)
        if e:
            problems.append("dedup=%s eval logit: %s" % (dedup, e))
        adam = O.init_adam(params)
        _, _, _, ls, _, _, out = O.train_step(params, bn, adam, 1, tf, hp)
        net.capture_grads = True
        got = net.train_step(net.upload(feed, True))
        torch.cuda.synchronize()
        e = close(got["logit"], out["logit"], 1e-4, 1e-4)
        if e:
            problems.append("dedup=%s train logit: %s" % (dedup, e))
        gl = net.read_losses()
        for k in ("loss", "data_loss", "regular_loss", "contrastive_loss", "discrepancy_loss"):
            e = close([gl[k]], [float(ls[k])], 1e-5, 1e-6)
            if e:
                problems.append("dedup=%s %s: %s" % (dedup, k, e))
        if P == 1:
            # one positive: its in-batch "negatives" are the same item, every row of the batch is identical, every
            # batch-norm output sits exactly on the ReLU kink (y = beta = 0) and the oracle's float64 masks are
            # decided by rounding noise -- gradients are not comparable there; values and losses are
            continue
        raw = out["raw_grads"]
        floor = 4e-6 * max(float(raw[n].abs().max()) for n in net.dense_names)
        for name in net.dense_names:
            scale = float(raw[name].abs().max()) + 1e-12
            e = close(net.captured["dense"][name], raw[name], 2e-3, 2e-4 * scale + floor)
            if e:
                problems.append("dedup=%s grad %s: %s" % (dedup, name, e))
                if os.environ.get("FUZZ_TRACE"):
                    print(name, "got", net.captured["dense"][name].reshape(-1)[:6].tolist(), "exp",
                          raw[name].reshape(-1)[:6].tolist())
    return desc, problems


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n):
        try:
            desc, problems = one_case(rng, i)
        except Exception as exc:
            bad += 1
            print("CRASH " + getattr(one_case, "desc", "case %d" % i))
            print("    " + traceback.format_exc().strip().splitlines()[-1][:400])
            if os.environ.get("FUZZ_TRACE"):
                traceback.print_exc()
            continue
        if problems:
            bad += 1
            print("FAIL " + desc)
            for p in problems[:8]:
                print("    " + p)
        else:
            print("ok   " + desc.split(" {")[0])
    print("%d of %d cases with problems" % (bad, n))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Shape fuzzing of the training / scoring step against the CPU oracle (test infrastructure, like tests/).

Random small shapes -- embedding width D (= Du = H), history length T, positives P, rows per positive G, encoder
kind, MLP widths, losses, sequence-length patterns incl. length-1 histories and P = 1 -- one training step and one
scoring pass each, compared with oracle/clsr_oracle.py: logits, the five loss terms, every dense gradient.

    python scripts/fuzz_step.py [n_cases] [seed] [clsr,gru4rec,din,sli_rec,a2svd,dien]
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
from clsr_amd.seqnet import SeqNet  # noqa: E402
from clsr_amd.synthetic import synthetic_feed  # noqa: E402
from oracle import clsr_oracle as O  # noqa: E402
from oracle import sibling_oracle as SO  # noqa: E402

# fp32 = the reference's arithmetic (every product at fp32 accuracy); fp32x3 = two-piece split-bf16 products (FUZZ_PRECISION)
PRECISION = os.environ.get("FUZZ_PRECISION", "fp32")
SIB_TYPES = {"gru4rec": "GRU4Rec", "din": "DIN", "sli_rec": "sli_rec", "a2svd": "A2SVD", "dien": "DIEN"}


def close(got, exp, rtol, atol):
    got = torch.as_tensor(got).detach().double().cpu().reshape(-1)
    exp = torch.as_tensor(exp).detach().double().cpu().reshape(-1)
    if got.shape != exp.shape:
        return "shape %s vs %s" % (tuple(got.shape), tuple(exp.shape))
    err = (got - exp).abs()
    if float((err - (atol + rtol * exp.abs())).max()) > 0:
        return "max abs err %.3e (max |exp| %.3e)" % (float(err.max()), float(exp.abs().max()))
    return None


def one_case(rng, idx, kind="clsr"):
    D = int(rng.choice([8, 12, 16, 20, 24, 32, 40, 48, 52, 64, 96, 128]))
    Dc = int(rng.choice([4, 8])) if D > 8 else 4
    T = int(rng.choice([1, 2, 3, 5, 8, 10, 17, 50]))
    P = int(rng.choice([1, 2, 3, 7, 16, 33, 64]))
    G = int(rng.choice([2, 3, 5, 10]))
    if os.environ.get("FUZZ_LONG"):      # histories across the 64-step chunks of the attention kernels (slow oracle)
        T = int(rng.choice([63, 64, 65, 100, 129, 250]))
        P = int(rng.choice([2, 3, 5]))
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
        if len(tok) > 9:                 # ",a0,a1,l0,l1": pin the MLP widths too
            over.update(att_fcn_layer_sizes=[int(tok[6]), int(tok[7])], layer_sizes=[int(tok[8]), int(tok[9])])
    if kind == "clsr":
        over.update(interest_evolve=bool(rng.random() < 0.7), predict_long_short=bool(rng.random() < 0.7),
                    manual_alpha=bool(rng.random() < 0.25), manual_alpha_value=float(rng.choice([0.0, 0.3, 1.0])),
                    embed_l2=float(rng.choice([0.0, 1e-6, 1e-3])), layer_l2=float(rng.choice([0.0, 1e-6, 1e-3])),
                    discrepancy_loss_weight=float(rng.choice([0.0, 0.01, 0.5])),
                    contrastive_loss_weight=float(rng.choice([0.0, 0.1, 1.0])),
                    contrastive_margin=float(rng.choice([0.5, 1.0, 2.0])),
                    max_grad_norm=float(rng.choice([0.01, 2.0])))
    if os.environ.get("FUZZ_OVERRIDE"):  # json dict applied on top of the drawn hyper-parameters (probing a failing case)
        import json
        over.update(json.loads(os.environ["FUZZ_OVERRIDE"]))
    if kind != "clsr":    # the siblings: free hidden / attention / user widths (reference sli_rec.yaml & co.)
        over.update(model_type=SIB_TYPES[kind], user_embedding_dim=int(rng.choice([4, 16, 40])),
                    attention_size=int(rng.choice([8, 20, 40])), hidden_size=int(rng.choice([8, 20, 40, 64])))
        if kind == "sli_rec":
            over["hidden_size"] = D      # alpha mixes the A2SVD feature (D wide) with the encoder feature (H wide)
        if kind in ("sli_rec", "a2svd"):
            over["attention_size"] = D   # the A2SVD query is contracted with the projected history (base_model.py:622)
    cfg = dict(T=T, Di=D - Dc, Dc=Dc, Du=D, H=over.get("hidden_size", D), Vu=50, Vi=200, Vc=12)
    desc = "case %d %s: D=%d Dc=%d T=%d P=%d G=%d %s" % (idx, kind, D, Dc, T, P, G, over)
    one_case.desc = desc
    hp = build_hparams(cfg, P, **over)
    dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
    lengths = str(rng.choice(["full", "uniform", "lognormal"])) if T > 1 else "full"
    feed_seed, short = int(rng.integers(1 << 30)), rng.random() < 0.5
    only = os.environ.get("FUZZ_ONLY")   # "6,11": run these case numbers only (the random stream stays the same)
    if only and str(idx) not in only.split(","):
        return desc, None
    feed = synthetic_feed(P, T, dims["Vu"], dims["Vi"], dims["Vc"], G=G, lengths=lengths,
                          ids="uniform", seed=feed_seed)
    if short and T > 1:     # force some length-1 histories
        k = max(1, P // 3)
        rows = np.arange(k * G)
        for key in ("item_history", "item_cate_history", "mask", "time_diff", "time_from_first_action",
                    "time_to_now"):
            feed[key][rows, 1:] = 0
    problems = []
    for dedup in (True, False):
        if kind == "clsr":
            orc, extra = O, ()
            params32 = O.init_params(dims, hp, seed=idx, scale_dense=8.0)
            net = CLSRNet(hp, dims, device="cuda:0", seed=0, dedup_histories=dedup, precision=PRECISION)
        else:
            orc, extra = SO, (kind,)
            params32 = SO.init_params(dims, hp, kind, seed=idx, scale_dense=8.0)
            net = SeqNet(hp, dims, kind=kind, device="cuda:0", seed=0, dedup_histories=dedup)
        sd = dict(params32)
        sd.update(orc.init_bn_state(params32))
        net.load_state_dict(sd, strict=True)
        params = type(params32)((k, v.double()) for k, v in params32.items())
        tf = orc.to_torch_feed(feed, dtype=torch.float64)
        bn = orc.init_bn_state(params)
        # scoring (moving statistics)
        ev = orc.forward(params, bn, tf, hp, *extra, False)
        got_ev = net.forward(net.upload(feed, False), False)
        torch.cuda.synchronize()
        e = close(got_ev["logit"], ev["logit"], 1e-4, 1e-4)
        if e:
            problems.append("dedup=%s eval logit: %s" % (dedup, e))
        adam = orc.init_adam(params)
        new_p, new_bn, _, ls, grads, _, out = orc.train_step(params, bn, adam, 1, tf, hp, *extra)
        if os.environ.get("FUZZ_WARM"):  # diagnosis: allocate every workspace in a throw-away pass first (a deviation
            net.train_step(net.upload(feed, True), apply=False)   # that disappears is a first-step allocation race)
            torch.cuda.synchronize()
        net.capture_grads = True
        got = net.train_step(net.upload(feed, True))
        torch.cuda.synchronize()
        e = close(got["logit"], out["logit"], 1e-4, 1e-4)
        if e:
            problems.append("dedup=%s train logit: %s" % (dedup, e))
        if os.environ.get("FUZZ_DEBUG") and kind == "clsr":   # gradients of the INTERMEDIATE tensors against autograd
            from collections import OrderedDict
            leaf = OrderedDict((k, v.detach().clone().requires_grad_(not k.endswith("/user_embedding")))
                               for k, v in params.items())
            o2 = O.forward(leaf, bn, tf, hp, True, OrderedDict(), {})
            keys = [k for k in ("att_fea_long", "att_fea_short", "short_intention", "causal_state", "target", "rnn_out",
                                "hist_input") if k in o2 and getattr(o2[k], "requires_grad", False)]
            for k in keys:
                o2[k].retain_grad()
            O.losses(leaf, o2, tf, hp)["loss"].backward()
            B_, T_, G_, Hn_ = net.last_shape
            D_, H_, Du_ = net.D, net.H, net.Du
            zp = [v for k, v in net._bufs.items() if k[0] == "zero_pool"][-1].detach().double().cpu()
            off = [0]

            def take(*shape):
                n = int(np.prod(shape))
                t = zp[off[0]:off[0] + n].view(*shape)
                off[0] += n
                return t
            dhist, drnn, dhist_lt = take(Hn_, T_, D_), take(Hn_, T_, H_), take(Hn_, T_, D_)
            dtarget, dS = take(B_, D_), take(B_, D_)
            dL, dM, dR = take(Hn_, D_), take(Hn_, D_), take(Hn_, D_)
            dfs, dsi = take(Hn_, H_), take(Hn_, Du_)
            for k in ("alpha", "att_fea_long", "att_fea_short", "model_output", "causal_state", "short_intention",
                      "rnn_out", "logit", "w_long", "w_short"):
                if k in got and k in o2 and got[k] is not None:
                    g_, e_ = got[k].detach().double().cpu(), o2[k].detach().double()
                    if g_.numel() != e_.numel() and e_.shape[0] == B_ and g_.shape[0] == Hn_:
                        e_ = e_.reshape(Hn_, B_ // Hn_, *e_.shape[1:])[:, 0]
                    if g_.numel() == e_.numel():
                        print("    [debug dedup=%s] fwd %-16s max abs err %.3e  (max |exp| %.3e)" % (
                            dedup, k, float((g_.reshape(-1) - e_.reshape(-1)).abs().max()), float(e_.abs().max())))
            grp = lambda t: t.reshape(Hn_, B_ // Hn_, *t.shape[1:]).sum(1) if t.shape[0] == B_ and Hn_ != B_ else t
            for nm, got_t, key in (("dS", dS, "att_fea_short"), ("dL", dL, "att_fea_long"), ("dsi", dsi, "short_intention"),
                                   ("dfs", dfs, "causal_state"), ("dtarget", dtarget, "target"), ("drnn", drnn, "rnn_out")):
                if key not in keys:
                    continue
                ex = o2[key].grad.double()
                ex = ex if ex.shape == got_t.shape else grp(ex)
                err = float((got_t - ex).abs().max())
                print("    [debug dedup=%s] %-8s max abs err %.3e  (max |exp| %.3e)" % (dedup, nm, err, float(ex.abs().max())))
                if os.environ.get("FUZZ_DEBUG") == "2":
                    rows = (got_t - ex).abs().reshape(got_t.shape[0], -1).max(1)[0]
                    print("        rows with err > 10%% of max:", (rows > 0.1 * rows.max()).nonzero().reshape(-1).tolist()[:40],
                          "of", got_t.shape[0])
        gl = net.read_losses()
        for k in ("loss", "data_loss", "regular_loss") + (("contrastive_loss", "discrepancy_loss")
                                                            if kind == "clsr" else ()):
            e = close([gl[k]], [float(ls[k])], 1e-5, 1e-6)
            if e:
                problems.append("dedup=%s %s: %s" % (dedup, k, e))
        if P == 1:
            # one positive: its in-batch "negatives" are the same item, every row of the batch is identical, every
            # batch-norm output sits exactly on the ReLU kink (y = beta = 0) and the oracle's float64 masks are
            # decided by rounding noise -- gradients are not comparable there; values and losses are
            continue
        raw = out["raw_grads"]
        raw32 = None
        if os.environ.get("FUZZ_F32"):   # conditioning check: how far is a float32 run of the ORACLE from its float64 run?
            p32 = type(params32)((k, v.float()) for k, v in params32.items())
            raw32 = orc.train_step(p32, orc.init_bn_state(p32), orc.init_adam(p32), 1,
                                   orc.to_torch_feed(feed, dtype=torch.float32), hp, *extra)[6]["raw_grads"]
        floor = 4e-6 * max(float(raw[n].abs().max()) for n in net.dense_names)
        for name in net.dense_names:
            scale = float(raw[name].abs().max()) + 1e-12
            e = close(net.captured["dense"][name], raw[name], 2e-3, 2e-4 * scale + floor)
            if e:
                problems.append("dedup=%s grad %s: %s" % (dedup, name, e))
                if raw32 is not None:
                    problems.append("      float32 oracle vs float64 oracle: max abs err %.3e" %
                                    float((raw32[name].double() - raw[name].double()).abs().max()))
                if os.environ.get("FUZZ_TRACE"):
                    gt, ex = net.captured["dense"][name].double().cpu(), raw[name].double().cpu()
                    badm = (gt - ex).abs() > (2e-3 * ex.abs() + 2e-4 * scale + floor)
                    idx_bad = badm.nonzero()
                    print(name, tuple(ex.shape), "mismatches", int(badm.sum()), "first", idx_bad[:4].tolist(), "last",
                          idx_bad[-2:].tolist())
                    for ij in idx_bad[:4].tolist():
                        print("   ", ij, "got", float(gt[tuple(ij)]), "exp", float(ex[tuple(ij)]))
        # the applied update (clip + Adam, dense variables) and the batch-norm moving statistics
        sd = net.state_dict()
        for name in net.dense_names:
            g_ = grads[name].double().reshape(-1)
            sel = g_.abs() > 100 * floor      # Adam's first step is ~lr * sign(g): skip gradients at noise level
            if int(sel.sum()):
                e = close((sd[name].double().cpu().reshape(-1) - params[name].reshape(-1))[sel],
                          (new_p[name].reshape(-1) - params[name].reshape(-1))[sel], 5e-3, 0.02 * hp.learning_rate)
                if e:
                    problems.append("dedup=%s adam update %s: %s" % (dedup, name, e))
        for k, v in new_bn.items():
            e = close(sd[k], v, 1e-4, 1e-6)
            if e:
                problems.append("dedup=%s %s: %s" % (dedup, k, e))
    return desc, problems


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["clsr"]
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n):
        try:
            desc, problems = one_case(rng, i, kinds[i % len(kinds)])
        except Exception as exc:
            bad += 1
            print("CRASH " + getattr(one_case, "desc", "case %d" % i))
            print("    " + traceback.format_exc().strip().splitlines()[-1][:400])
            if os.environ.get("FUZZ_TRACE"):
                traceback.print_exc()
            continue
        if problems is None:
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

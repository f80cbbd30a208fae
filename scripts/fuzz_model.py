#!/usr/bin/env python
"""Model-level fuzzing through the reference API (test infrastructure): random batch sizes / negatives per positive /
history lengths / file sizes, so that epochs end in ragged batches and evaluation batches cut groups apart.

Two models with the same seed and the same ``random`` stream are trained on the same files: (a) the product defaults
(history de-duplication, compact feeds, launch plans, look-ahead staging) and (b) the plain configuration (row by row,
no plans).  Their per-epoch losses and predictions must agree (learning rate 1e-5: Adam turns
float-noise-level gradients into full-size steps, at a normal rate the two trajectories drift apart chaotically).

    python scripts/fuzz_model.py [n_cases] [seed]
"""
import os
import random
import sys
import tempfile
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import clsr_amd.clsr as M  # noqa: E402
from clsr_amd.deeprec_utils import prepare_hparams  # noqa: E402
from clsr_amd.sequential_iterator import SASequentialIterator, SequentialIterator  # noqa: E402
from clsr_amd.synthetic import make_tsv_dataset  # noqa: E402

MODELS = {"clsr": ("CLSRModel", "clsr.yaml", SASequentialIterator), "din": ("DINModel", "din.yaml", SequentialIterator),
          "sli_rec": ("SLI_RECModel", "sli_rec.yaml", SequentialIterator),
          "gru4rec": ("GRU4RecModel", "gru4rec.yaml", SequentialIterator)}


def run(kind, hp_kw, paths, plain, seed, valid_ngs, test_ngs, tmp):
    cls_name, yaml_name, it = MODELS[kind]
    hp = prepare_hparams(os.path.join(ROOT, "clsr_amd", "config", yaml_name), **hp_kw)
    kw = dict(dedup_histories=False) if plain else {}
    random.seed(seed)
    model = getattr(M, cls_name)(hp, it, seed=3, **kw)
    if plain:
        model.net.use_plans = False
    losses = []
    orig = model.batch_train

    def spy(file_iterator, train_sess):
        loss = orig(file_iterator, train_sess)
        losses.append((0, float(loss)))
        return loss
    model.batch_train = spy
    model.fit(paths["train_data"], paths["valid_data"], valid_num_ngs=valid_ngs, eval_metric="group_auc")
    res = model.run_weighted_eval(paths["test_data"], num_ngs=test_ngs)
    out = os.path.join(tmp, "pred_%d.txt" % int(plain))
    model.predict(paths["test_data"], out)
    preds = np.array([float(x) for x in open(out).read().split()])
    return losses, res, preds


def one_case(rng, idx, tmp):
    kind = str(rng.choice(list(MODELS)))
    G = int(rng.choice([2, 3, 5]))
    valid_ngs, test_ngs = int(rng.choice([1, 4])), int(rng.choice([3, 9, 19]))
    bs = int(rng.choice([7, 16, 50, 64, 100, 128, 333]))
    T = int(rng.choice([3, 10, 20]))
    n_train = int(rng.integers(60, 900))
    desc = "case %d %s: batch %d G=%d T=%d n_train=%d valid_ngs=%d test_ngs=%d" % (idx, kind, bs, G, T, n_train,
                                                                                 valid_ngs, test_ngs)
    one_case.desc = desc
    d = os.path.join(tmp, "c%d" % idx)
    paths = make_tsv_dataset(d, n_users=80, n_items=300, n_cates=9, n_train=n_train, n_valid=int(rng.integers(5, 60)),
                             n_test=int(rng.integers(5, 60)), valid_ngs=valid_ngs, test_ngs=test_ngs,
                             max_hist=int(rng.choice([4, 12, 30])), seed=int(rng.integers(1 << 30)), signal=True)
    hp_kw = dict(user_vocab=paths["user_vocab"], item_vocab=paths["item_vocab"], cate_vocab=paths["category_vocab"],
                 max_seq_length=T, batch_size=bs, train_num_ngs=G - 1, time_unit="s", is_clip_norm=1, embed_l2=1e-6,
                 layer_l2=1e-6, learning_rate=1e-5, show_step=10 ** 9, save_model=False, MODEL_DIR=None,
                 SUMMARIES_DIR=None, write_tfevents=False, epochs=2, EARLY_STOP=10)
    if kind == "clsr":
        hp_kw.update(contrastive_loss="triplet", contrastive_length_threshold=2, discrepancy_loss_weight=0.01,
                     contrastive_loss_weight=0.1)
    seed = int(rng.integers(1 << 30))
    la, ra, pa = run(kind, hp_kw, paths, False, seed, valid_ngs, test_ngs, tmp)
    lb, rb, pb = run(kind, hp_kw, paths, True, seed, valid_ngs, test_ngs, tmp)
    problems = []
    if [s for s, _ in la] != [s for s, _ in lb]:
        problems.append("steps per epoch %s vs %s" % (la, lb))
    for (_, x), (_, y) in zip(la, lb):
        if abs(x - y) > 2e-3 * max(1.0, abs(y)):
            problems.append("epoch loss %.6f vs %.6f" % (x, y))
    if pa.shape != pb.shape or pa.shape[0] != sum(1 for _ in open(paths["test_data"])):
        problems.append("prediction count %s vs %s" % (pa.shape, pb.shape))
    elif float(np.abs(pa - pb).max()) > 2e-3:
        problems.append("predictions differ by %.3e" % float(np.abs(pa - pb).max()))
    if sorted(ra) != sorted(rb):
        problems.append("metric keys %s vs %s" % (sorted(ra), sorted(rb)))
    # (the ranking metrics of a barely trained model on a few dozen groups flip with the last digit of a prediction:
    # the predictions themselves are what is compared)
    return desc, problems


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(n):
            try:
                desc, problems = one_case(rng, i, tmp)
            except Exception:
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
                print("ok   " + desc)
    print("%d of %d cases with problems" % (bad, n))


if __name__ == "__main__":
    main()

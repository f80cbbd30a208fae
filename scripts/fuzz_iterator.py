#!/usr/bin/env python
"""Fuzz the host iterators against the REFERENCE iterators (build container only: needs /root/reference; the
reference is imported with the TensorFlow stub of scripts/make_golden.py, nothing of it is copied or travels).

Random files (sizes, history lengths, out-of-vocabulary rates), batch sizes, max_seq_length, time units, negatives
per positive and min_seq_length; every feed of an epoch of training (same ``random`` seed) and of an evaluation pass
is compared array by array (values AND dtypes) with the reference's.

    python scripts/fuzz_iterator.py [n_cases] [seed]
"""
import os
import random
import sys
import tempfile
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def feeds_of(it, keys, path, ngs, min_len, seed):
    random.seed(seed)
    out = []
    for feed in it.load_data_from_file(path, batch_num_ngs=ngs, min_seq_length=min_len):
        if not feed:
            out.append(None)
            continue
        out.append({keys[ph]: np.asarray(val) for ph, val in feed.items()})
    return out, random.random()      # the generator state after the epoch must agree too


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import make_golden

    sys.path.insert(0, make_golden.REF)
    make_golden.install_tf_stub()
    warnings.simplefilter("ignore")
    from reco_utils.recommender.deeprec.io import sequential_iterator as ref_it
    from clsr_amd import sequential_iterator as our_it
    from clsr_amd.synthetic import make_tsv_dataset

    rng = np.random.default_rng(seed)
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(n):
            kw = dict(n_users=int(rng.integers(5, 120)), n_items=int(rng.integers(20, 600)), n_cates=int(rng.integers(3, 30)),
                      n_train=int(rng.integers(1, 700)), n_valid=int(rng.integers(1, 40)), n_test=int(rng.integers(1, 40)),
                      valid_ngs=int(rng.choice([1, 4])), test_ngs=int(rng.choice([1, 9, 49])),
                      max_hist=int(rng.choice([1, 3, 14, 60])), seed=int(rng.integers(1 << 30)))
            paths = make_tsv_dataset(os.path.join(tmp, "c%d" % i), **kw)

            class HP(object):
                user_vocab, item_vocab, cate_vocab = paths["user_vocab"], paths["item_vocab"], paths["category_vocab"]
                max_seq_length = int(rng.choice([1, 2, 5, 10, 50]))
                batch_size = int(rng.choice([1, 3, 16, 64, 100, 1000]))
                time_unit = str(rng.choice(["s", "ms"]))
            ngs, min_len = int(rng.choice([1, 2, 4, 9])), int(rng.choice([1, 1, 2, 5]))
            sa = bool(rng.random() < 0.6)
            cname = "SASequentialIterator" if sa else "SequentialIterator"
            desc = "case %d %s %s batch %d T %d unit %s ngs %d min_len %d" % (
                i, cname, {k: kw[k] for k in ("n_train", "max_hist")}, HP.batch_size, HP.max_seq_length, HP.time_unit, ngs,
                min_len)
            ref = getattr(ref_it, cname)(HP, ref_it.tf.Graph())
            ours = getattr(our_it, cname)(HP, None)
            rkeys = {v: k for k, v in ref.__dict__.items() if isinstance(v, ref_it.tf.placeholder)}
            okeys = {getattr(ours, k): k for k in rkeys.values()}
            problems = []
            s = int(rng.integers(1 << 30))
            for path, g in ((paths["train_data"], ngs), (paths["valid_data"], 0), (paths["test_data"], 0)):
                (fa, ra), (fb, rb) = feeds_of(ref, rkeys, path, g, min_len, s), feeds_of(ours, okeys, path, g, min_len, s)
                if ra != rb:
                    problems.append("%s: random stream position differs after the pass" % os.path.basename(path))
                if len(fa) != len(fb):
                    problems.append("%s: %d vs %d feeds" % (os.path.basename(path), len(fa), len(fb)))
                    continue
                for b, (x, y) in enumerate(zip(fa, fb)):
                    if (x is None) != (y is None):
                        problems.append("%s batch %d: dropped on one side only" % (os.path.basename(path), b))
                        continue
                    if x is None:
                        continue
                    for k in x:
                        yv = np.asarray(y[k])
                        if x[k].dtype != yv.dtype or x[k].shape != yv.shape or not np.array_equal(x[k], yv):
                            problems.append("%s batch %d %s: %s%s vs %s%s" % (os.path.basename(path), b, k, x[k].dtype,
                                                                             x[k].shape, yv.dtype, yv.shape))
            if problems:
                bad += 1
                print("FAIL " + desc)
                for p in problems[:6]:
                    print("    " + p)
            else:
                print("ok   " + desc)
    print("%d of %d cases with problems" % (bad, n))

    # ---- metrics: the vectorised paths (2-D groups, segmented per-user AUC) against the reference's loops
    from reco_utils.recommender.deeprec import deeprec_utils as ref_utils
    from clsr_amd import deeprec_utils as our_utils

    badm = 0
    for i in range(n):
        groups, size = int(rng.integers(1, 80)), int(rng.choice([2, 5, 10, 100]))
        labels = np.zeros((groups, size), dtype=np.float32)
        labels[:, 0] = 1.0
        preds = rng.random((groups, size)).astype(np.float32)
        if rng.random() < 0.7:      # ties, saturated scores
            preds = np.round(preds, int(rng.choice([1, 2])))
        users = rng.integers(1, max(2, groups // 3 + 1), size=groups).repeat(size).astype(np.float32)
        fl, fp = labels.reshape(-1).tolist(), preds.reshape(-1).tolist()
        pair = ["mean_mrr", "ndcg@2;4;6", "hit@2;4;6", "group_auc"]
        exp = {}
        exp.update(ref_utils.cal_metric(fl, fp, ["auc", "logloss"]))
        exp.update(ref_utils.cal_metric(list(labels), list(preds), pair))
        exp.update(ref_utils.cal_weighted_metric(users.tolist(), fp, fl, ["wauc"]))
        got = {}
        got.update(our_utils.cal_metric(fl, fp, ["auc", "logloss"]))
        got.update(our_utils.cal_metric(labels, preds, pair))                  # vectorised 2-D path
        got.update(our_utils.cal_weighted_metric(users.tolist(), fp, fl, ["wauc"]))
        loop = our_utils.cal_metric(list(labels), list(preds), pair)           # per-group path
        diff = {k: (got.get(k), exp[k]) for k in exp if got.get(k) != exp[k]}
        diff.update({"loop:" + k: (loop.get(k), exp[k]) for k in loop if loop[k] != exp[k]})
        if diff:
            badm += 1
            print("FAIL metrics case %d (%d groups of %d): %s" % (i, groups, size, diff))
    print("%d of %d metric cases with problems" % (badm, n))


if __name__ == "__main__":
    main()

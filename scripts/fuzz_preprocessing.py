#!/usr/bin/env python
"""Fuzz clsr_amd.sequential_reviews.data_preprocessing against the REFERENCE's (build container only, like
make_golden_preprocessing.py): random raw logs (users, items, events per user), sample rates, negatives per positive,
history expansion on / off, same ``random`` / ``numpy.random`` seeds -- every output file must be byte-identical and
every vocabulary equal.

    python scripts/fuzz_preprocessing.py [n_cases] [seed]
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


def run(fn, raw, d, dataset, rate, vn, tn, expand, seed):
    os.makedirs(d)
    p = {k: os.path.join(d, k) for k in ("train_data", "valid_data", "test_data")}
    v = {k: os.path.join(d, k + ".pkl") for k in ("user_vocab", "item_vocab", "category_vocab")}
    random.seed(seed)
    np.random.seed(seed)
    fn(raw, os.path.join(d, ""), p["train_data"], p["valid_data"], p["test_data"], v["user_vocab"], v["item_vocab"],
       v["category_vocab"], sample_rate=rate, valid_num_ngs=vn, test_num_ngs=tn, dataset=dataset,
       is_history_expanding=expand)
    out = {k: open(path, "rb").read() for k, path in p.items()}
    out.update({k: open(path, "rb").read() for k, path in v.items()})
    return out, (random.random(), float(np.random.random()))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    from make_golden import REF, install_tf_stub

    sys.path.insert(0, REF)
    install_tf_stub()
    warnings.simplefilter("ignore")
    from reco_utils.dataset import sequential_reviews as ref
    from clsr_amd import sequential_reviews as ours
    from clsr_amd.synthetic import make_raw_kuaishou_csv, make_raw_taobao_csv

    rng = np.random.default_rng(seed)
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(n):
            dataset = str(rng.choice(["taobao", "kuaishou"]))
            rate, vn, tn = float(rng.choice([1.0, 0.7, 0.3])), int(rng.choice([1, 4])), int(rng.choice([1, 9, 19]))
            expand, s = bool(rng.random() < 0.7), int(rng.integers(1 << 30))
            d = os.path.join(tmp, "c%d" % i)
            os.makedirs(d)
            if dataset == "taobao":
                raw = os.path.join(d, "UserBehavior.csv")
                kw = dict(n_users=int(rng.integers(400, 2500)), n_items=int(rng.integers(40, 200)),
                          n_cates=int(rng.integers(3, 25)), events_per_user=int(rng.integers(20, 50)))
                make_raw_taobao_csv(raw, seed=int(rng.integers(1 << 30)), **kw)
            else:
                raw = os.path.join(d, "kuaishou.csv")
                kw = dict(n_users=int(rng.integers(40, 400)), n_items=int(rng.integers(40, 300)),
                          n_clusters=int(rng.integers(2, 20)), events_per_user=int(rng.integers(25, 70)))
                make_raw_kuaishou_csv(raw, seed=int(rng.integers(1 << 30)), **kw)
            desc = "case %d %s %s rate %.1f ngs %d/%d expand %s" % (i, dataset, kw, rate, vn, tn, expand)
            try:
                a, ra = run(ref.data_preprocessing, raw, os.path.join(d, "ref"), dataset, rate, vn, tn, expand, s)
            except Exception as e:        # logs the reference itself cannot process (e.g. nothing survives its filters)
                try:
                    run(ours.data_preprocessing, raw, os.path.join(d, "ours"), dataset, rate, vn, tn, expand, s)
                    print("FAIL " + desc + ": the reference raised %s, this build did not" % type(e).__name__)
                    bad += 1
                except Exception as e2:
                    same = type(e) is type(e2)
                    bad += 0 if same else 1
                    print(("ok   " if same else "FAIL ") + desc + " (both raise: %s / %s)" % (type(e).__name__,
                                                                                                type(e2).__name__))
                continue
            b, rb = run(ours.data_preprocessing, raw, os.path.join(d, "ours"), dataset, rate, vn, tn, expand, s)
            diff = [k for k in a if a[k] != b[k]]
            if ra != rb:
                diff.append("random streams")
            if diff:
                bad += 1
                print("FAIL " + desc + ": " + ", ".join(diff))
            else:
                print("ok   " + desc + " (%d train lines)" % a["train_data"].count(b"\n"))
    print("%d of %d cases with problems" % (bad, n))


if __name__ == "__main__":
    main()

"""Synthetic Taobao/Kuaishou-shaped inputs.

Two products:

* :func:`make_tsv_dataset` -- train/valid/test TSV + the three vocab pickles in the
  on-disk format the reference's iterator reads (``io/sequential_iterator.py:90-104``;
  vocab id 0 = ``default_*``, ``dataset/sequential_reviews.py:118-140``), used for
  end-to-end ``fit``/``run_weighted_eval``/``predict`` runs (BASELINE config 1).
* :func:`synthetic_feed` -- one training feed generated directly as arrays in the
  iterator's layout (bypassing file parsing), used by ``bench.py`` and the
  full-size GPU tests (BASELINE configs 2, 3, 5).  Seeded ``default_rng(20220425)``.
"""
import os
import pickle as pkl

import numpy as np

__all__ = ["CONFIGS", "make_tsv_dataset", "synthetic_feed", "zipf_sampler"]

SEED = 20220425

# name -> (Vu, Vi, Vc, Di, Dc, Du, H, T, P)
CONFIGS = {
    "plumbing": dict(Vu=200, Vi=1000, Vc=20, Di=32, Dc=8, Du=40, H=40, T=10, P=64),
    "taobao": dict(Vu=36915, Vi=64138, Vc=4096, Di=32, Dc=8, Du=40, H=40, T=50, P=4096),
    "kuaishou": dict(Vu=60813, Vi=292286, Vc=1000, Di=32, Dc=8, Du=40, H=40, T=250, P=1024),
    "catalogue100m": dict(Vu=1000000, Vi=100000000, Vc=10000, Di=96, Dc=32, Du=128, H=128,
                          T=50, P=4096),
}


def zipf_sampler(rng, n, s=1.05):
    """Return ``draw(size)`` sampling ids in ``[1, n)`` with a Zipf(s) popularity law."""
    ranks = np.arange(1, n, dtype=np.float64)
    cdf = np.cumsum(ranks ** (-s))
    cdf /= cdf[-1]

    def draw(size):
        return (np.searchsorted(cdf, rng.random(size)) + 1).astype(np.int64)

    return draw


def _lengths(rng, n, T, mode):
    if mode == "full":
        return np.full(n, T, dtype=np.int64)
    if mode == "lognormal":
        return np.clip(np.round(rng.lognormal(3.4, 0.6, n)), 1, T).astype(np.int64)
    if mode == "uniform":
        return rng.integers(1, T + 1, n)
    if mode == "uniform_long":
        return rng.integers(max(1, T // 5), T + 1, n)
    raise ValueError(mode)


def synthetic_feed(P, T, Vu, Vi, Vc, G=5, lengths="lognormal", ids="zipf", seed=SEED,
                   item2cate=None):
    """One training feed (``B = P*G`` rows, positive first in every group of ``G``).

    Layout and dtypes follow ``SASequentialIterator._convert_data`` in training mode
    (ref ``io/sequential_iterator.py:551-655``): histories left-aligned / zero padded,
    time features ``log(U(0.5, 1e3))`` on valid steps, negatives are the target
    (item, cate) of another line of the batch with a different item.
    """
    rng = np.random.default_rng(seed)
    if item2cate is None:
        item2cate = rng.integers(1, Vc, size=min(Vi, 1 << 22)).astype(np.int32)
    draw = zipf_sampler(rng, Vi) if ids == "zipf" else (lambda size: rng.integers(1, Vi, size))
    lens = _lengths(rng, P, T, lengths)
    valid = np.arange(T)[None, :] < lens[:, None]
    item_hist = np.where(valid, draw((P, T)).reshape(P, T), 0).astype(np.int32)
    cate_hist = np.where(valid, item2cate[item_hist % item2cate.size], 0).astype(np.int32)
    tgt_items = draw(P).astype(np.int32)
    tgt_cates = item2cate[tgt_items % item2cate.size].astype(np.int32)
    users = rng.integers(1, Vu, size=P).astype(np.int32)

    def tfeat():
        return np.where(valid, np.log(rng.uniform(0.5, 1e3, size=(P, T))), 0.0).astype(np.float32)

    tdiff, tfirst, tnow = tfeat(), tfeat(), tfeat()
    # in-batch negatives (different item than the positive)
    src = np.empty((P, G), dtype=np.int64)
    src[:, 0] = np.arange(P)
    for g in range(1, G):
        cand = rng.integers(0, P, size=P)
        for _ in range(64):
            bad = tgt_items[cand] == tgt_items
            if not bad.any():
                break
            cand[bad] = rng.integers(0, P, size=int(bad.sum()))
        src[:, g] = cand
    flat = src.reshape(-1)
    rep = np.repeat(np.arange(P), G)
    labels = np.zeros((P, G), dtype=np.float32)
    labels[:, 0] = 1.0
    row_cates = tgt_cates[flat]
    same = ((cate_hist[rep] == row_cates[:, None]) & valid[rep]).sum(1)
    feed = {
        "labels": labels.reshape(-1, 1),
        "attn_labels": (same / lens[rep]).astype(np.float32).reshape(-1, 1),
        "users": users[rep],
        "items": tgt_items[flat],
        "cates": row_cates,
        "item_history": item_hist[rep],
        "item_cate_history": cate_hist[rep],
        "mask": valid[rep].astype(np.float32),
        "time": np.zeros(P * G, dtype=np.float32),
        "time_diff": tdiff[rep],
        "time_from_first_action": tfirst[rep],
        "time_to_now": tnow[rep],
    }
    return feed


def make_tsv_dataset(out_dir, n_users=200, n_items=1000, n_cates=20, n_train=600, n_valid=40,
                     n_test=40, valid_ngs=4, test_ngs=9, max_hist=14, seed=SEED, signal=False):
    """Write ``train_data / valid_data / test_data`` + vocab pickles under ``out_dir``.

    Train lines are positives only (negatives are sampled in-batch); valid/test hold one
    positive line followed by ``*_ngs`` negative lines sharing its history, like the
    reference's offline negative sampling (``dataset/sequential_reviews.py:142-199``).
    A small fraction of tokens is left out of the vocabularies to exercise the
    id-0 (``default_*``) path.  Returns a dict of the file paths.

    ``signal=True`` makes the task learnable (for end-to-end training tests): every user browses ONE
    category, the positive target comes from that category and the offline negatives from other ones.
    """
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    item2cate = rng.integers(0, n_cates, size=n_items)
    base_ts = 1511539200  # seconds; time_unit 's'

    def one_line(label, u, it, ts, hist_items, hist_ts):
        return "\t".join([
            str(label), "u%d" % u, "i%d" % it, "c%d" % item2cate[it], str(ts),
            ",".join("i%d" % h for h in hist_items),
            ",".join("c%d" % item2cate[h] for h in hist_items),
            ",".join(str(t) for t in hist_ts),
        ]) + "\n"

    by_cate = [np.flatnonzero(item2cate == c) for c in range(n_cates)] if signal else None

    def sample_positive():
        u = int(rng.integers(0, n_users))
        n = int(rng.integers(1, max_hist + 1))
        if signal:
            pool = by_cate[u % n_cates]
            if len(pool) < 2:
                pool = np.arange(n_items)
            hist = rng.choice(pool, size=n)
            it = int(rng.choice(pool))
        else:
            hist = rng.integers(0, n_items, size=n)
            it = int(rng.integers(0, n_items))
        gaps = rng.integers(1, 86400 * 3, size=n + 1)
        ts = base_ts + np.cumsum(gaps)
        return u, it, int(ts[-1]), [int(h) for h in hist], [int(t) for t in ts[:-1]]

    paths = {k: os.path.join(out_dir, k) for k in ("train_data", "valid_data", "test_data")}
    train_lines = []
    for _ in range(n_train):
        u, it, ts, h, hts = sample_positive()
        train_lines.append(one_line(1, u, it, ts, h, hts))
    with open(paths["train_data"], "w") as f:
        f.writelines(train_lines)
    for name, n_pos, ngs in (("valid_data", n_valid, valid_ngs), ("test_data", n_test, test_ngs)):
        with open(paths[name], "w") as f:
            for _ in range(n_pos):
                u, it, ts, h, hts = sample_positive()
                f.write(one_line(1, u, it, ts, h, hts))
                negs = set()
                while len(negs) < ngs:
                    cand = int(rng.integers(0, n_items))
                    if cand != it and not (signal and item2cate[cand] == item2cate[it]):
                        negs.add(cand)
                for neg in sorted(negs):
                    f.write(one_line(0, u, neg, ts, h, hts))

    # vocabularies from the train file only, frequency-sorted, id 0 = default
    counts = ({}, {}, {})
    for line in train_lines:
        w = line.strip("\n").split("\t")
        for d, toks in zip(counts, ([w[1]], [w[2]] + w[5].split(","), [w[3]] + w[6].split(","))):
            for t in toks:
                d[t] = d.get(t, 0) + 1
    for d, default, fname in zip(counts, ("default_uid", "default_mid", "default_cat"),
                                 ("user_vocab.pkl", "item_vocab.pkl", "category_vocab.pkl")):
        voc = {default: 0}
        for i, (k, _) in enumerate(sorted(d.items(), key=lambda kv: (-kv[1], kv[0]))):
            voc[k] = i + 1
        p = os.path.join(out_dir, fname)
        with open(p, "wb") as f:
            pkl.dump(voc, f, protocol=2)
        paths[fname.split(".")[0]] = p
    return paths


def make_raw_taobao_csv(path, n_users=1200, n_items=90, n_cates=12, events_per_user=34, seed=SEED):
    """A synthetic ``UserBehavior.csv`` (no header: uid,iid,category,behavior,ts) with everything the Taobao
    preprocessing has to cope with (reference ``dataset/sequential_reviews.py:955-982``): non-``pv`` behaviours,
    repeated (user, item) pairs, a few items listed under two categories, timestamps outside the
    2017-11-25 .. 2017-12-03 window, and enough users that the hard-coded 5 % user sample plus the 10-core filters
    leave a usable log.  Deterministic for a given seed; returns the number of lines written."""
    rng = np.random.default_rng(seed)
    item2cate = rng.integers(1000, 1000 + n_cates, size=n_items)
    two_cates = set(rng.choice(n_items, size=max(1, n_items // 15), replace=False).tolist())
    t0, t1 = 1511600000, 1512300000            # well inside the window in any time zone
    lines = []
    for u in range(n_users):
        n = int(rng.integers(events_per_user - 8, events_per_user + 9))
        items = rng.integers(0, n_items, size=n)
        ts = np.sort(rng.integers(t0, t1, size=n))
        beh = rng.choice(["pv", "pv", "pv", "pv", "buy", "cart", "fav"], size=n)
        for j in range(n):
            it = int(items[j])
            cate = int(item2cate[it])
            if it in two_cates and rng.random() < 0.3:
                cate += 500
            t = int(ts[j])
            if rng.random() < 0.03:
                t = t0 - 40 * 86400 if rng.random() < 0.5 else t1 + 40 * 86400
            lines.append("%d,%d,%d,%s,%d\n" % (100000 + u, 5000 + it, cate, beh[j], t))
    order = rng.permutation(len(lines))         # the real file is not grouped by user
    with open(path, "w") as f:
        f.writelines(lines[i] for i in order)
    return len(lines)


def make_raw_kuaishou_csv(path, n_users=150, n_items=120, n_clusters=9, events_per_user=45, seed=SEED):
    """A synthetic ``kuaishou.csv`` (header row; the columns the preprocessing reads are ``user_id, photo_id,
    time_ms, effective_view, photo_kmeans_cluster_id`` -- reference ``dataset/sequential_reviews.py:999-1041``)
    with negative feedback rows (``effective_view = 0``), repeated pairs and a two-day span in milliseconds."""
    rng = np.random.default_rng(seed + 1)
    cluster = rng.integers(0, n_clusters, size=n_items)
    t0 = 1600000000000
    rows = []
    for u in range(n_users):
        n = int(rng.integers(events_per_user - 10, events_per_user + 11))
        items = rng.integers(0, n_items, size=n)
        ts = np.sort(rng.integers(t0, t0 + 2 * 86400 * 1000, size=n))
        eff = (rng.random(n) < 0.7).astype(int)
        for j in range(n):
            rows.append((7000 + u, 300000 + int(items[j]), int(ts[j]), int(eff[j]), int(cluster[items[j]]),
                         int(rng.integers(0, 5))))
    order = rng.permutation(len(rows))
    with open(path, "w") as f:
        f.write("user_id,photo_id,time_ms,effective_view,photo_kmeans_cluster_id,tab\n")
        f.writelines("%d,%d,%d,%d,%d,%d\n" % rows[i] for i in order)
    return len(rows)

"""Raw behaviour log -> train / valid / test TSV + vocabularies: the on-disk format the iterator consumes.

Host-side mirror of the reference's ``reco_utils/dataset/sequential_reviews.py`` for the two datasets CLSR is
published on (``data_preprocessing(..., dataset='taobao' | 'kuaishou')``, reference ``:27-74``), so that
``examples/sequential.py`` can start from ``UserBehavior.csv`` / ``kuaishou.csv`` like the reference's quick-start
does (``examples/00_quick_start/sequential.py:342-343``).  Same signature, same output files, and -- because the
reference draws from the global ``random`` / ``numpy.random`` streams -- the same draws in the same order, so a
seeded run produces byte-identical files (pinned by tests/golden/preprocessing_golden.json, generated from the
reference by scripts/make_golden_preprocessing.py).

What differs is the route: the reference round-trips five intermediate text files through python line loops; here
the log stays in memory as columns (pandas for the relational filters, numpy for the global time split and the
history-expansion bookkeeping) and only the six outputs are written.

Stages (reference file:line):
  taobao / kuaishou filters  :955-982 / :999-1041 (``filter_k_core`` :815-828, ``filter_k_core_consider_neg``
                             :830-843, ``filter_items_with_multiple_cids`` :936-943, ``downsample`` :946-952)
  instances                  ``_create_instance`` :592-630  (users in first-appearance order, each sorted by time)
  item sampling              ``_get_sampled_data`` :537-556, ``_create_item2cate`` :526-534
  global time split          ``_data_processing_taobao`` :705-735 (last day / day before),
                             ``_data_processing_ks`` :672-703 (last 12 h / 12 h before)
  history expansion          ``_data_generating_global`` :358-438, ``_data_generating_ks`` :275-355,
                             ``_data_generating_no_history_expanding`` :441-523
  vocabularies               ``_create_vocab`` :77-144
  offline negatives          ``_negative_sampling_offline`` :147-199
"""
import os
import pickle
import random
from datetime import datetime

import numpy as np
import pandas as pd

__all__ = ["data_preprocessing"]


# ------------------------------------------------------------------------------------------ raw log -> filtered log
def _k_core(log, k, key, counted):
    """Rows whose ``key`` has at least ``k`` (non-null) ``counted`` values; one pass, row order kept."""
    return log[log.groupby(key)[counted].transform("count") >= k]


def _taobao_log(reviews_file):
    """``pv`` events, first occurrence of every (user, item), items with a single category, the nine days from
    2017-11-25 (local time, like ``datetime.timestamp`` of a naive datetime), a 5 % sample of the users drawn from
    ``numpy.random`` (``Series.sample``), then the item and user 10-core filters (one pass each)."""
    log = pd.read_csv(reviews_file, header=None, names=["uid", "iid", "category", "behavior", "ts"])
    log = log[log["behavior"] == "pv"].drop_duplicates(subset=["uid", "iid"])
    n_cates = log[["iid", "category"]].drop_duplicates().groupby("iid")["category"].count()
    log = log[log["iid"].isin(n_cates.index[n_cates == 1])]
    lo = int(datetime.timestamp(datetime(2017, 11, 25, 0, 0, 0)))
    hi = int(datetime.timestamp(datetime(2017, 12, 3, 23, 59, 59)))
    log = log[(log["ts"] >= lo) & (log["ts"] <= hi)]
    users = log["uid"].drop_duplicates().sample(frac=0.05)
    log = log[log["uid"].isin(users)]
    log = _k_core(log, 10, "iid", "uid")
    log = _k_core(log, 10, "uid", "iid")
    return log


def _kuaishou_log(reviews_file):
    """First occurrence of every (user, photo), photos seen by >= 10 users, users with >= 10 effective views,
    then the effective views only."""
    log = pd.read_csv(reviews_file, header=0).rename(columns={
        "time_ms": "ts", "user_id": "uid", "photo_id": "iid", "photo_kmeans_cluster_id": "category"})
    log = log.drop_duplicates(subset=["uid", "iid"])
    log = _k_core(log, 10, "iid", "uid")
    positives = log[log["effective_view"] == 1].groupby("uid")["iid"].count()
    log = log[log["uid"].isin(positives.index[positives >= 10])]
    return log[log["effective_view"] == 1]


def _text(values):
    """A column as the text pandas writes for it and reads back (the reference round-trips every stage through
    ``to_csv`` / ``read_csv``): integers in canonical form, everything else as is."""
    arr = np.asarray(values)
    if arr.dtype.kind in "iu":
        return arr.astype(str).astype(object)
    tokens = np.asarray([str(v) for v in arr], dtype=object)
    try:
        return np.asarray([int(t) for t in tokens], dtype=np.int64).astype(str).astype(object)
    except ValueError:
        return tokens


def _typed(tokens):
    """The python values ``read_csv`` would hand back for a text column: ints when every token is one."""
    try:
        return [int(t) for t in tokens]
    except ValueError:
        return list(tokens)


# ------------------------------------------------------------------------------------------ instances
def _instances(log):
    """Columns (user, item, ts, cate) as text, users in order of first appearance and each user's events by
    ascending time (stable), every item with the first category listed for it (``default_cat`` if none)."""
    user, item, ts = _text(log["uid"]), _text(log["iid"]), _text(log["ts"])
    meta = log[["iid", "category"]].drop_duplicates()
    first_cate = {}
    for i, c in zip(_text(meta["iid"]), _text(meta["category"])):
        first_cate.setdefault(i, c)
    codes, _ = pd.factorize(user)
    order = np.lexsort((ts.astype(np.float64), codes))
    user, item, ts = user[order], item[order], ts[order]
    cate = np.asarray([first_cate.get(i, "default_cat") for i in item], dtype=object)
    return user, item, ts, cate


def _sample_items(inst, sample_rate):
    """Keep the events of ``int(n_items * rate)`` distinct items drawn popularity-weighted with ``random.choice``
    (rejection of repeats, like the reference -- same stream, same draws)."""
    user, item, ts, cate = inst
    pool = _typed(item)
    if sample_rate < 1:
        want = int(len(set(pool)) * sample_rate)
        chosen = set()
        while len(chosen) < want:
            chosen.add(random.choice(pool))
        keep = np.fromiter((p in chosen for p in pool), dtype=bool, count=len(pool))
        user, item, ts, cate = user[keep], item[keep], ts[keep], cate[keep]
        pool = [p for p, k in zip(pool, keep) if k]
    return (user, item, ts, cate), pool


# ------------------------------------------------------------------------------------------ split + expansion
_TRAIN, _VALID, _TEST = 0, 1, 2


def _global_time_split(ts, interval):
    """0 / 1 / 2 = train / valid / test: the last ``interval`` of the log is test, the one before it valid."""
    t = np.asarray([int(x) for x in ts], dtype=np.int64)
    if t.size == 0:     # the reference indexes the last element of an empty sorted list here (sequential_reviews.py:722)
        raise IndexError("list index out of range: no interaction survives the user sample and the 10-core filters")
    test_from = t.max() - interval
    valid_from = t.max() - 2 * interval
    return np.where(t < valid_from, _TRAIN, np.where(t < test_from, _VALID, _TEST))


def _line(label, user, item, cate, ts, items, cates, times):
    return "\t".join((label, user, item, cate, ts, ",".join(items), ",".join(cates), ",".join(times))) + "\n"


def _expand_histories(inst, split, keep_prob_draws, threshold, min_sequence=1):
    """One output line per event that has >= ``min_sequence`` earlier events of the same user (all of them as the
    history), thinned by the reference's ``round(uniform, 1) < threshold`` rule.  ``keep_prob_draws`` says which
    splits consume a draw from ``numpy.random`` (one per event of those splits, first events included)."""
    user, item, ts, cate = inst
    n = len(user)
    draws_for = np.isin(split, keep_prob_draws)
    prob = np.zeros(n)
    u = np.random.uniform(0, 1, size=int(draws_for.sum()))       # == that many successive scalar draws
    prob[draws_for] = [round(float(x), 1) for x in u]
    new_user = np.concatenate(([True], user[1:] != user[:-1]))
    start = np.maximum.accumulate(np.where(new_user, np.arange(n), 0))
    depth = np.arange(n) - start                                   # earlier events of the same user
    write = (~new_user) & (depth >= min_sequence) & (prob < threshold)
    out = ([], [], [])
    for i in np.flatnonzero(write):
        a = start[i]
        out[split[i]].append(_line("1", user[i], item[i], cate[i], ts[i], item[a:i], cate[a:i], ts[a:i]))
    return out


def _one_line_per_sequence(inst, split, min_sequence=1):
    """``is_history_expanding=False``: a line is emitted for the event BEFORE every boundary (user change, or the
    current event lying in valid / test), written to that previous event's split; the running history is only
    cleared when the boundary event is a training one, so it can carry over between users -- reference behaviour
    (``:441-523``), kept as is, as is the never-emitted last event."""
    user, item, ts, cate = inst
    out = ([], [], [])
    items, cates, times = [], [], []
    for i in range(len(user)):
        boundary = i == 0 or user[i] != user[i - 1] or split[i] != _TRAIN
        if boundary:
            if i > 0 and len(items) > min_sequence:
                out[split[i - 1]].append(_line("1", user[i - 1], item[i - 1], cate[i - 1], ts[i - 1],
                                               items[:-1], cates[:-1], times[:-1]))
            if split[i] == _TRAIN or i == 0:
                items, cates, times = [], [], []
        items.append(item[i])
        cates.append(cate[i])
        times.append(ts[i])
    return out


# ------------------------------------------------------------------------------------------ vocabularies
def _frequency_vocab(tokens, default):
    """``default`` -> 0, then tokens by descending count; ties keep first-appearance order (stable sort)."""
    codes, uniq = pd.factorize(np.asarray(tokens, dtype=object))
    counts = np.bincount(codes, minlength=len(uniq))
    voc = {} if default is None else {default: 0}
    base = len(voc)
    for rank, j in enumerate(np.argsort(-counts, kind="stable")):
        voc[uniq[j]] = base + rank
    return voc


def _create_vocab(train_lines, user_vocab, item_vocab, cate_vocab):
    users, items, cates = [], [], []
    for line in train_lines:
        w = line.strip("\n").split("\t")
        users.append(w[1])
        items.append(w[2])
        cates.append(w[3])
        if w[5]:
            items.extend(w[5].split(","))
            cates.extend(w[6].split(","))
    for tokens, default, path in ((users, "default_uid", user_vocab), (items, "default_mid", item_vocab),
                                  (cates, "default_cat", cate_vocab)):
        with open(path, "wb") as f:
            pickle.dump(_frequency_vocab(tokens, default), f)


# ------------------------------------------------------------------------------------------ offline negatives
def _with_negatives(lines, pool, item2cate, n_neg):
    """Every positive line followed by ``n_neg`` lines with distinct items drawn popularity-weighted
    (``random.choice`` over the event list).  As in the reference the candidate (a ``read_csv`` value: an int for
    numeric ids) is compared with the positive's TEXT, so with numeric ids the positive itself is not excluded."""
    out = []
    for line in lines:
        out.append(line)
        words = line.strip().split("\t")
        positive = words[2]
        seen = set()
        while len(seen) < n_neg:
            cand = random.choice(pool)
            if cand == positive or cand in seen:
                continue
            seen.add(cand)
            words[0], words[2], words[3] = "0", str(cand), str(item2cate[cand])
            out.append("\t".join(words) + "\n")
    return out


# ------------------------------------------------------------------------------------------ entry point
def data_preprocessing(reviews_file, meta_file, train_file, valid_file, test_file, user_vocab, item_vocab,
                       cate_vocab, sample_rate=0.01, valid_num_ngs=4, test_num_ngs=9, dataset="taobao",
                       is_history_expanding=True):
    """Create the training, validation and test files and the three vocabularies from the raw log
    (reference ``:27-74``; ``meta_file`` is unused for these two datasets there too).

    Consumes the global ``numpy.random`` stream (user sample, expansion thinning) and the global ``random``
    stream (item sample, offline negatives) exactly like the reference; seed both for reproducible files."""
    if dataset == "taobao":
        log = _taobao_log(reviews_file)
        interval, draws, threshold = 24 * 60 * 60, (_VALID, _TEST), 0.2
    elif dataset == "kuaishou":
        log = _kuaishou_log(reviews_file)
        interval, draws, threshold = 12 * 60 * 60 * 1000, (_TRAIN, _VALID, _TEST), 0.1
    else:
        raise ValueError("data_preprocessing supports dataset='taobao' or 'kuaishou', got %r" % (dataset,))
    inst = _instances(log)
    typed_items = _typed(inst[1])
    cate_values = _typed(inst[3])
    item2cate = dict(zip(typed_items, cate_values))
    inst, pool = _sample_items(inst, sample_rate)
    split = _global_time_split(inst[2], interval)
    if is_history_expanding:
        train, valid, test = _expand_histories(inst, split, draws, threshold)
    else:
        train, valid, test = _one_line_per_sequence(inst, split)
    _create_vocab(train, user_vocab, item_vocab, cate_vocab)
    valid = _with_negatives(valid, pool, item2cate, valid_num_ngs)
    test = _with_negatives(test, pool, item2cate, test_num_ngs)
    for path, lines in ((train_file, train), (valid_file, valid), (test_file, test)):
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(path, "w") as f:
            f.writelines(lines)

"""CPU tests: the host-side iterator / metrics / hparams against fixtures captured from the
reference (scripts/make_golden.py)."""
import json
import os
import random

import numpy as np
import pytest

from clsr_amd.deeprec_utils import (
    HParams, cal_mean_alpha_metric, cal_metric, cal_weighted_metric, prepare_hparams, roc_auc,
)
from clsr_amd.sequential_iterator import SASequentialIterator, SequentialIterator

FIELDS = ("labels users items cates item_history item_cate_history mask time time_diff "
          "time_from_first_action time_to_now").split()


def _check_feed(feed, gold, b, with_attn):
    names = FIELDS + (["attn_labels"] if with_attn else [])
    for name in names:
        exp = gold["b%d_%s" % (b, name)]
        got = feed[name]
        assert got.dtype == exp.dtype, (name, got.dtype, exp.dtype)
        assert got.shape == exp.shape, (name, got.shape, exp.shape)
        np.testing.assert_array_equal(got, exp, err_msg=name)


@pytest.mark.parametrize("cls,tag", [(SASequentialIterator, "sa"), (SequentialIterator, "plain")])
def test_iterator_matches_reference_feeds(golden_dir, golden_hparams, cls, tag):
    d = os.path.join(golden_dir, "data")
    it = cls(golden_hparams, None)
    gold = np.load(os.path.join(golden_dir, "iterator_train_%s.npz" % tag))
    random.seed(1234)
    gen = it.load_data_from_file(os.path.join(d, "train_data"), batch_num_ngs=4, min_seq_length=1)
    n = 0
    for feed in gen:
        if not feed:
            continue
        _check_feed(feed, gold, n, tag == "sa")
        n += 1
        if n == int(gold["n_batches"]):
            break
    assert n == int(gold["n_batches"])
    gold = np.load(os.path.join(golden_dir, "iterator_eval_%s.npz" % tag))
    n = 0
    for feed in it.load_data_from_file(os.path.join(d, "valid_data"), batch_num_ngs=0):
        _check_feed(feed, gold, n, tag == "sa")
        n += 1
    assert n == int(gold["n_batches"])


def test_iterator_drops_small_training_batches(golden_dir, golden_hparams, tmp_path):
    src = os.path.join(golden_dir, "data", "train_data")
    small = tmp_path / "small"
    with open(src) as f:
        lines = f.readlines()[:67]  # 64 + 3 -> trailing batch of 3 (<5) is dropped
    small.write_text("".join(lines))
    it = SASequentialIterator(golden_hparams, None)
    random.seed(0)
    feeds = list(it.load_data_from_file(str(small), batch_num_ngs=4))
    assert len(feeds) == 2 and feeds[1] is None
    assert feeds[0]["labels"].shape == (64 * 5, 1)
    # eval keeps the ragged tail and feeds users as float32 (reference quirk)
    feeds = list(it.load_data_from_file(str(small), batch_num_ngs=0))
    assert feeds[1]["labels"].shape == (3, 1) and feeds[1]["users"].dtype == np.float32


def test_iterator_history_truncation_and_oov(golden_hparams, tmp_path):
    hp = golden_hparams
    it = SASequentialIterator(hp, None)
    hist = ",".join("i%d" % i for i in range(25))
    cats = ",".join("c1" for _ in range(25))
    ts = ",".join(str(1511539200 + 100 * i) for i in range(25))
    line = "\t".join(["1", "nobody", "not_an_item", "not_a_cate", str(1511539200 + 5000), hist, cats, ts])
    parsed = it.parser_one_line(line)
    assert parsed[1] == 0 and parsed[2] == 0 and parsed[3] == 0
    res = it._convert_chunk([parsed], 0)
    assert res["mask"].sum() == hp.max_seq_length  # most recent T kept
    exp = [it.itemdict.get("i%d" % i, 0) for i in range(15, 25)]
    np.testing.assert_array_equal(res["item_history"][0], exp)


def test_metrics_match_reference_known_answers(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "metrics_golden.json")))
    labels = np.asarray(g["labels"], dtype=np.float32)
    preds = np.asarray(g["preds"], dtype=np.float32)
    users = g["users"]
    fl, fp = labels.reshape(-1).tolist(), preds.reshape(-1).tolist()
    res = {}
    res.update(cal_metric(fl, fp, ["auc", "logloss"]))
    res.update(cal_metric(list(labels), list(preds), ["mean_mrr", "ndcg@2;4;6", "hit@2;4;6", "group_auc"]))
    res.update(cal_weighted_metric(users, fp, fl, ["wauc", "whit@1;2"]))
    res.update(cal_mean_alpha_metric(fp, fl))
    for k, v in g["expected"].items():
        assert abs(res[k] - v) < 1e-9, (k, res[k], v)


def test_auc_matches_sklearn_with_ties():
    from sklearn.metrics import roc_auc_score

    rng = np.random.default_rng(3)
    for _ in range(20):
        y = rng.integers(0, 2, size=200)
        s = np.round(rng.random(200), 1)  # many ties
        assert abs(roc_auc(y, s) - roc_auc_score(y, s)) < 1e-12
    with pytest.raises(ValueError):
        roc_auc(np.ones(5), rng.random(5))


def test_f1_matches_sklearn():
    """cal_metric's "f1" (reference deeprec_utils.py:649-654: f1_score of the 0.5-thresholded predictions)."""
    from sklearn.metrics import f1_score

    from clsr_amd.deeprec_utils import cal_metric

    rng = np.random.default_rng(4)
    for _ in range(10):
        y = rng.integers(0, 2, size=150).astype(np.float64)
        s = rng.random(150)
        assert cal_metric(list(y), list(s), ["f1"])["f1"] == round(f1_score(y, (s >= 0.5).astype(np.float64)), 4)


def test_prepare_hparams_defaults_and_checks(golden_hparams):
    hp = golden_hparams
    assert isinstance(hp, HParams)
    assert "min_seq_length" in hp and hp.min_seq_length == 1
    assert hp.sequential_model == "time4lstm" and hp.contrastive_recent_k == 3
    assert hp.max_grad_norm == 2 and hp.optimizer == "adam" and hp.enable_BN is True
    hp.current_epoch = 3  # mutable like tf HParams
    assert hp.current_epoch == 3
    with pytest.raises(ValueError):
        prepare_hparams(None, model_type="clsr", item_embedding_dim=32)
    with pytest.raises(TypeError):
        prepare_hparams(None, model_type="other", learning_rate=1)


@pytest.mark.parametrize("n,ngs", [(5, 4), (64, 4), (1000, 2), (4096, 4), (2, 1)])
def test_vectorised_negative_sampling_replays_the_random_module(golden_hparams, n, ngs):
    """numpy MT19937 replay == literal random.randint rejection loop: same draws AND same RNG state after."""
    it = SASequentialIterator(golden_hparams, None)
    rng = np.random.default_rng(n)
    items = rng.integers(0, max(2, n // 3), size=n).tolist()   # many duplicates -> many rejections
    if len(set(items)) < 2:
        items[0] = 10 ** 6
    for seed in (0, 1, 12345):
        random.seed(seed)
        random.random()  # move the stream off its initial position
        src0 = np.empty((n, ngs + 1), dtype=np.int64)
        src0[:, 0] = np.arange(n)
        ref = it._sample_negatives_py(items, ngs, src0).copy()
        st_ref = random.getstate()
        random.seed(seed)
        random.random()
        got = it._sample_negatives(items, ngs)
        assert np.array_equal(got, ref)
        assert random.getstate() == st_ref


def test_grouped_metrics_vectorised_equals_per_group_loop():
    """2-D (equal-size groups) fast path of cal_metric == the reference-shaped loop over groups, ties included."""
    from clsr_amd.deeprec_utils import cal_metric

    rng = np.random.default_rng(3)
    metrics = ["mean_mrr", "ndcg@2;4;6", "hit@2;4;6"]
    for n_groups, size in ((1, 5), (37, 5), (200, 100), (50, 3)):
        preds = np.round(rng.random((n_groups, size)), 1)          # many ties
        preds[0] = 0.5                                             # a fully tied group
        labels = np.zeros((n_groups, size))
        labels[:, 0] = 1
        if n_groups > 2:
            labels[2, 1] = 1                                       # two positives in one group
        loop = cal_metric(list(labels), list(preds), metrics)
        fast = cal_metric(labels, preds, metrics)
        assert fast == loop and set(fast) == {"mean_mrr", "ndcg@2", "ndcg@4", "ndcg@6", "hit@2", "hit@4", "hit@6"}
        both = metrics + ["group_auc"]
        assert cal_metric(labels, preds, both) == cal_metric(list(labels), list(preds), both)
    # metrics without a vectorised form fall back to the loop
    assert cal_metric(labels, preds, ["mean_mrr", "acc"]) == cal_metric(list(labels), list(preds), ["mean_mrr", "acc"])


def test_group_and_user_auc_vectorised_equal_the_per_group_roc_auc():
    """_grouped_auc (rows of a 2-D array) and _segmented_auc (one lexsort over (user, score)) give bit-identical
    values to roc_auc called group by group (ties averaged), and keep sklearn's single-class ValueError."""
    from clsr_amd import deeprec_utils as du

    rng = np.random.default_rng(11)
    for g, n in ((1, 2), (300, 5), (40, 100), (11, 7)):
        labels = np.zeros((g, n))
        for i in range(g):
            labels[i, rng.choice(n, int(rng.integers(1, min(3, n))), replace=False)] = 1
        preds = np.round(rng.random((g, n)), 1 if n > 5 else 2)
        want = np.array([du.roc_auc(l, p) for l, p in zip(labels, preds)])
        assert np.array_equal(du._grouped_auc(labels, preds), want)
        users = np.repeat(rng.permutation(g) // 3, n)              # a user owns up to three (non-adjacent) groups
        vals, weight = du._segmented_auc(users, preds.reshape(-1), labels.reshape(-1))
        groups, weight_loop = du._group_by_user(users, preds.reshape(-1), labels.reshape(-1))
        assert np.array_equal(vals, np.array([du.roc_auc(l, p) for p, l in groups]))
        assert np.array_equal(weight, weight_loop)
    with pytest.raises(ValueError):
        du.cal_weighted_metric([1, 1, 2, 2], [.1, .2, .3, .4], [1, 0, 1, 1], ["wauc"])
    with pytest.raises(ValueError):
        du.cal_metric(np.array([[1., 0.], [0., 0.]]), np.array([[.3, .2], [.1, .2]]), ["group_auc"])


def test_native_mt_replay_equals_python_random():
    """clsr_host_mt_shuffle / clsr_host_mt_sample_negatives continue the `random` stream exactly like
    random.shuffle / the literal randint loop (same results, same generator state afterwards)."""
    import random

    from clsr_amd import sequential_iterator as si

    if si._native_lib() is None:
        pytest.skip("libclsr_hip.so not built")
    for n in (1, 2, 7, 1000, 4097, 70000):
        random.seed(n)
        lst = list(range(n))
        random.shuffle(lst)
        st_py = random.getstate()
        random.seed(n)
        perm = si.shuffle_like_random(np.arange(n, dtype=np.int64))
        assert perm.tolist() == lst and random.getstate() == st_py
    it = si.SequentialIterator.__new__(si.SequentialIterator)
    rng = np.random.default_rng(1)
    for n, V in ((5, 2), (37, 3), (4096, 50), (4097, 3000)):
        items = rng.integers(0, V, n).tolist()
        if len(set(items)) < 2:
            items[0] = V + 1
        random.seed(n)
        a = it._sample_negatives(items, 4)
        sa = random.getstate()
        random.seed(n)
        src = np.empty((n, 5), dtype=np.int64)
        src[:, 0] = np.arange(n)
        b = it._sample_negatives_py(items, 4, src)
        assert np.array_equal(a, b) and sa == random.getstate()


def test_native_tsv_tokenizer_equals_the_literal_parser(golden_dir, golden_hparams, tmp_path):
    """clsr_host_tsv_parse + the numpy time features == parse_file / parser_one_line + _columns_of, array for array
    (both time units, histories longer than max_seq_length, unknown tokens); irregular files are left to the literal
    parser; iter_data still answers len() and yields the reference-shaped tuples on demand."""
    import copy

    from clsr_amd import sequential_iterator as S

    if S._native_lib() is None or not hasattr(S._native_lib(), "clsr_host_tsv_parse"):
        pytest.skip("libclsr_hip.so not built")
    d = os.path.join(golden_dir, "data")
    for unit, T in (("s", 10), ("ms", 4)):
        hp = copy.deepcopy(golden_hparams)
        hp.time_unit, hp.max_seq_length = unit, T
        it = SASequentialIterator(hp, None)
        for name in ("train_data", "valid_data", "test_data"):
            path = os.path.join(d, name)
            got = it._parse_columns_native(path)
            assert got is not None
            want = it._columns_of("literal:" + name, it.parse_file(path))
            assert set(got) == set(want)
            for k, v in want.items():
                if isinstance(v, np.ndarray):
                    assert got[k].dtype == v.dtype and np.array_equal(got[k], v), (unit, name, k)
                else:
                    assert got[k] == v
    it = SASequentialIterator(golden_hparams, None)
    path = os.path.join(d, "train_data")
    feeds = [f for f in it.load_data_from_file(path, batch_num_ngs=0) if f]
    lines = it.iter_data[path]
    assert len(lines) == sum(f["labels"].shape[0] for f in feeds)
    assert lines._lines is None                       # nobody asked for the tuples: never parsed line by line
    first = lines[0]
    assert len(first) == 10 and first[0] in (0, 1) and lines._lines is not None
    # irregular inputs: ragged history columns, a short line, a non-numeric timestamp -> the literal parser decides
    good = open(path).readline()
    w = good.rstrip("\n").split("\t")
    for bad in ("\t".join(w[:5] + [w[5] + ",i1"] + w[6:]), "\t".join(w[:7]), "\t".join(w[:4] + ["abc"] + w[5:])):
        p = tmp_path / "bad.tsv"
        p.write_text(good + bad + "\n")
        assert it._parse_columns_native(str(p)) is None

"""GPU tests of the device-side evaluation metrics (clsr_amd/csrc/metrics.hip, clsr_amd/device_metrics.py) against
the host implementations of clsr_amd/deeprec_utils.py -- which are themselves pinned to the reference's cal_metric /
cal_weighted_metric by tests/golden/metrics_golden.json -- and against that reference-captured fixture directly."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd import device_metrics as DM  # noqa: E402
from clsr_amd.deeprec_utils import cal_metric, cal_weighted_metric  # noqa: E402

DEV = "cuda:0"


def _hp(**kw):
    d = dict(metrics=["auc", "logloss"], pairwise_metrics=["mean_mrr", "ndcg@2;4;6", "hit@2;4;6", "group_auc"],
             weighted_metrics=["wauc"])
    d.update(kw)
    return types.SimpleNamespace(**d)


def _same(got, exp, raw, k):
    """equal after the 4-decimal rounding -- or one unit apart when the exact value sits ON a rounding boundary (a mean
    of dyadic group AUCs such as 1999/4000 = 0.49975: summation order decides the last bit, hence the direction)"""
    if abs(got - exp) < 1e-9:
        return True
    frac = (raw[k] * 1e4) % 1.0
    return abs(got - exp) < 1.0000001e-4 and abs(frac - 0.5) < 1e-7


def _device(preds, labels, users, hp, group, chunks=3, raw=None):
    acc = DM.DeviceScores(torch.device(DEV))
    n = len(preds)
    cut = sorted(set([0, n] + [n * i // chunks for i in range(1, chunks)]))
    for a, b in zip(cut[:-1], cut[1:]):       # appended in pieces like an evaluation loop does
        acc.append(torch.tensor(preds[a:b], dtype=torch.float32, device=DEV),
                   torch.tensor(labels[a:b], dtype=torch.float32, device=DEV),
                   torch.tensor(users[a:b], dtype=torch.int32, device=DEV))
    return DM.compute(acc, hp, group, True, raw=raw)


def _host(preds, labels, users, hp, group):
    p32 = np.asarray(preds, dtype=np.float32)
    res = cal_metric(list(labels), list(p32), hp.metrics)
    res.update(cal_metric(np.reshape(np.asarray(labels), (-1, group)), np.reshape(p32, (-1, group)), hp.pairwise_metrics))
    res.update(cal_weighted_metric(users, p32, labels, hp.weighted_metrics))
    return res


def test_reference_captured_known_answers(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "metrics_golden.json")))
    labels = np.asarray(g["labels"], dtype=np.float64)
    preds = np.asarray(g["preds"], dtype=np.float64)
    group = labels.shape[1]
    users = np.asarray(g["users"]).astype(np.int64)
    res = _device(preds.reshape(-1), labels.reshape(-1), users, _hp(), group)
    # the three AUCs and the logloss are tie-exact: equal to what the REFERENCE computed (the fixture)
    for k in ("auc", "logloss", "group_auc", "wauc"):
        assert abs(res[k] - g["expected"][k]) < 1e-9, (k, res[k], g["expected"][k])
    # rank metrics: the fixture holds groups where the positive's score is tied with a negative's; numpy's default
    # argsort (SIMD quicksort on this generation of numpy) orders such ties platform-dependently -- neither "earlier
    # line first" nor "later line first" reproduces the captured 0.2559 -- so the reference itself is ill-defined
    # there.  On the groups WITHOUT such a tie the device equals the host functions that the CPU suite pins to the
    # fixture; the device's own rule for ties (later line first) is exercised in the next test.
    tied = np.array([any(p[j] == p[i] and j != i for i in np.flatnonzero(l == 1) for j in range(group))
                     for l, p in zip(labels, preds)])
    assert tied.any() and not tied.all()
    lab, prd = labels[~tied], preds[~tied]
    hp = _hp(metrics=[], weighted_metrics=[], pairwise_metrics=["mean_mrr", "ndcg@2;4;6", "hit@2;4;6"])
    got = _device(prd.reshape(-1), lab.reshape(-1), np.zeros(lab.size, dtype=np.int64), hp, group)
    exp = cal_metric(lab, prd.astype(np.float32), hp.pairwise_metrics)
    for k in exp:
        assert abs(got[k] - exp[k]) < 1e-9, (k, got[k], exp[k])


def test_rank_ties_break_towards_the_later_line():
    """Documented tie rule of the device metrics: equal scores rank like a stable ascending sort read backwards."""
    hp = _hp(metrics=[], weighted_metrics=[], pairwise_metrics=["mean_mrr", "hit@1;2"])
    preds = np.array([0.5, 0.5, 0.5, 0.1,      # positive first, tied with two later negatives: rank 3
                      0.2, 0.7, 0.7, 0.1])     # positive at index 2 of the group, tied with an EARLIER negative: rank 1
    labels = np.array([1.0, 0, 0, 0, 0, 0, 1.0, 0])
    got = _device(preds, labels, np.zeros(8, dtype=np.int64), hp, 4, chunks=1)
    assert abs(got["mean_mrr"] - round((1.0 / 3 + 1.0) / 2, 4)) < 1e-12
    assert got["hit@1"] == 0.5 and got["hit@2"] == 0.5


@pytest.mark.parametrize("n_groups,group,n_users,ties", [(2000, 5, 300, False), (700, 100, 64, False), (300, 100, 40, True),
                                                        (5000, 10, 5000, True), (1, 7, 1, False)])
def test_device_metrics_equal_the_host_metrics(n_groups, group, n_users, ties):
    """Random files: tie-free scores for the rank metrics; heavy ties (scores rounded to 2 decimals) for the three AUCs
    and logloss, which are tie-exact by construction (numpy's argsort orders ties platform-dependently even in small
    groups, see above: mean_mrr / ndcg / hit are compared on tie-free scores only)."""
    rng = np.random.default_rng(n_groups + group)
    N = n_groups * group
    labels = np.zeros((n_groups, group))
    labels[:, 0] = 1.0
    extra = rng.random((n_groups, group)) < 0.03          # a few groups with more than one positive
    extra[:, -1] = False                                   # (always at least one negative)
    labels = np.maximum(labels, extra)
    preds = ((rng.permutation(N) + 0.5) / N).astype(np.float32)     # distinct float32 scores (N < 2^24): tie-free
    assert np.unique(preds).size == N
    if ties:
        preds = np.round(preds, 2).astype(np.float32)
        preds[:7] = [0.0, 1.0, 1e-13, 1.0 - 1e-9, 0.5, 0.5, 0.25][: min(7, N)]  # clipping edges of the logloss
    users = np.repeat(rng.integers(0, n_users, n_groups), group)
    hp = _hp() if not ties else _hp(pairwise_metrics=["group_auc"])
    raw = {}
    got = _device(preds, labels.reshape(-1), users, hp, group, raw=raw)
    exp = _host(preds, labels.reshape(-1), users, hp, group)
    assert set(got) == set(exp)
    for k in exp:
        assert _same(got[k], exp[k], raw, k), (k, got[k], exp[k], raw[k])


def test_undefined_auc_raises_like_the_host_path():
    hp = _hp(pairwise_metrics=[], weighted_metrics=[])
    with pytest.raises(ValueError):
        _device(np.array([0.2, 0.4, 0.6, 0.8]), np.zeros(4), np.zeros(4, dtype=np.int64), hp, 2)
    hp = _hp(metrics=[], pairwise_metrics=[], weighted_metrics=["wauc"])
    with pytest.raises(ValueError):    # user 1 has positives only
        _device(np.array([0.2, 0.4, 0.6, 0.8]), np.array([1.0, 0.0, 1.0, 1.0]), np.array([0, 0, 1, 1]), hp, 2)


def test_model_evaluation_on_the_device_equals_the_host_path(golden_dir, golden_hparams, monkeypatch):
    """run_eval / run_weighted_eval through the device metrics == the same calls with CLSR_HOST_METRICS=1."""
    from clsr_amd.clsr import CLSRModel
    from clsr_amd.sequential_iterator import SASequentialIterator

    import pickle

    from oracle import clsr_oracle as O

    hp = golden_hparams
    model = CLSRModel(hp, SASequentialIterator, seed=2)
    # weights with some spread: the freshly initialised model (sigma 0.01) scores every line 0.5 +- 1e-7, i.e. with
    # exact float32 ties inside groups, where the host's numpy argsort and the device's rule legitimately differ
    dims = dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))
    params = O.init_params(dims, hp, seed=5, scale_dense=8.0)
    sd = dict(params)
    sd.update(O.init_bn_state(params))
    model.net.load_state_dict(sd)
    valid = os.path.join(golden_dir, "data", "valid_data")
    dev_w, dev_e = model.run_weighted_eval(valid, num_ngs=4), model.run_eval(valid, num_ngs=4)
    monkeypatch.setenv("CLSR_HOST_METRICS", "1")
    host_w, host_e = model.run_weighted_eval(valid, num_ngs=4), model.run_eval(valid, num_ngs=4)
    assert dev_w == host_w and dev_e == host_e and "wauc" in dev_w and "wauc" not in dev_e

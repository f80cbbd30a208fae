"""Evaluation metrics computed on the device (csrc/metrics.hip) -- the scores of a whole validation / test file stay
in HBM, no per-batch synchronisation, and a handful of doubles comes back at the end.

Mirrors ``cal_metric`` / ``cal_weighted_metric`` of the reference (deeprec_utils.py:621-806) as
``SequentialBaseModel.run_eval`` / ``run_weighted_eval`` call them (sequential_base_model.py:204-292): same keys,
same 4-decimal rounding, same ``ValueError`` when an AUC is undefined.  Supported: ``auc``, ``logloss`` (all lines);
``mean_mrr``, ``ndcg@k``, ``hit@k``, ``group_auc`` (groups of 1 + num_ngs consecutive lines); ``wauc`` (per user).
Anything else -- or more than 2^18 users, whose grouping is not exact on the device -- makes :func:`supported` return
False and the caller keeps the host path (clsr_amd/deeprec_utils.py)."""
import ctypes

import torch

from clsr_amd import ops
from clsr_amd.deeprec_utils import _ks

_POINT = {"auc", "logloss"}


def supported(hp, n_users):
    if any(m not in _POINT for m in (hp.metrics or [])):
        return False
    ks = set()
    for m in (hp.pairwise_metrics or []):
        if m.startswith("ndcg") or m.startswith("hit"):
            ks.update(_ks(m))
        elif m not in ("mean_mrr", "group_auc"):
            return False
    if len(ks) > 8:
        return False
    if any(m != "wauc" for m in (getattr(hp, "weighted_metrics", None) or [])):
        return False
    return n_users <= (1 << 18)


class DeviceScores(object):
    """Growing device buffers of (pred, label, user) per scored line; appends are stream-ordered copies."""

    def __init__(self, device):
        self.device, self.n, self.cap = device, 0, 0
        self.pred = self.labels = self.users = None

    def _grow(self, need):
        cap = max(need, 2 * self.cap, 1 << 16)
        new = (torch.empty(cap, dtype=torch.float32, device=self.device),
               torch.empty(cap, dtype=torch.float32, device=self.device),
               torch.empty(cap, dtype=torch.int32, device=self.device))
        if self.n:
            for dst, src in zip(new, (self.pred, self.labels, self.users)):
                dst[: self.n].copy_(src[: self.n])
        self.pred, self.labels, self.users = new
        self.cap = cap

    def append(self, pred, labels, users):
        b = pred.numel()
        if self.n + b > self.cap:
            self._grow(self.n + b)
        self.pred[self.n:self.n + b].copy_(pred.reshape(-1))
        self.labels[self.n:self.n + b].copy_(labels.reshape(-1))
        if users is not None:
            self.users[self.n:self.n + b].copy_(users.reshape(-1))
        self.n += b


def compute(scores, hp, group, weighted, raw=None):
    """-> dict of metrics (the union of what cal_metric(metrics), cal_metric(pairwise_metrics) and, when
    ``weighted``, cal_weighted_metric(weighted_metrics) return).  Synchronises once.  ``raw`` (a dict) receives the
    unrounded values (tests: a mean that sits exactly on a 4-decimal rounding boundary may round either way)."""
    N, dev = scores.n, scores.device
    if N == 0 or N % group:
        raise ValueError("%d scored lines do not divide into groups of %d" % (N, group))
    pred, labels = scores.pred[:N], scores.labels[:N]
    outd = torch.zeros(32, dtype=torch.float64, device=dev)      # [0] logloss | [1..] group metrics | [30] wauc
    outu = torch.zeros(3, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    err = torch.zeros(2, dtype=torch.int32, device=dev)
    metrics = list(hp.metrics or [])
    if "logloss" in metrics:
        ops.call("clsr_eval_logloss", pred, labels, N, outd)
    if "auc" in metrics:
        pos = torch.empty(N, dtype=torch.float32, device=dev)
        ops.call("clsr_eval_compact_pos", pred, labels, N, pos, cnt)
        ops.call("clsr_eval_auc_pairs", pred, labels, N, pos, cnt, outu)
    pair = list(hp.pairwise_metrics or [])
    ks = sorted({k for m in pair if m.startswith("ndcg") or m.startswith("hit") for k in _ks(m)})
    if pair:
        karr = (ctypes.c_int * max(len(ks), 1))(*ks)
        ops.call("clsr_eval_group_metrics", pred, labels, N // group, group, ctypes.addressof(karr), len(ks),
                 1 if "group_auc" in pair else 0, outd[1:], err)
    wm = list(getattr(hp, "weighted_metrics", None) or []) if weighted else []
    if wm:
        users = scores.users[:N]
        bits = ops.query("clsr_sort_ids_bits", 1 << 18)
        nb = 1 << bits
        counts = torch.zeros(nb, dtype=torch.int32, device=dev)
        keys = torch.empty(N, dtype=torch.int32, device=dev)
        perm = torch.empty(N, dtype=torch.int32, device=dev)
        ops.sort_ids_multi([(users.data_ptr(), keys.data_ptr(), perm.data_ptr(), counts.data_ptr(), N, 1, 1, bits)])
        ops.call("clsr_eval_user_auc", pred, labels, perm, counts, nb, N, outd[30:], err[1:])
    d, u, c, e = outd.cpu().tolist(), outu.cpu().tolist(), int(cnt.cpu()), err.cpu().tolist()   # the one sync
    res = {}
    for m in metrics:
        if m == "auc":
            if c == 0 or u[2] == 0:
                raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
            res["auc"] = (u[0] + 0.5 * u[1]) / (float(c) * float(u[2]))
        elif m == "logloss":
            res["logloss"] = d[0] / N
    ng = N // group
    if pair and e[0]:
        raise ValueError("%d groups without a positive (or, for group_auc, without a negative) line" % e[0])
    for m in pair:
        if m == "mean_mrr":
            res["mean_mrr"] = d[1] / ng
        elif m == "group_auc":
            res["group_auc"] = d[2] / ng
        elif m.startswith("ndcg"):
            for k in _ks(m):
                res["ndcg@{0}".format(k)] = d[3 + ks.index(k)] / ng
        elif m.startswith("hit"):
            for k in _ks(m):
                res["hit@{0}".format(k)] = d[3 + len(ks) + ks.index(k)] / ng
    if wm:
        if e[1]:
            raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
        res["wauc"] = d[30]
    if raw is not None:
        raw.update(res)
    return {k: round(v, 4) for k, v in res.items()}

"""Config (yaml -> HParams) and evaluation metrics for the CLSR path.

Host-side mirror of the reference's
``reco_utils/recommender/deeprec/deeprec_utils.py`` for the pieces the CLSR
step touches:

* ``prepare_hparams`` / ``create_hparams`` / ``check_nn_config`` / ``check_type``
  (reference ``deeprec_utils.py:25-135, 245-262, 327-534``) -- same keys, same
  defaults, same error types; the TF ``HParams`` container is replaced by a
  plain attribute bag that also supports ``in`` (``sequential_base_model.py:38``).
* ``cal_metric`` / ``cal_weighted_metric`` and the per-group scores
  (reference ``deeprec_utils.py:554-821``) -- re-stated with numpy; parity with
  the reference functions is pinned by ``tests/golden/metrics_golden.json``.

Nothing here runs on the GPU.
"""
import pickle as pkl

import numpy as np
import yaml

__all__ = [
    "HParams",
    "prepare_hparams",
    "create_hparams",
    "check_nn_config",
    "check_type",
    "flat_config",
    "load_yaml",
    "load_dict",
    "cal_metric",
    "cal_weighted_metric",
    "cal_mean_alpha_metric",
    "mrr_score",
    "ndcg_score",
    "hit_score",
    "dcg_score",
    "roc_auc",
]


class HParams(object):
    """Attribute bag standing in for ``tf.contrib.training.HParams``.

    Supports attribute get/set, ``key in hparams`` and ``values()``
    (the operations the reference performs on the TF object).
    """

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            object.__setattr__(self, k, v)

    def __contains__(self, key):
        return key in self.__dict__

    def values(self):
        return dict(self.__dict__)

    def add_hparam(self, name, value):
        if name in self.__dict__:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        setattr(self, name, value)

    def set_hparam(self, name, value):
        setattr(self, name, value)

    def __repr__(self):
        return "HParams(%s)" % ", ".join(
            "%s=%r" % kv for kv in sorted(self.__dict__.items())
        )


def flat_config(config):
    """Flatten the ``data/model/train/info`` yaml sections (ref ``:25-39``)."""
    flat = {}
    for section in config.keys():
        for key, val in config[section].items():
            flat[key] = val
    return flat


_INT_PARAMS = (
    "word_size entity_size doc_size history_size FEATURE_COUNT FIELD_COUNT dim epochs "
    "batch_size show_step save_epoch PAIR_NUM DNN_FIELD_NUM attention_layer_sizes n_user "
    "n_item n_user_attr n_item_attr item_embedding_dim cate_embedding_dim user_embedding_dim "
    "max_seq_length hidden_size T L n_v n_h kernel_size min_seq_length attention_size "
    "train_num_ngs"
).split()
_FLOAT_PARAMS = "init_value learning_rate embed_l2 embed_l1 layer_l2 layer_l1 mu".split()
_STR_PARAMS = (
    "train_file eval_file test_file infer_file method load_model_name infer_model_name loss "
    "optimizer init_method attention_activation user_vocab item_vocab cate_vocab"
).split()
_LIST_PARAMS = "layer_sizes activation dropout att_fcn_layer_sizes dilations".split()


def check_type(config):
    """Type-check config values; raises ``TypeError`` like ref ``:42-135``."""
    for names, typ, label in (
        (_INT_PARAMS, int, "int"),
        (_FLOAT_PARAMS, float, "float"),
        (_STR_PARAMS, str, "str"),
        (_LIST_PARAMS, list, "list"),
    ):
        for p in names:
            if p in config and not isinstance(config[p], typ):
                raise TypeError("Parameters {0} must be {1}".format(p, label))


_CLSR_REQUIRED = [
    "item_embedding_dim",
    "cate_embedding_dim",
    "max_seq_length",
    "loss",
    "method",
    "user_vocab",
    "item_vocab",
    "cate_vocab",
    "attention_size",
    "hidden_size",
    "att_fcn_layer_sizes",
    "discrepancy_loss_weight",
    "contrastive_loss_weight",
    "is_clip_norm",
    "contrastive_length_threshold",
]


def check_nn_config(f_config):
    """Required-key check for ``model_type`` clsr (ref ``:245-262, 288-307``).

    Only the CLSR branch is in scope; other model types get no required keys.
    """
    if f_config.get("model_type") in ("clsr", "CLSR"):
        required = _CLSR_REQUIRED
    else:
        required = []
    for p in required:
        if p not in f_config:
            raise ValueError("Parameters {0} must be set".format(p))
    check_type(f_config)


def load_yaml(filename):
    """Load a yaml file; non-FileNotFound failures become ``IOError`` (ref ``:310-324``)."""
    try:
        with open(filename, "r") as f:
            return yaml.load(f, yaml.SafeLoader)
    except FileNotFoundError:
        raise
    except Exception:
        raise IOError("load {0} error!".format(filename))


# (key, default) exactly as create_hparams fills them (ref deeprec_utils.py:327-511).
_DEFAULTS = [
    ("kg_file", None), ("user_clicks", None), ("FEATURE_COUNT", None), ("FIELD_COUNT", None),
    ("data_format", None), ("PAIR_NUM", None), ("DNN_FIELD_NUM", None), ("n_user", None),
    ("n_item", None), ("n_user_attr", None), ("n_item_attr", None), ("iterator_type", None),
    ("SUMMARIES_DIR", None), ("MODEL_DIR", None),
    ("wordEmb_file", None), ("entityEmb_file", None), ("contextEmb_file", None),
    ("news_feature_file", None), ("user_history_file", None), ("use_entity", True),
    ("use_context", True), ("doc_size", None), ("history_size", None), ("word_size", None),
    ("entity_size", None), ("entity_dim", None), ("entity_embedding_method", None),
    ("transform", None), ("train_ratio", None),
    ("dim", None), ("layer_sizes", None), ("cross_layer_sizes", None), ("cross_layers", None),
    ("activation", None), ("cross_activation", "identity"), ("user_dropout", False),
    ("dropout", [0.0]), ("attention_layer_sizes", None), ("attention_activation", None),
    ("attention_dropout", 0.0), ("model_type", None), ("method", None),
    ("load_saved_model", False), ("load_model_name", None), ("filter_sizes", None),
    ("num_filters", None), ("mu", None), ("fast_CIN_d", 0), ("use_Linear_part", False),
    ("use_FM_part", False), ("use_CIN_part", False), ("use_DNN_part", False),
    ("init_method", "tnormal"), ("init_value", 0.01), ("embed_l2", 0.0), ("embed_l1", 0.0),
    ("layer_l2", 0.0), ("layer_l1", 0.0), ("cross_l2", 0.0), ("cross_l1", 0.0),
    ("attn_loss_weight", 0.0), ("contrastive_loss", "bpr"), ("triplet_margin", 1.0),
    ("discrepancy_loss_weight", 0.0), ("contrastive_loss_weight", 0.0),
    ("contrastive_length_threshold", 1), ("contrastive_recent_k", 3), ("reg_kg", 0.0),
    ("learning_rate", 0.001), ("lr_rs", 1), ("lr_kg", 0.5), ("kg_training_interval", 5),
    ("max_grad_norm", 2), ("is_clip_norm", 0), ("vector_alpha", False), ("manual_alpha", False),
    ("manual_alpha_value", 0.5), ("interest_evolve", True), ("predict_long_short", True),
    ("dtype", 32), ("loss", None), ("optimizer", "adam"), ("epochs", 10), ("batch_size", 1),
    ("enable_BN", False),
    ("show_step", 1), ("save_model", True), ("save_epoch", 5), ("metrics", None),
    ("write_tfevents", False),
    ("item_embedding_dim", None), ("cate_embedding_dim", None), ("user_embedding_dim", None),
    ("train_num_ngs", 4), ("need_sample", True), ("embedding_dropout", 0.3),
    ("user_vocab", None), ("item_vocab", None), ("cate_vocab", None),
    ("pairwise_metrics", None), ("weighted_metrics", None), ("EARLY_STOP", 100),
    ("max_seq_length", None), ("hidden_size", None),
    ("L", None), ("T", None), ("n_v", None), ("n_h", None), ("min_seq_length", 1),
    ("attention_size", None), ("att_fcn_layer_sizes", None),
    ("dilations", None), ("kernel_size", None),
    ("embed_size", None), ("n_layers", None), ("decay", None), ("eval_epoch", None),
    ("top_k", None), ("counterfactual_recent_k", 5), ("use_complex_attention", False),
    ("sequential_model", "time4lstm"), ("time_unit", "s"), ("ncf_layer_sizes", [80, 40]),
]


def create_hparams(flags):
    """Fill defaults for every key the reference knows (ref ``:327-511``)."""
    vals = {}
    for key, default in _DEFAULTS:
        vals[key] = flags[key] if key in flags else default
    return HParams(**vals)


def prepare_hparams(yaml_file=None, **kwargs):
    """yaml (flattened) + kwargs overrides -> checked ``HParams`` (ref ``:514-534``)."""
    if yaml_file is not None:
        config = flat_config(load_yaml(yaml_file))
    else:
        config = {}
    for name, value in kwargs.items():
        config[name] = value
    check_nn_config(config)
    return create_hparams(config)


def load_dict(filename):
    """Load a pickled vocabulary ``dict[str -> int]`` (ref ``:816-828``)."""
    with open(filename, "rb") as f:
        return pkl.load(f)


# --------------------------------------------------------------------------- metrics
def roc_auc(y_true, y_score):
    """ROC-AUC via mid-ranks (ties averaged) == ``sklearn.metrics.roc_auc_score``.

    Raises ``ValueError`` when only one class is present, like sklearn does."""
    y_true = np.asarray(y_true).reshape(-1)
    y_score = np.asarray(y_score, dtype=np.float64).reshape(-1)
    pos = y_true == 1
    n_pos = int(pos.sum())
    n_neg = y_true.size - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError(
            "Only one class present in y_true. ROC AUC score is not defined in that case."
        )
    order = np.argsort(y_score, kind="mergesort")
    s = y_score[order]
    new = np.concatenate(([True], s[1:] != s[:-1]))
    grp = np.cumsum(new) - 1
    starts = np.flatnonzero(new)
    ends = np.concatenate((starts[1:], [s.size]))
    mid = 0.5 * (starts + ends - 1) + 1.0
    ranks = np.empty(s.size, dtype=np.float64)
    ranks[order] = mid[grp]
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def _log_loss(labels, preds):
    """Binary log-loss == ``sklearn.metrics.log_loss`` for probabilities in (0,1)."""
    y = np.asarray(labels, dtype=np.float64).reshape(-1)
    p = np.asarray(preds, dtype=np.float64).reshape(-1)
    eps = np.finfo(np.float64).eps
    p = np.clip(p, eps, 1 - eps)
    return float(-np.mean(y * np.log(p) + (1 - y) * np.log(1 - p)))


def mrr_score(y_true, y_score):
    """Reciprocal rank of the positives under descending score (ref ``:554-567``)."""
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order)
    rr = y_true / (np.arange(len(y_true)) + 1)
    return np.sum(rr) / np.sum(y_true)


def dcg_score(y_true, y_score, k=10):
    """DCG@k with gains ``2^y - 1`` and log2 discounts (ref ``:606-618``)."""
    k = min(np.shape(y_true)[-1], k)
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order[:k])
    gains = 2 ** y_true - 1
    discounts = np.log2(np.arange(len(y_true)) + 2)
    return np.sum(gains / discounts)


def ndcg_score(y_true, y_score, k=10):
    """NDCG@k (ref ``:570-582``)."""
    return dcg_score(y_true, y_score, k) / dcg_score(y_true, y_true, k)


def hit_score(y_true, y_score, k=10):
    """1 if any positive is ranked in the top k (ref ``:585-603``)."""
    y_true = np.asarray(y_true)
    top = np.argsort(y_score)[::-1][:k]
    return 1 if np.any(y_true[top] == 1) else 0


def _ks(metric, default=(1, 2)):
    parts = metric.split("@")
    if len(parts) > 1:
        return [int(tok) for tok in parts[1].split(";")]
    return list(default)


def _grouped_rank_metrics(labels, preds, metrics):
    """mean_mrr / ndcg@k / hit@k for equal-size groups given as 2-D arrays ``[groups, size]``: ONE argsort
    along axis 1 instead of a python loop over groups.  Every row is sorted by the same routine as the
    per-group functions above (``np.argsort(row)[::-1]``), so ties break identically and the results are
    bit-identical to the loop."""
    labels = np.asarray(labels)
    order = np.argsort(preds, axis=1)[:, ::-1]
    ranked = np.take_along_axis(labels, order, axis=1)
    n = labels.shape[1]
    res = {}
    for metric in metrics:
        if metric == "mean_mrr":
            rr = ranked / (np.arange(n) + 1)
            res["mean_mrr"] = round(float(np.mean(np.sum(rr, axis=1) / np.sum(ranked, axis=1))), 4)
        elif metric.startswith("ndcg"):
            ideal = np.take_along_axis(labels, np.argsort(labels, axis=1)[:, ::-1], axis=1)
            for k in _ks(metric):
                kk = min(n, k)
                disc = np.log2(np.arange(kk) + 2)
                dcg = np.sum((2 ** ranked[:, :kk] - 1) / disc, axis=1)
                idcg = np.sum((2 ** ideal[:, :kk] - 1) / disc, axis=1)
                res["ndcg@{0}".format(k)] = round(float(np.mean(dcg / idcg)), 4)
        elif metric.startswith("hit"):
            for k in _ks(metric):
                hit = np.any(ranked[:, :k] == 1, axis=1)
                res["hit@{0}".format(k)] = round(float(np.mean(np.where(hit, 1, 0))), 4)
        elif metric == "group_auc":
            res["group_auc"] = round(float(np.mean(_grouped_auc(labels, preds))), 4)
        else:
            return None
    return res


def _grouped_auc(labels, preds):
    """``roc_auc`` of every row of ``[groups, n]`` arrays at once.  The mid-rank formula of ``roc_auc`` equals
    ``(#{pos > neg} + 0.5 #{pos == neg}) / (n_pos n_neg)``; both numerators are exact in float64 (multiples of
    0.5), so the quotients are bit-identical to the per-group loop."""
    labels = np.asarray(labels)
    preds = np.asarray(preds, dtype=np.float64)
    g, n = labels.shape
    pos = labels == 1
    n_pos = pos.sum(axis=1).astype(np.float64)
    n_neg = n - n_pos
    if np.any(n_pos == 0) or np.any(n_neg == 0):
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    out = np.empty(g, dtype=np.float64)
    step = max(1, (1 << 23) // (n * n))
    for a in range(0, g, step):
        s, m = preds[a:a + step], pos[a:a + step]
        pair = m[:, :, None] & ~m[:, None, :]                      # (i positive, j negative)
        gt = (s[:, :, None] > s[:, None, :]) & pair
        eq = (s[:, :, None] == s[:, None, :]) & pair
        out[a:a + step] = gt.sum(axis=(1, 2)) + 0.5 * eq.sum(axis=(1, 2))
    return out / (n_pos * n_neg)


def _segmented_auc(users, preds, labels):
    """(per-user ``roc_auc``, per-user weight) with ONE lexsort over (user, score) instead of a python loop over
    users: mid-ranks inside each user's segment, rank sums by ``bincount`` (exact: multiples of 0.5)."""
    users = np.asarray(users).reshape(-1)
    preds = np.asarray(preds, dtype=np.float64).reshape(-1)
    pos = (np.asarray(labels).reshape(-1) == 1)
    order = np.lexsort((preds, users))
    u, s, m = users[order], preds[order], pos[order]
    n = u.size
    new_u = np.concatenate(([True], u[1:] != u[:-1]))
    uid = np.cumsum(new_u) - 1
    ustart = np.flatnonzero(new_u)
    new_t = new_u | np.concatenate(([True], s[1:] != s[:-1]))       # start of a tie group inside a user
    tid = np.cumsum(new_t) - 1
    tstart = np.flatnonzero(new_t)
    tend = np.concatenate((tstart[1:], [n]))
    mid = 0.5 * (tstart + tend - 1) + 1.0 - ustart[uid[tstart]]     # mid-rank (1-based) inside the user
    ranks = mid[tid]
    cnt = np.bincount(uid).astype(np.float64)
    n_pos = np.bincount(uid, weights=m.astype(np.float64))
    n_neg = cnt - n_pos
    if np.any(n_pos == 0) or np.any(n_neg == 0):
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    rsum = np.bincount(uid, weights=np.where(m, ranks, 0.0))
    return (rsum - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg), cnt / float(n)


def cal_metric(labels, preds, metrics):
    """Pointwise / groupwise metrics rounded to 4 dp (ref ``:621-699``).

    ``labels``/``preds`` are flat lists for auc/logloss and lists of groups for the
    pairwise metrics, exactly as the reference's callers pass them.  Equal-size groups passed
    as 2-D numpy arrays take a vectorised path with identical results.
    """
    res = {}
    if not metrics:
        return res
    if isinstance(labels, np.ndarray) and isinstance(preds, np.ndarray) and labels.ndim == 2 \
            and preds.shape == labels.shape:
        fast = _grouped_rank_metrics(labels, preds, metrics)
        if fast is not None:
            return fast
    for metric in metrics:
        if metric == "auc":
            res["auc"] = round(roc_auc(np.asarray(labels), np.asarray(preds)), 4)
        elif metric == "rmse":
            mse = float(np.mean((np.asarray(labels, np.float64) - np.asarray(preds, np.float64)) ** 2))
            res["rmse"] = np.sqrt(round(mse, 4))
        elif metric == "logloss":
            clipped = np.clip(np.asarray(preds, dtype=np.float64), 10e-12, 1.0 - 10e-12)
            preds = clipped
            res["logloss"] = round(_log_loss(labels, clipped), 4)
        elif metric == "acc":
            pred = (np.asarray(preds) >= 0.5).astype(np.float64)
            res["acc"] = round(float(np.mean(pred == np.asarray(labels))), 4)
        elif metric == "f1":
            # sklearn.metrics.f1_score(labels, pred >= 0.5), binary: 2TP / (2TP + FP + FN); 0 when undefined
            pred = np.asarray(preds).reshape(-1) >= 0.5
            lab = np.asarray(labels).reshape(-1) == 1
            tp = float(np.sum(pred & lab))
            den = 2.0 * tp + float(np.sum(pred & ~lab)) + float(np.sum(~pred & lab))
            res["f1"] = round(tp * 2.0 / den if den > 0 else 0.0, 4)
        elif metric == "mean_mrr":
            res["mean_mrr"] = round(
                float(np.mean([mrr_score(l, p) for l, p in zip(labels, preds)])), 4
            )
        elif metric.startswith("ndcg"):
            for k in _ks(metric):
                res["ndcg@{0}".format(k)] = round(
                    float(np.mean([ndcg_score(l, p, k) for l, p in zip(labels, preds)])), 4
                )
        elif metric.startswith("hit"):
            for k in _ks(metric):
                res["hit@{0}".format(k)] = round(
                    float(np.mean([hit_score(l, p, k) for l, p in zip(labels, preds)])), 4
                )
        elif metric == "group_auc":
            res["group_auc"] = round(
                float(np.mean([roc_auc(l, p) for l, p in zip(labels, preds)])), 4
            )
        else:
            raise ValueError("not define this metric {0}".format(metric))
    return res


def _group_by_user(users, preds, labels):
    users = np.asarray(users).reshape(-1)
    preds = np.asarray(preds, dtype=np.float64).reshape(-1)
    labels = np.asarray(labels).reshape(-1)
    order = np.argsort(users, kind="mergesort")  # stable: keeps file order inside a user
    u = users[order]
    starts = np.flatnonzero(np.concatenate(([True], u[1:] != u[:-1])))
    ends = np.concatenate((starts[1:], [u.size]))
    p, l = preds[order], labels[order]
    return [(p[a:b], l[a:b]) for a, b in zip(starts, ends)], (ends - starts) / float(u.size)


def cal_weighted_metric(users, preds, labels, metrics):
    """Per-user metrics weighted by each user's line count (ref ``:702-806``).

    ``wauc`` is the README's "GAUC".  Like the reference (pandas groupby +
    ``roc_auc_score``) a user whose lines are all one class raises ``ValueError``.
    """
    res = {}
    if not metrics:
        return res
    groups = None
    for metric in metrics:
        if metric == "wauc":
            vals, weight = _segmented_auc(users, preds, labels)
            res["wauc"] = round(float((weight * vals).sum()), 4)
            continue
        if groups is None:
            groups, weight = _group_by_user(users, preds, labels)
        if False:
            pass
        elif metric == "wmrr":
            vals = np.array([mrr_score(l, p) for p, l in groups])
            res["wmrr"] = round(float((weight * vals).sum()), 4)
        elif metric.startswith("whit"):
            ks = _ks(metric)
            vals = np.array([[hit_score(l, p, k) for k in ks] for p, l in groups], dtype=np.float64)
            tot = (weight[:, None] * vals).sum(0)
            for i, k in enumerate(ks):
                res["whit@{0}".format(k)] = round(float(tot[i]), 4)
        elif metric.startswith("wndcg"):
            ks = _ks(metric)
            vals = np.array([[ndcg_score(l, p, k) for k in ks] for p, l in groups], dtype=np.float64)
            tot = (weight[:, None] * vals).sum(0)
            for i, k in enumerate(ks):
                res["wndcg@{0}".format(k)] = round(float(tot[i]), 4)
        else:
            raise ValueError("not define this metric {0}".format(metric))
    return res


def cal_mean_alpha_metric(alphas, labels):
    """Mean fusion weight over positive lines (ref ``:809-813``)."""
    alphas = np.asarray(alphas)
    labels = np.asarray(labels)
    return {"mean_alpha": round(float((alphas * labels).sum() / labels.sum()), 4)}

"""CPU oracle for the CLSR training / scoring step  --  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``clsr_amd``) never does and fails loudly when the
HIP extension is missing.

PARITY UNPINNED.  The arithmetic of the reference's hot path lives in TensorFlow 1.15.2
(pinned only by the sentence at reference ``README.md:7``; no lock file), which cannot be
installed or run in the build container (no network, Python 3.10).  The reference ships no
tests, golden vectors or checkpoints for this path (``tests/__init__.py`` is empty;
``.MISSING_LARGE_BLOBS`` lists the pretrained model).  This file therefore *restates*, in
torch-CPU, the graph the reference builds, following these reference files line by line:

  models/sequential/clsr.py:22-82      losses            -> :func:`losses`
  models/sequential/clsr.py:84-135     tables + lookups  -> :func:`forward` (prologue)
  models/sequential/clsr.py:137-277    _build_seq_graph  -> :func:`forward`
  models/sequential/clsr.py:343-381    _attention_fcn    -> :func:`attention_fcn`
  models/sequential/sequential_base_model.py:55-74,354-461   embeddings, logit MLP, _add_norm
  models/base_model.py:89-109,118-159,215-247,281-297,627-708  pred, regularisers, softmax loss,
                                                               clip+apply, _fcn_net (MLP + BN)
  models/sequential/rnn_cell_implement.py:129-298             Time4LSTMCell.call -> :func:`time4lstm`
and the TF-1.15 semantics listed in SURVEY.md section 8c (GRUCell gate order r,u with gate bias
init 1; dynamic_rnn zero output / state copy-through past sequence_length; non-fused
batch_normalization over all-but-last axes with biased variance, momentum 0.95, eps 1e-4;
softmax mask constant -(2**32)+1; AdamOptimizer dense apply for IndexedSlices; clip_by_norm
of an IndexedSlices over its concatenated, un-deduplicated values).

What IS pinned against the real reference: the batch layout and the metrics (golden fixtures
under tests/golden/, captured by scripts/make_golden.py from the reference's own iterator
and ``cal_metric``).  Backward passes are torch autograd on this same restatement, which is
the independent check for the hand-written HIP backward kernels.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

MASK_PAD = float(-(2 ** 32) + 1)
BN_MOMENTUM = 0.95
BN_EPS = 1e-4

EMB = "sequential/embedding/"
CL = "sequential/clsr/"


# ----------------------------------------------------------------------------- parameters
def _tnormal(gen, shape, std, dtype):
    """truncated_normal: resample values beyond 2 sigma (base_model.py:161-165)."""
    x = torch.empty(shape, dtype=torch.float64)
    torch.nn.init.trunc_normal_(x, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return x.to(dtype)


def _glorot(gen, shape, dtype):
    """TF default initializer (glorot_uniform) used by the RNN cells' get_variable calls."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        fan_in, fan_out = shape[0], shape[1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


def mlp_names(scope, in_dim, sizes):
    """Variable (name, shape, kind) list of one ``_fcn_net`` (base_model.py:627-708)."""
    out = []
    last = in_dim
    for i, s in enumerate(sizes):
        out.append((scope + "nn_part/w_nn_layer%d" % i, (last, s), "w"))
        out.append((scope + "nn_part/b_nn_layer%d" % i, (s,), "zero"))
        bn = scope + "nn_part/batch_normalization" + ("" if i == 0 else "_%d" % i) + "/"
        out.append((bn + "gamma", (s,), "one"))
        out.append((bn + "beta", (s,), "zero"))
        last = s
    out.append((scope + "nn_part/w_nn_output", (last, 1), "w"))
    out.append((scope + "nn_part/b_nn_output", (1,), "zero"))
    return out


def param_specs(dims, hp):
    """Ordered (name, shape, init-kind) of every trainable variable the reference creates."""
    Vu, Vi, Vc = dims["Vu"], dims["Vi"], dims["Vc"]
    Di, Dc, Du, H = hp.item_embedding_dim, hp.cate_embedding_dim, hp.user_embedding_dim, hp.hidden_size
    D = Di + Dc
    att = list(hp.att_fcn_layer_sizes)
    specs = [
        (EMB + "user_embedding", (Vu, Du), "w"),
        (EMB + "item_embedding", (Vi, Di), "w"),
        (EMB + "cate_embedding", (Vc, Dc), "w"),
        (EMB + "user_long_embedding", (Vu, Du), "w"),
        (EMB + "user_short_embedding", (Vu, Du), "w"),
    ]
    lt = CL + "long_term/attention_fcn/"
    specs.append((lt + "attention_mat", (D, Du), "w"))
    specs += mlp_names(lt + "att_fcn/", 4 * Du, att)
    st = CL + "short_term/"
    if hp.interest_evolve:
        g = st + "short_term_intention/gru_cell/"
        specs += [(g + "gates/kernel", (D + Du, 2 * Du), "glorot"), (g + "gates/bias", (2 * Du,), "one"),
                  (g + "candidate/kernel", (D + Du, Du), "glorot"), (g + "candidate/bias", (Du,), "zero")]
    sm = hp.sequential_model
    if sm == "time4lstm":
        t = st + "time4lstm/time4lstm_cell/"      # plain RNNCell: variables live under the layer's own scope (rnn_cell_implement.py:46, TF r1.15 RNNCell.__call__ -> Layer._set_scope)
        for n in ("_time_input_w1", "_time_input_bias1", "_time_input_w2", "_time_input_bias2"):
            specs.append((t + n, (H,), "glorot"))
        specs += [(t + "_time_kernel_w1", (D, H), "glorot"), (t + "_time_kernel_t1", (H, H), "glorot"),
                  (t + "_time_bias1", (H,), "glorot"),
                  (t + "_time_kernel_w2", (D, H), "glorot"), (t + "_time_kernel_t2", (H, H), "glorot"),
                  (t + "_time_bias2", (H,), "glorot"),
                  (t + "_o_kernel_t1", (H, H), "glorot"), (t + "_o_kernel_t2", (H, H), "glorot"),
                  (t + "kernel", (D + H, 4 * H), "glorot"), (t + "bias", (4 * H,), "zero")]
    elif sm == "gru":
        g = st + "simple_gru/gru_cell/"
        specs += [(g + "gates/kernel", (D + H, 2 * H), "glorot"), (g + "gates/bias", (2 * H,), "one"),
                  (g + "candidate/kernel", (D + H, H), "glorot"), (g + "candidate/bias", (H,), "zero")]
    elif sm == "lstm":
        g = st + "simple_lstm/lstm_cell/"
        specs += [(g + "kernel", (D + H, 4 * H), "glorot"), (g + "bias", (4 * H,), "zero")]
    else:
        raise ValueError(sm)
    sa = st + "attention_fcn/"
    Qs = Du + D
    specs.append((sa + "attention_mat", (H, Qs), "w"))
    specs += mlp_names(sa + "att_fcn/", 4 * Qs, att)
    if not hp.manual_alpha:
        if hp.predict_long_short:
            g = CL + "causal2/causal2/gru_cell/"
            specs += [(g + "gates/kernel", (D + H, 2 * H), "glorot"), (g + "gates/bias", (2 * H,), "one"),
                      (g + "candidate/kernel", (D + H, H), "glorot"), (g + "candidate/bias", (H,), "zero")]
            a_in = H + 3 * D + 1
        else:
            a_in = 3 * D + 1
        specs += mlp_names(CL + "fcn_alpha/", a_in, att)
    specs += mlp_names("sequential/logit_fcn/", 2 * D, list(hp.layer_sizes))
    return specs


def init_params(dims, hp, seed=0, dtype=torch.float32, scale_dense=1.0):
    """Deterministic weights with the reference's initialiser kinds.

    ``scale_dense`` (>1) optionally widens the tnormal-initialised weights so that
    parity tests exercise non-trivial activations (tnormal sigma=0.01 makes every MLP
    nearly linear around zero).
    """
    gen = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape, kind in param_specs(dims, hp):
        if kind == "w":
            std = hp.init_value * (scale_dense if not name.startswith(EMB) else scale_dense)
            params[name] = _tnormal(gen, shape, std, dtype)
        elif kind == "glorot":
            params[name] = _glorot(gen, shape, dtype)
        elif kind == "one":
            params[name] = torch.ones(shape, dtype=dtype)
        else:
            params[name] = torch.zeros(shape, dtype=dtype)
    return params


def init_bn_state(params):
    """moving_mean = 0, moving_variance = 1 for every BN layer."""
    st = OrderedDict()
    for name, p in params.items():
        if name.endswith("/gamma"):
            base = name[: -len("gamma")]
            st[base + "moving_mean"] = torch.zeros_like(p)
            st[base + "moving_variance"] = torch.ones_like(p)
    return st


# ----------------------------------------------------------------------------- building blocks
def batch_norm(x, scope, params, bn_state, training, new_bn):
    """tf.layers.batch_normalization(momentum=0.95, epsilon=1e-4) -- non-fused path:
    moments over every axis but the last, biased variance (base_model.py:673-679)."""
    gamma, beta = params[scope + "gamma"], params[scope + "beta"]
    if training:
        axes = tuple(range(x.dim() - 1))
        mean = x.mean(dim=axes)
        var = ((x - mean) ** 2).mean(dim=axes)
        if new_bn is not None:
            new_bn[scope + "moving_mean"] = (
                bn_state[scope + "moving_mean"] * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM))
            new_bn[scope + "moving_variance"] = (
                bn_state[scope + "moving_variance"] * BN_MOMENTUM + var.detach() * (1 - BN_MOMENTUM))
    else:
        mean, var = bn_state[scope + "moving_mean"], bn_state[scope + "moving_variance"]
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


def _activate(x, name):
    if name == "relu":
        return torch.relu(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "identity":
        return x
    if name == "elu":
        return torch.nn.functional.elu(x)
    raise ValueError("this activations not defined {0}".format(name))


def fcn_net(x, sizes, scope, params, bn_state, hp, training, new_bn):
    """``_fcn_net``: (x.W + b -> BN -> act) per hidden layer, then linear to 1."""
    h = x
    for i in range(len(sizes)):
        h = h @ params[scope + "nn_part/w_nn_layer%d" % i] + params[scope + "nn_part/b_nn_layer%d" % i]
        if hp.enable_BN is True:
            bn = scope + "nn_part/batch_normalization" + ("" if i == 0 else "_%d" % i) + "/"
            h = batch_norm(h, bn, params, bn_state, training, new_bn)
        h = _activate(h, hp.activation[i])
    return h @ params[scope + "nn_part/w_nn_output"] + params[scope + "nn_part/b_nn_output"]


#: Emulation of the bf16 SPEED mode of the HIP attention block (clsr_amd/csrc/hgemm.hip, CLSRNet(precision="bf16")):
#: the same graph with the (row, step)-level activations rounded to bfloat16 where that mode stores / multiplies
#: them (straight-through for autograd).  Rounding a ReLU / batch-norm network's activations to 8 mantissa bits moves
#: some of its gradients by 10-30 % on small batches (tests/test_bf16_gpu.py prints the table), so the bf16 mode is
#: checked against THIS oracle tightly and against the exact one at the logit / loss bars only.
BF16_ATTENTION = False


def _ste_bf16(x):
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def _attention_fcn_bf16(query, keys, mask, scope, params, bn_state, hp, training, new_bn):
    """``_attention_fcn`` as the speed mode evaluates it: first layer re-associated into
    U[h,t] + V[r] + (a*q).Wp (fp32 U, V; bf16 product operands, bf16 z0), bf16 h0 / W1 / z1, everything else exact."""
    nn = scope + "att_fcn/nn_part/"
    a = keys @ params[scope + "attention_mat"]
    q = query.unsqueeze(1).expand_as(a)
    Q = a.shape[-1]
    W0 = params[nn + "w_nn_layer0"]
    Wu, Wv, Wp = W0[:Q] + W0[2 * Q:3 * Q], W0[Q:2 * Q] - W0[2 * Q:3 * Q], W0[3 * Q:]
    r = _ste_bf16
    z0 = r(r(a * q) @ r(Wp) + a @ Wu + q @ Wv + params[nn + "b_nn_layer0"])
    h0 = r(_activate(batch_norm(z0, nn + "batch_normalization/", params, bn_state, training, new_bn), hp.activation[0]))
    z1 = r(h0 @ r(params[nn + "w_nn_layer1"]) + params[nn + "b_nn_layer1"])
    h1 = _activate(batch_norm(z1, nn + "batch_normalization_1/", params, bn_state, training, new_bn), hp.activation[1])
    score = (h1 @ params[nn + "w_nn_output"] + params[nn + "b_nn_output"]).squeeze(-1)
    score = torch.where(mask == 1, score, torch.full_like(score, MASK_PAD))
    w = torch.softmax(score, dim=-1)
    return keys * w.unsqueeze(-1), w


def attention_fcn(query, keys, mask, scope, params, bn_state, hp, training, new_bn):
    """``_attention_fcn`` (clsr.py:343-381); returns keys * weights (caller sums over T)."""
    if BF16_ATTENTION:
        return _attention_fcn_bf16(query, keys, mask, scope, params, bn_state, hp, training, new_bn)
    att_inputs = keys @ params[scope + "attention_mat"]                       # [B,T,Q]
    q = query.unsqueeze(1).expand_as(att_inputs)
    feat = torch.cat([att_inputs, q, att_inputs - q, att_inputs * q], -1)    # [B,T,4Q]
    score = fcn_net(feat, list(hp.att_fcn_layer_sizes), scope + "att_fcn/", params, bn_state, hp,
                    training, new_bn).squeeze(-1)
    score = torch.where(mask == 1, score, torch.full_like(score, MASK_PAD))
    w = torch.softmax(score, dim=-1)
    return keys * w.unsqueeze(-1), w


def gru_cell(x, h, Wg, bg, Wc, bc):
    """tf.nn.rnn_cell.GRUCell: [r,u]=sigmoid([x,h]Wg+bg); c=tanh([x,r*h]Wc+bc); h'=u*h+(1-u)*c."""
    n = h.shape[-1]
    ru = torch.sigmoid(torch.cat([x, h], -1) @ Wg + bg)
    r, u = ru[..., :n], ru[..., n:]
    c = torch.tanh(torch.cat([x, r * h], -1) @ Wc + bc)
    return u * h + (1 - u) * c


#: bench.py's cpu_baseline leg sets this: the input-side products of the recurrent cells ([x, h] W = x W[:D] + h W[D:])
#: are computed for all T steps in one batched product instead of T small ones.  Same math (fp32 rounding order of
#: the two partial products differs); the parity tests run the literal per-step form above.
FAST_RNN = False


def dynamic_gru(x, seq_len, h0, scope, params):
    """dynamic_rnn(GRUCell): zero output and state copy-through for t >= sequence_length."""
    B, T, D = x.shape
    Wg, bg = params[scope + "gates/kernel"], params[scope + "gates/bias"]
    Wc, bc = params[scope + "candidate/kernel"], params[scope + "candidate/bias"]
    h = h0
    outs = []
    if FAST_RNN:
        n = h0.shape[-1]
        xg, xc = x @ Wg[:D] + bg, x @ Wc[:D] + bc
        Wgh, Wch = Wg[D:], Wc[D:]
    for t in range(T):
        if FAST_RNN:
            ru = torch.sigmoid(xg[:, t] + h @ Wgh)
            r, u = ru[..., :n], ru[..., n:]
            c = torch.tanh(xc[:, t] + (r * h) @ Wch)
            nh = u * h + (1 - u) * c
        else:
            nh = gru_cell(x[:, t], h, Wg, bg, Wc, bc)
        live = (t < seq_len).unsqueeze(-1)
        h = torch.where(live, nh, h)
        outs.append(torch.where(live, nh, torch.zeros_like(nh)))
    return torch.stack(outs, 1), h


def time4lstm(x, t_first, t_now, seq_len, scope, params, H):
    """dynamic_rnn(Time4LSTMCell) (rnn_cell_implement.py:129-298); inputs[:, -1] is
    time_to_now ("time_now_score"), inputs[:, -2] is time_from_first_action ("time_last_score")
    because of the concat order at clsr.py:180-193."""
    p = lambda n: params[scope + n]
    B, T, D = x.shape
    c = torch.zeros(B, H, dtype=x.dtype)
    m = torch.zeros(B, H, dtype=x.dtype)
    outs = []
    if FAST_RNN:   # everything that does not depend on the recurrent state, for all steps at once
        tn_all = torch.tanh(t_now.unsqueeze(-1) * p("_time_input_w1") + p("_time_input_bias1"))
        tl_all = torch.tanh(t_first.unsqueeze(-1) * p("_time_input_w2") + p("_time_input_bias2"))
        tns_all = x @ p("_time_kernel_w1") + tn_all @ p("_time_kernel_t1") + p("_time_bias1")
        tls_all = x @ p("_time_kernel_w2") + tl_all @ p("_time_kernel_t2") + p("_time_bias2")
        zx_all = x @ p("kernel")[:D] + p("bias")
        zx_all = torch.cat([zx_all[..., :3 * H],
                            zx_all[..., 3 * H:] + tn_all @ p("_o_kernel_t1") + tl_all @ p("_o_kernel_t2")], -1)
        Wm = p("kernel")[D:]
    for t in range(T):
        if FAST_RNN:
            z = zx_all[:, t] + m @ Wm
            i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
            tns, tls = tns_all[:, t], tls_all[:, t]
        else:
            xt = x[:, t]
            tn = torch.tanh(t_now[:, t:t + 1] * p("_time_input_w1") + p("_time_input_bias1"))
            tl = torch.tanh(t_first[:, t:t + 1] * p("_time_input_w2") + p("_time_input_bias2"))
            tns = xt @ p("_time_kernel_w1") + tn @ p("_time_kernel_t1") + p("_time_bias1")
            tls = xt @ p("_time_kernel_w2") + tl @ p("_time_kernel_t2") + p("_time_bias2")
            z = torch.cat([xt, m], -1) @ p("kernel") + p("bias")
            i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
            o = o + tn @ p("_o_kernel_t1") + tl @ p("_o_kernel_t2")
        nc = torch.sigmoid(f + 1.0) * torch.sigmoid(tls) * c + torch.sigmoid(i) * torch.sigmoid(tns) * torch.tanh(j)
        nm = torch.sigmoid(o) * torch.tanh(nc)
        live = (t < seq_len).unsqueeze(-1)
        c = torch.where(live, nc, c)
        m = torch.where(live, nm, m)
        outs.append(torch.where(live, nm, torch.zeros_like(nm)))
    return torch.stack(outs, 1)


def dynamic_lstm(x, seq_len, scope, params, H):
    """dynamic_rnn(tf.nn.rnn_cell.LSTMCell) with forget_bias=1, no peepholes."""
    B, T, _ = x.shape
    W, b = params[scope + "kernel"], params[scope + "bias"]
    c = torch.zeros(B, H, dtype=x.dtype)
    m = torch.zeros(B, H, dtype=x.dtype)
    outs = []
    for t in range(T):
        z = torch.cat([x[:, t], m], -1) @ W + b
        i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        nc = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
        nm = torch.sigmoid(o) * torch.tanh(nc)
        live = (t < seq_len).unsqueeze(-1)
        c = torch.where(live, nc, c)
        m = torch.where(live, nm, m)
        outs.append(torch.where(live, nm, torch.zeros_like(nm)))
    return torch.stack(outs, 1)


def _unique(ids):
    return torch.unique(ids.reshape(-1))


# ----------------------------------------------------------------------------- forward
def to_torch_feed(feed, dtype=torch.float32):
    """numpy feed (iterator layout) -> torch tensors with the placeholder dtypes
    (sequential_iterator.py:48-70: users/items/cates/histories/mask int32, the rest float32)."""
    out = {}
    for k, v in feed.items():
        v = np.asarray(v)
        if k in ("users", "items", "cates", "item_history", "item_cate_history", "mask"):
            out[k] = torch.from_numpy(v.astype(np.int64))
        else:
            out[k] = torch.from_numpy(v.astype(np.float64)).to(dtype)
    return out


def forward(params, bn_state, feed, hp, training, new_bn=None, sites=None):
    """Forward pass of ``SequentialBaseModel._build_graph`` + ``CLSRModel._build_seq_graph``.

    ``sites`` (optional dict) receives the gathered embedding tensors (one entry per
    ``tf.nn.embedding_lookup`` call site that reaches the loss) so that their autograd
    gradients can stand in for the reference's ``IndexedSlices`` values.
    Returns a dict with logit, pred, alpha and every intermediate the tests compare.
    """
    D_i, D_c = hp.item_embedding_dim, hp.cate_embedding_dim
    H = hp.hidden_size
    users, items, cates = feed["users"], feed["items"], feed["cates"]
    ih, ch, mask = feed["item_history"], feed["item_cate_history"], feed["mask"]
    item_tbl, cate_tbl = params[EMB + "item_embedding"], params[EMB + "cate_embedding"]
    ul_tbl, us_tbl = params[EMB + "user_long_embedding"], params[EMB + "user_short_embedding"]

    def site(name, tbl, idx):
        g = tbl[idx]
        if sites is not None:
            if g.requires_grad:
                g.retain_grad()
            sites[name] = (g, idx)
        return g

    item_emb = site("item/target", item_tbl, items)
    item_hist = site("item/history", item_tbl, ih)
    cate_emb = site("cate/target", cate_tbl, cates)
    cate_hist = site("cate/history", cate_tbl, ch)
    inv_items = _unique(torch.cat([ih.reshape(-1), items.reshape(-1)]))
    inv_cates = _unique(torch.cat([ch.reshape(-1), cates.reshape(-1)]))
    inv_users = _unique(users)
    inv_item_emb = site("item/involved", item_tbl, inv_items)
    inv_cate_emb = site("cate/involved", cate_tbl, inv_cates)
    u_long = site("user_long/row", ul_tbl, users)
    u_short = site("user_short/row", us_tbl, users)
    inv_ul = site("user_long/involved", ul_tbl, inv_users)
    inv_us = site("user_short/involved", us_tbl, inv_users)
    target = torch.cat([item_emb, cate_emb], -1)
    # dropout keep_prob = 1 - embedding_dropout; the only supported value is keep 1.0 (identity)
    hist_input = torch.cat([item_hist, cate_hist], 2)
    real_mask = mask.to(hist_input.dtype)
    seq_len = mask.sum(1)

    out = {}
    # ---- long term (clsr.py:152-157)
    lt = CL + "long_term/attention_fcn/"
    att_long_seq, w_long = attention_fcn(u_long, hist_input, mask, lt, params, bn_state, hp, training, new_bn)
    att_fea_long = att_long_seq.sum(1)
    hist_mean = (hist_input * real_mask.unsqueeze(-1)).sum(1) / real_mask.sum(1, keepdim=True)

    # ---- short term (clsr.py:159-222)
    st = CL + "short_term/"
    if hp.interest_evolve:
        _, short_int = dynamic_gru(hist_input, seq_len, u_short, st + "short_term_intention/gru_cell/", params)
    else:
        short_int = u_short
    position = torch.flip(torch.cumsum(torch.flip(real_mask, [1]), 1), [1])
    recent = ((position >= 1) & (position <= hp.contrastive_recent_k)).to(hist_input.dtype)
    hist_recent = (hist_input * recent.unsqueeze(-1)).sum(1) / recent.sum(1, keepdim=True)
    if hp.sequential_model == "time4lstm":
        rnn_out = time4lstm(hist_input, feed["time_from_first_action"], feed["time_to_now"], seq_len,
                            st + "time4lstm/time4lstm_cell/", params, H)
    elif hp.sequential_model == "gru":
        rnn_out, _ = dynamic_gru(hist_input, seq_len, torch.zeros(hist_input.shape[0], H, dtype=hist_input.dtype),
                                 st + "simple_gru/gru_cell/", params)
    else:
        rnn_out = dynamic_lstm(hist_input, seq_len, st + "simple_lstm/lstm_cell/", params, H)
    sq = torch.cat([short_int, target], -1)
    att_short_seq, w_short = attention_fcn(sq, rnn_out, mask, st + "attention_fcn/", params, bn_state, hp,
                                           training, new_bn)
    att_fea_short = att_short_seq.sum(1)

    # ---- alpha fusion (clsr.py:227-274)
    if not hp.manual_alpha:
        last_tnow = feed["time_to_now"][:, -1:]
        if hp.predict_long_short:
            _, final_state = dynamic_gru(hist_input, seq_len,
                                         torch.zeros(hist_input.shape[0], H, dtype=hist_input.dtype),
                                         CL + "causal2/causal2/gru_cell/", params)
            concat_all = torch.cat([final_state, target, att_fea_long, att_fea_short, last_tnow], 1)
            out["causal_state"] = final_state
        else:
            concat_all = torch.cat([target, att_fea_long, att_fea_short, last_tnow], 1)
        alpha_logit = fcn_net(concat_all, list(hp.att_fcn_layer_sizes), CL + "fcn_alpha/", params, bn_state,
                              hp, training, new_bn)
        alpha = torch.sigmoid(alpha_logit)
        user_embed = att_fea_long * alpha + att_fea_short * (1.0 - alpha)
    else:
        alpha = torch.full((1, 1), float(hp.manual_alpha_value), dtype=hist_input.dtype)
        user_embed = att_fea_long * hp.manual_alpha_value + att_fea_short * (1.0 - hp.manual_alpha_value)
    model_output = torch.cat([user_embed, target], 1)
    logit = fcn_net(model_output, list(hp.layer_sizes), "sequential/logit_fcn/", params, bn_state, hp,
                    training, new_bn)
    out.update(dict(
        logit=logit, pred=torch.sigmoid(logit), alpha=alpha, hist_input=hist_input, target=target,
        u_long=u_long, u_short=u_short, att_fea_long=att_fea_long, att_fea_short=att_fea_short,
        hist_mean=hist_mean, hist_recent=hist_recent, short_intention=short_int, rnn_out=rnn_out,
        w_long=w_long, w_short=w_short, seq_len=seq_len, user_embed=user_embed,
        involved=dict(item=inv_item_emb, cate=inv_cate_emb, user_long=inv_ul, user_short=inv_us),
    ))
    return out


# ----------------------------------------------------------------------------- losses
def losses(params, out, feed, hp):
    """data + regular + contrastive + discrepancy (clsr.py:22-82, base_model.py:118-159,215-247)."""
    group = hp.train_num_ngs + 1
    logits = out["logit"].reshape(-1, group)
    labels = feed["labels"].reshape(-1, group)
    sm = torch.softmax(logits, -1)
    pos = torch.where(labels == 1, sm, torch.ones_like(sm))
    data_loss = -group * torch.log(pos).mean()

    l2 = lambda t: (t ** 2).sum() / 2
    reg = out["logit"].new_zeros(())
    inv = out["involved"]
    for k in ("item", "cate", "user_long", "user_short"):
        reg = reg + hp.embed_l2 * l2(inv[k]) + hp.embed_l1 * inv[k].abs().sum()
    for name, p in params.items():
        if not name.startswith(EMB):
            reg = reg + hp.layer_l2 * l2(p) + hp.layer_l1 * p.abs().sum()

    cmask = (out["seq_len"] > hp.contrastive_length_threshold).to(out["logit"].dtype)
    L, S, M, R = out["att_fea_long"], out["att_fea_short"], out["hist_mean"], out["hist_recent"]
    denom = cmask.sum()
    if hp.contrastive_loss == "bpr":
        sp = torch.nn.functional.softplus
        terms = [sp((L * (-M + R)).sum(-1)), sp((S * (-R + M)).sum(-1)),
                 sp((M * (-L + S)).sum(-1)), sp((R * (-S + L)).sum(-1))]
    elif hp.contrastive_loss == "triplet":
        mg = hp.triplet_margin
        dLM, dLR, dSM, dSR = (L - M) ** 2, (L - R) ** 2, (S - M) ** 2, (S - R) ** 2
        relu = torch.relu
        terms = [relu(dLM - dLR + mg).sum(-1), relu(dSR - dSM + mg).sum(-1),
                 relu(dLM - dSM + mg).sum(-1), relu(dSR - dLR + mg).sum(-1)]
    else:
        raise ValueError(hp.contrastive_loss)
    contrastive = sum((cmask * t).sum() / denom for t in terms) * hp.contrastive_loss_weight
    discrepancy = -hp.discrepancy_loss_weight * ((inv["user_long"].reshape(-1) - inv["user_short"].reshape(-1)) ** 2).mean()
    total = data_loss + reg + contrastive + discrepancy
    return dict(loss=total, data_loss=data_loss, regular_loss=reg, contrastive_loss=contrastive,
                discrepancy_loss=discrepancy)


# ----------------------------------------------------------------------------- training step
TABLES = {"item": EMB + "item_embedding", "cate": EMB + "cate_embedding",
          "user_long": EMB + "user_long_embedding", "user_short": EMB + "user_short_embedding"}


def _clip_factor(sumsq, clip_norm):
    """tf.clip_by_norm: t * clip_norm / max(||t||, clip_norm)."""
    norm = math.sqrt(sumsq) if sumsq > 0 else 0.0
    return clip_norm / max(norm, clip_norm)


def gradients(params, bn_state, feed, hp):
    """Loss + per-variable gradients with the reference's clipping semantics
    (base_model.py:281-297).  Embedding tables: the gradient is an IndexedSlices whose
    values are the concatenation of every lookup site's slice; ``clip_by_norm`` uses the
    norm of those un-deduplicated values.  Returns (loss dict, dense grads dict,
    clip-norm dict, new_bn, forward outputs)."""
    leaf = OrderedDict((k, v.detach().clone().requires_grad_(not k.endswith("/user_embedding")))
                       for k, v in params.items())
    new_bn = OrderedDict()
    sites = {}
    out = forward(leaf, bn_state, feed, hp, True, new_bn, sites)
    ls = losses(leaf, out, feed, hp)
    ls["loss"].backward()
    grads, norms, raw = OrderedDict(), OrderedDict(), OrderedDict()
    table_names = set(TABLES.values())
    for name, p in leaf.items():
        if name.endswith("/user_embedding"):
            continue
        if name in table_names:
            key = [k for k, v in TABLES.items() if v == name][0]
            sumsq = 0.0
            for sname, (g, idx) in sites.items():
                if sname.startswith(key + "/") and g.grad is not None:
                    sumsq += float((g.grad.double() ** 2).sum())
            norms[name] = math.sqrt(sumsq)
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
        else:
            gr = p.grad
            norms[name] = float(gr.double().norm())
        raw[name] = gr.detach().clone()
        if hp.is_clip_norm:
            gr = gr * _clip_factor(norms[name] ** 2, float(hp.max_grad_norm))
        grads[name] = gr.detach()
    out["raw_grads"] = raw
    return {k: v.detach() for k, v in ls.items()}, grads, norms, new_bn, out


def init_adam(params):
    st = OrderedDict()
    for k, v in params.items():
        if k.endswith("/user_embedding"):
            continue
        st[k] = (torch.zeros_like(v), torch.zeros_like(v))
    return st


def adam_apply(params, grads, adam, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, lazy_rows=None):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); var -= lr_t*m/(sqrt(v)+eps).
    For IndexedSlices the TF op decays m, v over the WHOLE table and updates every row, which
    equals this dense update with zero gradient on untouched rows (base_model.py:263-264)."""
    lr_t = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    new_p, new_a = OrderedDict(), OrderedDict()
    for k, v in params.items():
        if k not in grads:
            new_p[k] = v
            continue
        m, s = adam[k]
        g = grads[k]
        m2 = beta1 * m + (1 - beta1) * g
        s2 = beta2 * s + (1 - beta2) * g * g
        new_a[k] = (m2, s2)
        new_p[k] = v - lr_t * m2 / (torch.sqrt(s2) + eps)
    return new_p, new_a


def train_step(params, bn_state, adam, step, feed, hp):
    """One ``CLSRModel.train`` call: forward, backward, per-tensor clip, Adam, BN moving stats."""
    ls, grads, norms, new_bn, out = gradients(params, bn_state, feed, hp)
    new_params, new_adam = adam_apply(params, grads, adam, step, hp.learning_rate)
    bn2 = OrderedDict(bn_state)
    bn2.update(new_bn)
    return new_params, bn2, new_adam, ls, grads, norms, out


@torch.no_grad()
def predict(params, bn_state, feed, hp):
    """``eval_with_user`` / ``infer``: forward with moving BN statistics."""
    return forward(params, bn_state, feed, hp, False)

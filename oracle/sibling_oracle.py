"""CPU oracle for the sibling models that share CLSR's kernels  --  TEST INFRASTRUCTURE ONLY (see clsr_oracle.py:
same rules, same "PARITY UNPINNED" status: the arithmetic lives in TensorFlow 1.15, the reference ships no vectors).

torch-CPU restatement of ``_build_seq_graph`` of

  GRU4RecModel   models/sequential/gru4rec.py:21-76     dynamic_rnn(GRUCell) final state ++ target
  DINModel       models/sequential/din.py:13-34          target ++ masked history sum ++ _attention_fcn(target, history)
  A2SVDModel     models/sequential/asvd.py:13-45         A2SVD attention (base_model.py:595-625) ++ target
  DIENModel      models/sequential/dien.py:13-64         GRU, _attention_fcn weights, attentional GRU (VecAttGRUCell)
                                                         final state ++ target ++ history sum ++ their product
  SLI_RECModel   models/sequential/sli_rec.py:25-147     A2SVD attention (base_model.py:595-625, UNMASKED softmax over
                                                         all T steps), Time4LSTM over [item emb, t_first, t_now],
                                                         _attention_fcn(target, rnn_outputs), alpha fusion

on top of ``SequentialBaseModel`` (sequential_base_model.py:55-74,354-452: embeddings, involved-row lists, logit MLP)
and ``BaseModel`` (base_model.py:78-87 loss = data + regular; :215-235 softmax loss; :281-297 clip + Adam).  Building
blocks (MLP + BN, attention_fcn, GRU, Time4LSTM, Adam, clip semantics) are the ones of clsr_oracle.py; backward passes
are torch autograd on this restatement.
"""
import math
from collections import OrderedDict

import torch

from oracle import clsr_oracle as C

EMB = C.EMB
TABLES = {"item": EMB + "item_embedding", "cate": EMB + "cate_embedding"}


def _mlp(scope, in_dim, sizes):
    return C.mlp_names(scope, in_dim, sizes)


def param_specs(dims, hp, kind):
    Vu, Vi, Vc = dims["Vu"], dims["Vi"], dims["Vc"]
    Di, Dc, Du, H = hp.item_embedding_dim, hp.cate_embedding_dim, hp.user_embedding_dim, hp.hidden_size
    D = Di + Dc
    att = list(hp.att_fcn_layer_sizes) if kind in ("din", "sli_rec", "dien") else None
    specs = [(EMB + "user_embedding", (Vu, Du), "w"), (EMB + "item_embedding", (Vi, Di), "w"),
             (EMB + "cate_embedding", (Vc, Dc), "w")]

    def gru(scope, i, n):
        return [(scope + "gates/kernel", (i + n, 2 * n), "glorot"), (scope + "gates/bias", (2 * n,), "one"),
                (scope + "candidate/kernel", (i + n, n), "glorot"), (scope + "candidate/bias", (n,), "zero")]

    if kind == "gru4rec":
        specs += gru("sequential/gru4rec/gru/gru_cell/", D, H)
        out_dim = H + D
    elif kind == "dien":
        specs += gru("sequential/gru1/gru_cell/", D, H)
        specs.append(("sequential/attention_fcn/attention_mat", (H, D), "w"))
        specs += _mlp("sequential/attention_fcn/att_fcn/", 4 * D, list(hp.att_fcn_layer_sizes))
        specs += gru("sequential/gru2/vec_att_gru_cell/", H, H)
        out_dim = 3 * D + H
    elif kind == "a2svd":
        specs += [("sequential/a2svd/Attention_layer/attention_mat", (D, D), "w"),
                  ("sequential/a2svd/Attention_layer/query", (hp.attention_size,), "w")]
        out_dim = 2 * D
    elif kind == "din":
        specs.append(("sequential/attention_fcn/attention_mat", (D, D), "w"))
        specs += _mlp("sequential/attention_fcn/att_fcn/", 4 * D, att)
        out_dim = 3 * D
    elif kind == "sli_rec":
        s = "sequential/sli_rec/"
        specs += [(s + "long_term_asvd/attention_mat", (D, D), "w"),
                  (s + "long_term_asvd/query", (hp.attention_size,), "w")]
        t = s + "rnn/time4lstm/time4lstm_cell/"
        for n_ in ("_time_input_w1", "_time_input_bias1", "_time_input_w2", "_time_input_bias2"):
            specs.append((t + n_, (H,), "glorot"))
        specs += [(t + "_time_kernel_w1", (Di, H), "glorot"), (t + "_time_kernel_t1", (H, H), "glorot"),
                  (t + "_time_bias1", (H,), "glorot"),
                  (t + "_time_kernel_w2", (Di, H), "glorot"), (t + "_time_kernel_t2", (H, H), "glorot"),
                  (t + "_time_bias2", (H,), "glorot"),
                  (t + "_o_kernel_t1", (H, H), "glorot"), (t + "_o_kernel_t2", (H, H), "glorot"),
                  (t + "kernel", (Di + H, 4 * H), "glorot"), (t + "bias", (4 * H,), "zero")]
        a = s + "attention_fcn/attention_fcn/"
        specs.append((a + "attention_mat", (H, D), "w"))
        specs += _mlp(a + "att_fcn/", 4 * D, att)
        if not hp.manual_alpha:
            specs += _mlp(s + "fcn_alpha/", 3 * D + 1, att)
        out_dim = 2 * D
    else:
        raise ValueError(kind)
    specs += _mlp("sequential/logit_fcn/", out_dim, list(hp.layer_sizes))
    return specs


def init_params(dims, hp, kind, seed=0, dtype=torch.float32, scale_dense=1.0):
    gen = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape, k in param_specs(dims, hp, kind):
        if k == "w":
            params[name] = C._tnormal(gen, shape, hp.init_value * scale_dense, dtype)
        elif k == "glorot":
            params[name] = C._glorot(gen, shape, dtype)
        elif k == "one":
            params[name] = torch.ones(shape, dtype=dtype)
        else:
            params[name] = torch.zeros(shape, dtype=dtype)
    return params


init_bn_state = C.init_bn_state
to_torch_feed = C.to_torch_feed
init_adam = C.init_adam
adam_apply = C.adam_apply


def asvd_attention(x, scope, params):
    """``BaseModel._attention`` (base_model.py:595-625): softmax over the T axis WITHOUT a mask."""
    att_inputs = x @ params[scope + "attention_mat"]
    logits = att_inputs @ params[scope + "query"]
    w = torch.softmax(logits, dim=-1)
    return x * w.unsqueeze(-1), w


def dynamic_augru(x, att, seq_len, scope, params, H):
    """dynamic_rnn(VecAttGRUCell) (rnn_cell_implement.py:594-623, rnn_dien.py): a GRU whose update gate is scaled by
    (1 - att_score) of the step; zero output / state copy-through past sequence_length."""
    Wg, bg = params[scope + "gates/kernel"], params[scope + "gates/bias"]
    Wc, bc = params[scope + "candidate/kernel"], params[scope + "candidate/bias"]
    h = torch.zeros(x.shape[0], H, dtype=x.dtype)
    for t in range(x.shape[1]):
        ru = torch.sigmoid(torch.cat([x[:, t], h], -1) @ Wg + bg)
        r, u = ru[:, :H], ru[:, H:]
        c = torch.tanh(torch.cat([x[:, t], r * h], -1) @ Wc + bc)
        u = (1.0 - att[:, t:t + 1]) * u
        nh = u * h + (1 - u) * c
        h = torch.where((t < seq_len).unsqueeze(-1), nh, h)
    return h


def forward(params, bn_state, feed, hp, kind, training, new_bn=None, sites=None):
    H = hp.hidden_size
    items, cates = feed["items"], feed["cates"]
    ih, ch, mask = feed["item_history"], feed["item_cate_history"], feed["mask"]
    item_tbl, cate_tbl = params[EMB + "item_embedding"], params[EMB + "cate_embedding"]

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
    inv_item = site("item/involved", item_tbl, C._unique(torch.cat([ih.reshape(-1), items.reshape(-1)])))
    inv_cate = site("cate/involved", cate_tbl, C._unique(torch.cat([ch.reshape(-1), cates.reshape(-1)])))
    target = torch.cat([item_emb, cate_emb], -1)
    hist = torch.cat([item_hist, cate_hist], 2)
    real_mask = mask.to(hist.dtype)
    seq_len = mask.sum(1)
    out = {}
    if kind == "gru4rec":
        _, final = C.dynamic_gru(hist, seq_len, torch.zeros(hist.shape[0], H, dtype=hist.dtype),
                                 "sequential/gru4rec/gru/gru_cell/", params)
        model_output = torch.cat([final, target], 1)
        out["final_state"] = final
    elif kind == "dien":
        hist_sum = (hist * real_mask.unsqueeze(-1)).sum(1)
        rnn1, _ = C.dynamic_gru(hist, seq_len, torch.zeros(hist.shape[0], H, dtype=hist.dtype),
                                "sequential/gru1/gru_cell/", params)
        _, alphas = C.attention_fcn(target, rnn1, mask, "sequential/attention_fcn/", params, bn_state, hp, training,
                                    new_bn)
        final = dynamic_augru(rnn1, alphas, seq_len, "sequential/gru2/vec_att_gru_cell/", params, H)
        model_output = torch.cat([target, final, hist_sum, target * hist_sum], 1)
        out.update(hist_sum=hist_sum, rnn_out=rnn1, w_att=alphas, final_state=final)
    elif kind == "a2svd":
        a_seq, w = asvd_attention(hist, "sequential/a2svd/Attention_layer/", params)
        asvd_output = a_seq.sum(1)
        model_output = torch.cat([asvd_output, target], 1)
        out.update(asvd_output=asvd_output, w_asvd=w)
    elif kind == "din":
        hist_sum = (hist * real_mask.unsqueeze(-1)).sum(1)
        att_seq, w = C.attention_fcn(target, hist, mask, "sequential/attention_fcn/", params, bn_state, hp,
                                     training, new_bn)
        att_fea = att_seq.sum(1)
        model_output = torch.cat([target, hist_sum, att_fea], -1)
        out.update(hist_sum=hist_sum, att_fea=att_fea, w_att=w)
    elif kind == "sli_rec":
        s = "sequential/sli_rec/"
        a1_seq, w1 = asvd_attention(hist, s + "long_term_asvd/", params)
        att_fea1 = a1_seq.sum(1)
        rnn_out = C.time4lstm(item_hist, feed["time_from_first_action"], feed["time_to_now"], seq_len,
                              s + "rnn/time4lstm/time4lstm_cell/", params, H)
        a2_seq, w2 = C.attention_fcn(target, rnn_out, mask, s + "attention_fcn/attention_fcn/", params, bn_state, hp,
                                     training, new_bn)
        att_fea2 = a2_seq.sum(1)
        if not hp.manual_alpha:
            concat_all = torch.cat([target, att_fea1, att_fea2, feed["time_to_now"][:, -1:]], 1)
            alpha = torch.sigmoid(C.fcn_net(concat_all, list(hp.att_fcn_layer_sizes), s + "fcn_alpha/", params,
                                            bn_state, hp, training, new_bn))
            user_embed = att_fea1 * alpha + att_fea2 * (1.0 - alpha)
        else:
            alpha = torch.full((1, 1), float(hp.manual_alpha_value), dtype=hist.dtype)
            user_embed = att_fea1 * hp.manual_alpha_value + att_fea2 * (1.0 - hp.manual_alpha_value)
        model_output = torch.cat([user_embed, target], 1)
        out.update(att_fea1=att_fea1, att_fea2=att_fea2, rnn_out=rnn_out, alpha=alpha, w_asvd=w1, w_att=w2)
    else:
        raise ValueError(kind)
    logit = C.fcn_net(model_output, list(hp.layer_sizes), "sequential/logit_fcn/", params, bn_state, hp, training,
                      new_bn)
    out.update(logit=logit, pred=torch.sigmoid(logit), model_output=model_output, target=target, hist_input=hist,
               seq_len=seq_len, involved=dict(item=inv_item, cate=inv_cate))
    return out


def losses(params, out, feed, hp):
    """``BaseModel._get_loss``: softmax data loss + L2 of the involved embedding rows and of every other variable."""
    group = hp.train_num_ngs + 1
    logits = out["logit"].reshape(-1, group)
    labels = feed["labels"].reshape(-1, group)
    sm = torch.softmax(logits, -1)
    data_loss = -group * torch.log(torch.where(labels == 1, sm, torch.ones_like(sm))).mean()
    l2 = lambda t: (t ** 2).sum() / 2
    reg = out["logit"].new_zeros(())
    for k in ("item", "cate"):
        reg = reg + hp.embed_l2 * l2(out["involved"][k])
    for name, p in params.items():
        if not name.startswith(EMB):
            reg = reg + hp.layer_l2 * l2(p)
    return dict(loss=data_loss + reg, data_loss=data_loss, regular_loss=reg)


def gradients(params, bn_state, feed, hp, kind):
    leaf = OrderedDict((k, v.detach().clone().requires_grad_(not k.endswith("/user_embedding")))
                       for k, v in params.items())
    new_bn, sites = OrderedDict(), {}
    out = forward(leaf, bn_state, feed, hp, kind, True, new_bn, sites)
    ls = losses(leaf, out, feed, hp)
    ls["loss"].backward()
    grads, norms, raw = OrderedDict(), OrderedDict(), OrderedDict()
    tables = set(TABLES.values())
    for name, p in leaf.items():
        if name.endswith("/user_embedding"):
            continue
        if name in tables:
            key = [k for k, v in TABLES.items() if v == name][0]
            sumsq = sum(float((g.grad.double() ** 2).sum()) for sname, (g, _) in sites.items()
                        if sname.startswith(key + "/") and g.grad is not None)
            norms[name] = math.sqrt(sumsq)
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
        else:
            gr = p.grad
            norms[name] = float(gr.double().norm())
        raw[name] = gr.detach().clone()
        if hp.is_clip_norm:
            gr = gr * C._clip_factor(norms[name] ** 2, float(hp.max_grad_norm))
        grads[name] = gr.detach()
    out["raw_grads"] = raw
    return {k: v.detach() for k, v in ls.items()}, grads, norms, new_bn, out


def train_step(params, bn_state, adam, step, feed, hp, kind):
    ls, grads, norms, new_bn, out = gradients(params, bn_state, feed, hp, kind)
    new_params, new_adam = adam_apply(params, grads, adam, step, hp.learning_rate)
    bn2 = OrderedDict(bn_state)
    bn2.update(new_bn)
    return new_params, bn2, new_adam, ls, grads, norms, out


@torch.no_grad()
def predict(params, bn_state, feed, hp, kind):
    return forward(params, bn_state, feed, hp, kind, False)

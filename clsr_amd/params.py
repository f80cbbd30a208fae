"""Variable inventory of the CLSR graph: names (the reference's TF variable names), shapes and
initialiser kinds.

Restated from the reference's graph construction:
  embeddings      models/sequential/sequential_base_model.py:354-379, clsr.py:84-101
  attention_fcn   clsr.py:343-381 (attention_mat) + models/base_model.py:627-708 (_fcn_net: w/b per
                  layer, tf.layers.batch_normalization gamma/beta, w_nn_output/b_nn_output)
  GRUCell         gates/{kernel,bias}, candidate/{kernel,bias}; gate bias init 1.0 (TF 1.15)
  Time4LSTMCell   models/sequential/rnn_cell_implement.py:150-230
Initialisers: the "embedding" and "nn_part" scopes and attention_mat use hparams.init_method
(tnormal, sigma = init_value; base_model.py:161-189); RNN cell variables fall back to TF's
default glorot_uniform; biases zero; BN gamma one / beta zero.
"""
import math
from collections import OrderedDict

import torch

EMB = "sequential/embedding/"
CL = "sequential/clsr/"

TABLES = OrderedDict([
    ("item", EMB + "item_embedding"),
    ("cate", EMB + "cate_embedding"),
    ("user_long", EMB + "user_long_embedding"),
    ("user_short", EMB + "user_short_embedding"),
])
UNUSED_TABLE = EMB + "user_embedding"  # created and gathered by the reference, never reaches the loss


def mlp_specs(scope, in_dim, sizes):
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


def gru_specs(scope, in_dim, n):
    return [(scope + "gates/kernel", (in_dim + n, 2 * n), "glorot"), (scope + "gates/bias", (2 * n,), "one"),
            (scope + "candidate/kernel", (in_dim + n, n), "glorot"), (scope + "candidate/bias", (n,), "zero")]


def param_specs(dims, hp):
    """Ordered list of (name, shape, init kind) for every variable of the CLSR graph."""
    Vu, Vi, Vc = dims["Vu"], dims["Vi"], dims["Vc"]
    Di, Dc, Du, H = hp.item_embedding_dim, hp.cate_embedding_dim, hp.user_embedding_dim, hp.hidden_size
    D = Di + Dc
    att = list(hp.att_fcn_layer_sizes)
    specs = [
        (UNUSED_TABLE, (Vu, Du), "w"),
        (TABLES["item"], (Vi, Di), "w"),
        (TABLES["cate"], (Vc, Dc), "w"),
        (TABLES["user_long"], (Vu, Du), "w"),
        (TABLES["user_short"], (Vu, Du), "w"),
    ]
    lt = CL + "long_term/attention_fcn/"
    specs.append((lt + "attention_mat", (D, Du), "w"))
    specs += mlp_specs(lt + "att_fcn/", 4 * Du, att)
    st = CL + "short_term/"
    if hp.interest_evolve:
        specs += gru_specs(st + "short_term_intention/gru_cell/", D, Du)
    sm = hp.sequential_model
    if sm == "time4lstm":
        t = st + "time4lstm/time4lstm_cell/"      # plain RNNCell: variables live under the layer's own scope (rnn_cell_implement.py:46, TF r1.15 RNNCell.__call__ -> Layer._set_scope)
        for n_ in ("_time_input_w1", "_time_input_bias1", "_time_input_w2", "_time_input_bias2"):
            specs.append((t + n_, (H,), "glorot"))
        specs += [(t + "_time_kernel_w1", (D, H), "glorot"), (t + "_time_kernel_t1", (H, H), "glorot"),
                  (t + "_time_bias1", (H,), "glorot"),
                  (t + "_time_kernel_w2", (D, H), "glorot"), (t + "_time_kernel_t2", (H, H), "glorot"),
                  (t + "_time_bias2", (H,), "glorot"),
                  (t + "_o_kernel_t1", (H, H), "glorot"), (t + "_o_kernel_t2", (H, H), "glorot"),
                  (t + "kernel", (D + H, 4 * H), "glorot"), (t + "bias", (4 * H,), "zero")]
    elif sm == "gru":
        specs += gru_specs(st + "simple_gru/gru_cell/", D, H)
    elif sm == "lstm":
        g = st + "simple_lstm/lstm_cell/"
        specs += [(g + "kernel", (D + H, 4 * H), "glorot"), (g + "bias", (4 * H,), "zero")]
    else:
        raise ValueError("unknown sequential_model {0}".format(sm))
    sa = st + "attention_fcn/"
    Qs = Du + D
    specs.append((sa + "attention_mat", (H, Qs), "w"))
    specs += mlp_specs(sa + "att_fcn/", 4 * Qs, att)
    if not hp.manual_alpha:
        if hp.predict_long_short:
            specs += gru_specs(CL + "causal2/causal2/gru_cell/", D, H)
            a_in = H + 3 * D + 1
        else:
            a_in = 3 * D + 1
        specs += mlp_specs(CL + "fcn_alpha/", a_in, att)
    specs += mlp_specs("sequential/logit_fcn/", 2 * D, list(hp.layer_sizes))
    return specs


def init_tensor(kind, shape, hp, gen):
    """Host-side initial value (float32 cpu tensor) for one variable."""
    if kind == "w":
        method = hp.init_method
        v = float(hp.init_value)
        if method == "uniform":
            return (torch.rand(shape, generator=gen) * 2 - 1) * v
        if method == "normal":
            return torch.randn(shape, generator=gen) * v
        if method in ("xavier_normal", "xavier_uniform", "he_normal", "he_uniform"):
            fan_in = shape[0]
            fan_out = shape[1] if len(shape) > 1 else shape[0]
            if method == "xavier_uniform":
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                return (torch.rand(shape, generator=gen) * 2 - 1) * lim
            if method == "xavier_normal":
                return torch.randn(shape, generator=gen) * math.sqrt(2.0 / (fan_in + fan_out))
            if method == "he_uniform":
                lim = math.sqrt(6.0 / fan_in)
                return (torch.rand(shape, generator=gen) * 2 - 1) * lim
            return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_in)
        # tnormal (default): resample beyond two sigma
        x = torch.empty(shape)
        torch.nn.init.trunc_normal_(x, mean=0.0, std=v, a=-2 * v, b=2 * v, generator=gen)
        return x
    if kind == "glorot":
        fan_in = shape[0]
        fan_out = shape[1] if len(shape) > 1 else shape[0]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=gen) * 2 - 1) * lim
    if kind == "one":
        return torch.ones(shape)
    return torch.zeros(shape)


# ------------------------------------------------------------------------------------------ sibling models
# The three models of the reference's quick-start that are built from the same pieces as CLSR
# (examples/00_quick_start/sequential.py:94-205): they share the embedding layer and the logit MLP of
# SequentialBaseModel (sequential_base_model.py:55-74,354-452) and differ in _build_seq_graph.
SIB_TABLES = OrderedDict([("item", EMB + "item_embedding"), ("cate", EMB + "cate_embedding")])
SIBLINGS = ("gru4rec", "din", "sli_rec", "a2svd", "dien")


def sibling_kind(model_type):
    """Canonical key for ``hparams.model_type`` (the reference's yaml files spell them GRU4Rec / DIN / sli_rec)."""
    k = str(model_type).lower()
    return k if k in SIBLINGS else None


def sibling_scopes(kind):
    """Variable scopes of the model-specific part (TF names; name_scope blocks do not show up in them)."""
    if kind == "gru4rec":      # gru4rec.py:29-31,66-75: variable_scope("gru4rec"), dynamic_rnn(scope="gru")
        return dict(gru="sequential/gru4rec/gru/gru_cell/")
    if kind == "din":          # din.py:21-31 (name_scope only) + sli_rec.py:118 variable_scope("attention_fcn")
        return dict(att="sequential/attention_fcn/")
    if kind == "sli_rec":      # sli_rec.py:32-103
        s = "sequential/sli_rec/"
        return dict(asvd=s + "long_term_asvd/", t4=s + "rnn/time4lstm/time4lstm_cell/", att=s + "attention_fcn/attention_fcn/",
                    alpha=s + "fcn_alpha/")
    if kind == "dien":         # dien.py:21-57 (name_scope only): dynamic_rnn scopes "gru1" / "gru2", sli_rec.py:118
        return dict(gru1="sequential/gru1/gru_cell/", att="sequential/attention_fcn/",
                    gru2="sequential/gru2/vec_att_gru_cell/")
    if kind == "a2svd":        # asvd.py:31-38
        return dict(asvd="sequential/a2svd/Attention_layer/")
    raise ValueError("unknown sibling model %r" % (kind,))


def sibling_specs(dims, hp, kind):
    """Ordered (name, shape, init kind) of every variable of GRU4Rec / DIN / SLi-Rec."""
    Vu, Vi, Vc = dims["Vu"], dims["Vi"], dims["Vc"]
    Di, Dc, Du, H = hp.item_embedding_dim, hp.cate_embedding_dim, hp.user_embedding_dim, hp.hidden_size
    D = Di + Dc
    sc = sibling_scopes(kind)
    specs = [(UNUSED_TABLE, (Vu, Du), "w"), (SIB_TABLES["item"], (Vi, Di), "w"), (SIB_TABLES["cate"], (Vc, Dc), "w")]
    if kind == "gru4rec":
        specs += gru_specs(sc["gru"], D, H)
        out_dim = H + D
    elif kind == "a2svd":
        specs += [(sc["asvd"] + "attention_mat", (D, D), "w"), (sc["asvd"] + "query", (hp.attention_size,), "w")]
        out_dim = 2 * D
    elif kind == "dien":
        att = list(hp.att_fcn_layer_sizes)
        specs += gru_specs(sc["gru1"], D, H)
        specs.append((sc["att"] + "attention_mat", (H, D), "w"))
        specs += mlp_specs(sc["att"] + "att_fcn/", 4 * D, att)
        specs += gru_specs(sc["gru2"], H, H)
        out_dim = 3 * D + H
    elif kind == "din":
        att = list(hp.att_fcn_layer_sizes)
        specs.append((sc["att"] + "attention_mat", (D, D), "w"))
        specs += mlp_specs(sc["att"] + "att_fcn/", 4 * D, att)
        out_dim = 3 * D
    else:
        att = list(hp.att_fcn_layer_sizes)
        specs += [(sc["asvd"] + "attention_mat", (D, D), "w"), (sc["asvd"] + "query", (hp.attention_size,), "w")]
        t = sc["t4"]      # Time4LSTM over the ITEM embedding only (sli_rec.py:43-57): input width Di
        for n_ in ("_time_input_w1", "_time_input_bias1", "_time_input_w2", "_time_input_bias2"):
            specs.append((t + n_, (H,), "glorot"))
        specs += [(t + "_time_kernel_w1", (Di, H), "glorot"), (t + "_time_kernel_t1", (H, H), "glorot"),
                  (t + "_time_bias1", (H,), "glorot"),
                  (t + "_time_kernel_w2", (Di, H), "glorot"), (t + "_time_kernel_t2", (H, H), "glorot"),
                  (t + "_time_bias2", (H,), "glorot"),
                  (t + "_o_kernel_t1", (H, H), "glorot"), (t + "_o_kernel_t2", (H, H), "glorot"),
                  (t + "kernel", (Di + H, 4 * H), "glorot"), (t + "bias", (4 * H,), "zero")]
        specs.append((sc["att"] + "attention_mat", (H, D), "w"))
        specs += mlp_specs(sc["att"] + "att_fcn/", 4 * D, att)
        if not hp.manual_alpha:
            specs += mlp_specs(sc["alpha"], 3 * D + 1, att)
        out_dim = 2 * D
    specs += mlp_specs("sequential/logit_fcn/", out_dim, list(hp.layer_sizes))
    return specs

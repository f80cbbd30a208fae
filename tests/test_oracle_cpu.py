"""CPU self-tests of the oracle (oracle/clsr_oracle.py).

The reference ships no vectors for the model arithmetic (parity unpinned, see the oracle's header), so these tests
pin the restatement against the PUBLISHED definitions of the TF-1.15 ops it follows -- literal scalar loops written
independently of the oracle's tensor code, on inputs small enough to check by hand -- and against the structural
facts of the reference graph (variable set and sizes from SURVEY.md 8a, loss composition, clip / Adam formulas)."""
import math

import numpy as np
import torch

from oracle import clsr_oracle as O


def _sig(x):
    return 1.0 / (1.0 + math.exp(-x))


def test_gru_cell_follows_the_tf_formula_scalar_loop():
    """tf.nn.rnn_cell.GRUCell: [r,u] = sigmoid([x,h].Wg + bg); c = tanh([x, r*h].Wc + bc); h' = u*h + (1-u)*c."""
    g = torch.Generator().manual_seed(0)
    D, n = 3, 2
    x, h = torch.randn(1, D, generator=g, dtype=torch.float64), torch.randn(1, n, generator=g, dtype=torch.float64)
    Wg, bg = torch.randn(D + n, 2 * n, generator=g, dtype=torch.float64), torch.randn(2 * n, generator=g, dtype=torch.float64)
    Wc, bc = torch.randn(D + n, n, generator=g, dtype=torch.float64), torch.randn(n, generator=g, dtype=torch.float64)
    got = O.gru_cell(x, h, Wg, bg, Wc, bc)[0].tolist()
    xh = x[0].tolist() + h[0].tolist()
    ru = [_sig(sum(xh[i] * float(Wg[i, j]) for i in range(D + n)) + float(bg[j])) for j in range(2 * n)]
    r, u = ru[:n], ru[n:]
    xrh = x[0].tolist() + [r[k] * float(h[0, k]) for k in range(n)]
    c = [math.tanh(sum(xrh[i] * float(Wc[i, j]) for i in range(D + n)) + float(bc[j])) for j in range(n)]
    exp = [u[k] * float(h[0, k]) + (1 - u[k]) * c[k] for k in range(n)]
    assert np.allclose(got, exp, atol=1e-12)


def test_dynamic_rnn_zero_output_and_state_copy_through():
    """dynamic_rnn past sequence_length: outputs are zero, the state stops changing."""
    g = torch.Generator().manual_seed(1)
    B, T, D, n = 2, 5, 3, 2
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    P = {"s/gates/kernel": torch.randn(D + n, 2 * n, generator=g, dtype=torch.float64),
         "s/gates/bias": torch.ones(2 * n, dtype=torch.float64),
         "s/candidate/kernel": torch.randn(D + n, n, generator=g, dtype=torch.float64),
         "s/candidate/bias": torch.zeros(n, dtype=torch.float64)}
    h0 = torch.randn(B, n, generator=g, dtype=torch.float64)
    outs, hT = O.dynamic_gru(x, torch.tensor([5, 2]), h0, "s/", P)
    assert float(outs[1, 2:].abs().max()) == 0.0 and torch.equal(hT[1], outs[1, 1])
    assert torch.equal(hT[0], outs[0, 4]) and float(outs[0].abs().min()) > 0.0
    # changing inputs past the length changes nothing
    x2 = x.clone()
    x2[1, 2:] += 7.0
    outs2, hT2 = O.dynamic_gru(x2, torch.tensor([5, 2]), h0, "s/", P)
    assert torch.equal(outs2, outs) and torch.equal(hT2, hT)


def test_time4lstm_reduces_to_lstm_when_time_gates_are_open():
    """Time4LSTMCell (rnn_cell_implement.py:200-272) with both time gates saturated open and no time term in the
    output gate is tf LSTMCell with forget_bias 1: the restatements of the two cells must agree."""
    g = torch.Generator().manual_seed(2)
    B, T, D, H = 3, 4, 5, 4
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    W, b = torch.randn(D + H, 4 * H, generator=g, dtype=torch.float64) * 0.4, torch.randn(4 * H, generator=g, dtype=torch.float64) * 0.1
    z = lambda *s: torch.zeros(*s, dtype=torch.float64)
    P = {"t/kernel": W, "t/bias": b, "t/_time_input_w1": z(H), "t/_time_input_bias1": z(H), "t/_time_input_w2": z(H),
         "t/_time_input_bias2": z(H), "t/_time_kernel_w1": z(D, H), "t/_time_kernel_t1": z(H, H),
         "t/_time_bias1": torch.full((H,), 40.0, dtype=torch.float64), "t/_time_kernel_w2": z(D, H),
         "t/_time_kernel_t2": z(H, H), "t/_time_bias2": torch.full((H,), 40.0, dtype=torch.float64),
         "t/_o_kernel_t1": z(H, H), "t/_o_kernel_t2": z(H, H), "l/kernel": W, "l/bias": b}
    lens = torch.tensor([4, 1, 3])
    tt = torch.randn(B, T, generator=g, dtype=torch.float64)
    a = O.time4lstm(x, tt, tt, lens, "t/", P, H)
    c = O.dynamic_lstm(x, lens, "l/", P, H)
    assert float((a - c).abs().max()) < 1e-12


def test_batch_norm_biased_variance_all_axes_but_last_and_moving_update():
    x = torch.tensor([[[1.0, 10.0], [3.0, 10.0]], [[5.0, 10.0], [7.0, 14.0]]], dtype=torch.float64)   # [2, 2, 2]
    P = {"bn/gamma": torch.tensor([2.0, 1.0], dtype=torch.float64), "bn/beta": torch.tensor([0.5, 0.0], dtype=torch.float64)}
    st = {"bn/moving_mean": torch.zeros(2, dtype=torch.float64), "bn/moving_variance": torch.ones(2, dtype=torch.float64)}
    new = {}
    y = O.batch_norm(x, "bn/", P, st, True, new)
    mean0, var0 = 4.0, 5.0            # over the four positions (biased: mean of squared deviations)
    assert abs(float(y[0, 0, 0]) - ((1.0 - mean0) / math.sqrt(var0 + 1e-4) * 2.0 + 0.5)) < 1e-12
    assert abs(float(new["bn/moving_mean"][0]) - 0.05 * mean0) < 1e-12
    assert abs(float(new["bn/moving_variance"][0]) - (0.95 + 0.05 * var0)) < 1e-12
    y_eval = O.batch_norm(x, "bn/", P, st, False, None)    # inference: moving statistics
    assert abs(float(y_eval[1, 1, 1]) - 14.0 / math.sqrt(1.0 + 1e-4)) < 1e-12


def test_attention_mask_constant_and_softmax():
    """Padded steps get the score -(2**32)+1: their weight is exactly 0 and the rest is a softmax over valid steps."""
    class HP(object):
        att_fcn_layer_sizes = [2, 2]
        enable_BN = True
        activation = ["relu", "relu"]

    g = torch.Generator().manual_seed(3)
    B, T, Dk, Q = 2, 4, 3, 3
    names = O.mlp_names("a/att_fcn/", 4 * Q, [2, 2])
    P = {"a/attention_mat": torch.randn(Dk, Q, generator=g, dtype=torch.float64)}
    for nm, shape, kind in names:
        P[nm] = torch.ones(shape, dtype=torch.float64) if kind == "one" else torch.randn(shape, generator=g, dtype=torch.float64)
    st = O.init_bn_state(P)
    keys, q = torch.randn(B, T, Dk, generator=g, dtype=torch.float64), torch.randn(B, Q, generator=g, dtype=torch.float64)
    mask = torch.tensor([[1, 1, 0, 0], [1, 1, 1, 1]])
    out, w = O.attention_fcn(q, keys, mask, "a/", P, st, HP, True, {})
    assert float(w[0, 2:].abs().max()) == 0.0 and abs(float(w[0].sum()) - 1.0) < 1e-12
    assert abs(float(w[1].sum()) - 1.0) < 1e-12 and torch.equal(out, keys * w.unsqueeze(-1))
    assert O.MASK_PAD == -4294967295.0


def test_clip_by_norm_and_adam_first_step_formulas():
    assert O._clip_factor(0.0, 2.0) == 1.0 and O._clip_factor(1.0, 2.0) == 1.0
    assert abs(O._clip_factor(16.0, 2.0) - 0.5) < 1e-15                     # ||g|| = 4 -> scaled to norm 2
    p = {"w": torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)}
    gr = {"w": torch.tensor([0.3, -4.0, 0.0], dtype=torch.float64)}
    new_p, new_a = O.adam_apply(p, gr, O.init_adam(p), 1, 1e-3)
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    for i in range(3):
        gi = float(gr["w"][i])
        m, v = 0.1 * gi, 0.001 * gi * gi
        assert abs(float(new_p["w"][i]) - (float(p["w"][i]) - lr_t * m / (math.sqrt(v) + 1e-8))) < 1e-15
    assert torch.allclose(new_a["w"][0], 0.1 * gr["w"], rtol=0, atol=1e-15)


def test_variable_set_matches_the_reference_graph(golden_hparams):
    """Dense (non-embedding) trainables of the default graph: 124 080 floats in the scopes SURVEY.md 8a lists."""
    dims = dict(Vu=50, Vi=70, Vc=9)
    specs = O.param_specs(dims, golden_hparams)
    dense = [(n, s) for n, s, _ in specs if not n.startswith(O.EMB)]
    assert sum(int(np.prod(s)) for _, s in dense) == 124080
    names = {n for n, _, _ in specs}
    for must in ("sequential/embedding/item_embedding", "sequential/embedding/user_long_embedding",
                 "sequential/clsr/long_term/attention_fcn/attention_mat",
                 "sequential/clsr/short_term/short_term_intention/gru_cell/gates/kernel",
                 "sequential/clsr/short_term/time4lstm/time4lstm_cell/_time_kernel_t2",
                 "sequential/clsr/causal2/causal2/gru_cell/candidate/bias",
                 "sequential/clsr/fcn_alpha/nn_part/batch_normalization_1/gamma",
                 "sequential/logit_fcn/nn_part/w_nn_output"):
        assert must in names, must
    gates_bias = [k for n, s, k in specs if n.endswith("short_term_intention/gru_cell/gates/bias")]
    assert gates_bias == ["one"]                                            # GRUCell gate bias starts at 1.0


def test_losses_on_a_hand_checkable_batch(golden_hparams):
    """Group softmax (one positive first in each group), triplet terms, discrepancy and L2 on hand-made tensors."""
    hp = golden_hparams
    G = hp.train_num_ngs + 1
    logit = torch.tensor([[2.0], [0.0], [0.0], [0.0], [0.0], [1.0], [1.0], [1.0], [1.0], [1.0]], dtype=torch.float64)
    feed = {"labels": torch.tensor([[1.0]] + [[0.0]] * 4 + [[1.0]] + [[0.0]] * 4, dtype=torch.float64)}
    D = 2
    z = torch.zeros(2 * G, D, dtype=torch.float64)
    L = z.clone()
    L[:, 0] = 1.0                                                           # long = (1, 0); everything else 0
    inv = {"item": torch.tensor([[3.0, 4.0]], dtype=torch.float64), "cate": z[:1], "user_long": torch.tensor([[1.0, 1.0]], dtype=torch.float64),
           "user_short": torch.tensor([[0.0, -1.0]], dtype=torch.float64)}
    out = dict(logit=logit, involved=inv, seq_len=torch.tensor([9] * G + [1] * G), att_fea_long=L, att_fea_short=z,
               hist_mean=z, hist_recent=z)
    ls = O.losses({}, out, feed, hp)
    p1 = math.exp(2.0) / (math.exp(2.0) + 4.0)
    assert abs(float(ls["data_loss"]) - (-(math.log(p1) + math.log(0.2)) / 2)) < 1e-12
    assert abs(float(ls["regular_loss"]) - hp.embed_l2 * (25.0 / 2 + 2.0 / 2 + 1.0 / 2)) < 1e-15
    assert abs(float(ls["discrepancy_loss"]) - (-hp.discrepancy_loss_weight * (1.0 + 4.0) / 2)) < 1e-15
    # triplet, margin m: rows with len > threshold only (the first group).  dLM = dLR = (1, 0), dSM = dSR = 0:
    # relu(dLM-dLR+m) = 2m, relu(dSR-dSM+m) = 2m, relu(dLM-dSM+m) = 1+2m, relu(dSR-dLR+m) = relu(m-1) + m
    m = float(hp.triplet_margin)
    exp = (2 * m + 2 * m + (1 + 2 * m) + (max(m - 1.0, 0.0) + m)) * hp.contrastive_loss_weight
    assert abs(float(ls["contrastive_loss"]) - exp) < 1e-12
    assert abs(float(ls["loss"]) - sum(float(ls[k]) for k in ("data_loss", "regular_loss", "contrastive_loss",
                                                               "discrepancy_loss"))) < 1e-12


def test_hoisted_rnn_projections_equal_the_literal_per_step_form(golden_hparams, golden_dir):
    """bench.py's cpu_baseline leg runs the oracle with FAST_RNN (input-side products of the recurrent cells batched
    over T): same forward values and gradients as the literal per-step form, in float64 to rounding."""
    import os

    hp = golden_hparams
    g = np.load(os.path.join(golden_dir, "iterator_train_sa.npz"))
    feed = O.to_torch_feed({k[3:]: g[k] for k in g.files if k.startswith("b0_")}, dtype=torch.float64)
    dims = dict(Vu=int(feed["users"].max()) + 1, Vi=int(max(feed["items"].max(), feed["item_history"].max())) + 1,
                Vc=int(max(feed["cates"].max(), feed["item_cate_history"].max())) + 1)
    params = O.init_params(dims, hp, seed=5, dtype=torch.float64, scale_dense=8.0)
    bn = O.init_bn_state(params)
    ls0, g0, _, _, out0 = O.gradients(params, bn, feed, hp)
    O.FAST_RNN = True
    try:
        ls1, g1, _, _, out1 = O.gradients(params, bn, feed, hp)
    finally:
        O.FAST_RNN = False
    assert float((out0["logit"] - out1["logit"]).abs().max()) < 1e-12
    assert abs(float(ls0["loss"]) - float(ls1["loss"])) < 1e-12
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) < 1e-11, k

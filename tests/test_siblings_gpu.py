"""GPU parity of the sibling models (GRU4Rec, DIN, SLi-Rec: clsr_amd/seqnet.py) against their torch-CPU oracle
(oracle/sibling_oracle.py, float64, autograd): forward values, loss terms, every dense and table gradient with the
reference's clip norms, the Adam step and the BN moving statistics -- the same checks as tests/test_step_gpu.py."""
import copy
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd.params import SIB_TABLES  # noqa: E402
from clsr_amd.seqnet import SeqNet  # noqa: E402

KINDS = {"gru4rec": dict(model_type="GRU4Rec"), "din": dict(model_type="DIN"), "sli_rec": dict(model_type="sli_rec"),
         "a2svd": dict(model_type="A2SVD"), "dien": dict(model_type="DIEN")}


def _hp(golden_hparams, kind, **kw):
    hp = copy.deepcopy(golden_hparams)
    for k, v in dict(KINDS[kind], user_embedding_dim=16, attention_size=40, **kw).items():
        setattr(hp, k, v)
    return hp


def _dims(hp):
    return dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))


def _feed(golden_dir, name, b=0):
    g = np.load(os.path.join(golden_dir, name))
    pre = "b%d_" % b
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


def _close(got, exp, rtol, atol, name):
    got = torch.as_tensor(got).detach().double().cpu().reshape(-1)
    exp = torch.as_tensor(exp).detach().double().cpu().reshape(-1)
    assert got.shape == exp.shape, (name, got.shape, exp.shape)
    err = (got - exp).abs()
    excess = float((err - (atol + rtol * exp.abs())).max())
    assert excess <= 0, "%s: max abs err %.3e, max |exp| %.3e" % (name, float(err.max()), float(exp.abs().max()))


def _setup(hp, kind, dedup):
    from oracle import sibling_oracle as O

    dims = _dims(hp)
    params32 = O.init_params(dims, hp, kind, seed=5, scale_dense=8.0)
    net = SeqNet(hp, dims, kind=kind, device="cuda:0", seed=0, dedup_histories=dedup)
    sd = dict(params32)
    sd.update(O.init_bn_state(params32))
    net.load_state_dict(sd, strict=True)
    params64 = type(params32)((k, v.double()) for k, v in params32.items())
    return O, net, params64


@pytest.mark.parametrize("dedup", [True, False])
@pytest.mark.parametrize("kind,extra", [("gru4rec", {}), ("din", {}), ("sli_rec", {}), ("a2svd", {}), ("dien", {}),
                                        ("sli_rec", dict(manual_alpha=True, manual_alpha_value=0.3))])
def test_train_step_matches_oracle(golden_dir, golden_hparams, kind, extra, dedup):
    hp = _hp(golden_hparams, kind, **extra)
    O, net, params = _setup(hp, kind, dedup)
    feed = _feed(golden_dir, "iterator_train_sa.npz", b=1)
    tf = O.to_torch_feed(feed, dtype=torch.float64)
    bn, adam = O.init_bn_state(params), O.init_adam(params)
    new_p, new_bn, _, ls, grads, norms, out = O.train_step(params, bn, adam, 1, tf, hp, kind)
    net.capture_grads = True
    got = net.train_step(net.upload(feed, True))
    torch.cuda.synchronize()
    B, T, G, Hn = net.last_shape
    rep = (lambda t: t) if G == 1 else (lambda t: t[::G])
    _close(got["logit"], out["logit"], 1e-4, 1e-4, "logit")
    _close(got["model_output"], out["model_output"], 1e-4, 1e-5, "model_output")
    if kind == "sli_rec":
        _close(got["att_fea1"], rep(out["att_fea1"]), 1e-4, 1e-5, "att_fea1 (A2SVD)")
        _close(got["w_asvd"], rep(out["w_asvd"]), 1e-4, 1e-6, "A2SVD weights")
        _close(got["rnn_out"], rep(out["rnn_out"]), 1e-4, 1e-5, "rnn_outputs")
        _close(got["att_fea2"], out["att_fea2"], 1e-4, 1e-5, "att_fea2")
        if not hp.manual_alpha:
            _close(got["alpha"], out["alpha"], 1e-4, 1e-5, "alpha")
    if kind == "dien":
        _close(got["rnn_out"], rep(out["rnn_out"]), 1e-4, 1e-5, "first GRU outputs")
        _close(got["w_att"], out["w_att"], 1e-4, 1e-6, "attention weights")
        _close(got["final_state"], out["final_state"], 1e-4, 1e-5, "attentional GRU final state")
        _close(got["hist_sum"], rep(out["hist_sum"]), 1e-5, 1e-6, "hist_sum")
    if kind == "a2svd":
        _close(got["asvd_output"], rep(out["asvd_output"]), 1e-4, 1e-5, "asvd_output")
        _close(got["w_asvd"], rep(out["w_asvd"]), 1e-4, 1e-6, "A2SVD weights")
    if kind == "din":
        _close(got["hist_sum"], rep(out["hist_sum"]), 1e-5, 1e-6, "hist_sum")
        _close(got["att_fea"], out["att_fea"], 1e-4, 1e-5, "att_fea")
    if kind == "gru4rec":
        _close(got["final_state"], rep(out["final_state"]), 1e-4, 1e-5, "final_state")
    gl = net.read_losses()
    for k in ("loss", "data_loss", "regular_loss"):
        _close([gl[k]], [float(ls[k])], 1e-5, 1e-7, k)
    cap, raw = net.captured, out["raw_grads"]
    # absolute floor: exactly-zero gradients (biases feeding a batch-norm, the softmax-invariant output bias) come
    # back as fp32 accumulation noise ~1e-6 of the largest gradient
    floor = 4e-6 * max(float(raw[n].abs().max()) for n in net.dense_names)
    assert set(net.dense_names) | set(SIB_TABLES.values()) == set(raw)
    for i, name in enumerate(net.dense_names):
        scale = float(raw[name].abs().max()) + 1e-12
        _close(cap["dense"][name], raw[name], 2e-3, 2e-4 * scale + floor, "grad " + name)
        _close([float(cap["dense_sumsq"][i]) ** 0.5], [norms[name]], 1e-3, 20 * floor, "norm " + name)
    ss = cap["table_sumsq"].cpu().numpy()
    tab_norm = dict(item=(ss[0] + ss[2] + ss[4]) ** 0.5, cate=(ss[1] + ss[3] + ss[5]) ** 0.5)
    for key, name in SIB_TABLES.items():
        scale = float(raw[name].abs().max()) + 1e-12
        _close(cap["tables"][key], raw[name], 2e-3, 2e-4 * scale + floor, "grad " + name)
        if not dedup:
            _close([tab_norm[key]], [norms[name]], 1e-3, 1e-7, "clip norm " + name)
    sd = net.state_dict()
    lr = hp.learning_rate
    for name in list(net.dense_names) + list(SIB_TABLES.values()):
        g_ = grads[name].double().reshape(-1)
        sel = g_.abs() > 100 * floor
        upd_got = (sd[name].double().reshape(-1) - params[name].reshape(-1))[sel]
        upd_exp = (new_p[name].reshape(-1) - params[name].reshape(-1))[sel]
        if upd_exp.numel():
            _close(upd_got, upd_exp, 5e-3, 0.02 * lr, "adam update " + name)
    for k, v in new_bn.items():
        _close(sd[k], v, 1e-4, 1e-6, k)
    # the user table exists (checkpoint compatibility) but is never trained
    assert torch.equal(sd["sequential/embedding/user_embedding"].double(), params["sequential/embedding/user_embedding"])


@pytest.mark.parametrize("kind", ["gru4rec", "din", "sli_rec", "a2svd", "dien"])
def test_eval_scores_match_oracle(golden_dir, golden_hparams, kind):
    hp = _hp(golden_hparams, kind)
    O, net, params = _setup(hp, kind, True)
    bn = O.init_bn_state(params)
    feed_t = _feed(golden_dir, "iterator_train_sa.npz", b=0)
    new_p, new_bn, _, _, _, _, _ = O.train_step(params, bn, O.init_adam(params), 1,
                                               O.to_torch_feed(feed_t, dtype=torch.float64), hp, kind)
    net.train_step(net.upload(feed_t, True))
    feed = _feed(golden_dir, "iterator_eval_sa.npz", b=0)
    exp = O.predict(new_p, new_bn, O.to_torch_feed(feed, dtype=torch.float64), hp, kind)
    got = net.forward(net.upload(feed, False), False)
    torch.cuda.synchronize()
    _close(torch.sigmoid(got["logit"]), exp["pred"], 2e-3, 2e-4, "pred")

"""CPU tests of the sibling-model plumbing that needs no GPU: the variable inventory of clsr_amd.params (what the HIP
path allocates and checkpoints) equals the oracle's independently written one -- TF variable names, shapes and
initialiser kinds of GRU4Rec / DIN / SLi-Rec -- and the oracle's three graphs run and train on the golden feeds."""
import copy
import os

import numpy as np
import pytest
import torch

from clsr_amd.params import sibling_kind, sibling_specs

KINDS = {"gru4rec": "GRU4Rec", "din": "DIN", "sli_rec": "sli_rec", "a2svd": "A2SVD", "dien": "DIEN"}


def _hp(golden_hparams, kind, **kw):
    hp = copy.deepcopy(golden_hparams)
    for k, v in dict(model_type=KINDS[kind], user_embedding_dim=16, attention_size=40, **kw).items():
        setattr(hp, k, v)
    return hp


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_variable_inventory_matches_the_oracle(golden_hparams, kind):
    from oracle import sibling_oracle as O

    dims = dict(Vu=50, Vi=70, Vc=9)
    for extra in ({}, {"manual_alpha": True}):
        hp = _hp(golden_hparams, kind, **extra)
        assert sibling_kind(hp.model_type) == kind
        assert [tuple(x) for x in sibling_specs(dims, hp, kind)] == [tuple(x) for x in O.param_specs(dims, hp, kind)]
    names = [n for n, _, _ in sibling_specs(dims, _hp(golden_hparams, kind), kind)]
    assert len(names) == len(set(names)) and names[0] == "sequential/embedding/user_embedding"
    expect = {"gru4rec": "sequential/gru4rec/gru/gru_cell/gates/kernel",
              "din": "sequential/attention_fcn/att_fcn/nn_part/w_nn_layer0",
              "sli_rec": "sequential/sli_rec/attention_fcn/attention_fcn/attention_mat",
              "a2svd": "sequential/a2svd/Attention_layer/query",
              "dien": "sequential/gru2/vec_att_gru_cell/candidate/kernel"}[kind]
    assert expect in names and "sequential/logit_fcn/nn_part/w_nn_output" in names


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_oracle_graphs_train(golden_dir, golden_hparams, kind):
    """One oracle step per model on a golden training feed: finite losses, the loss decreases along the update,
    the unmasked A2SVD softmax sums to one over ALL steps, untouched table rows only move by Adam's zero-grad decay."""
    from oracle import sibling_oracle as O

    hp = _hp(golden_hparams, kind)
    g = np.load(os.path.join(golden_dir, "iterator_train_sa.npz"))
    feed = {k[3:]: g[k] for k in g.files if k.startswith("b0_")}
    dims = dict(Vu=int(feed["users"].max()) + 1, Vi=int(max(feed["items"].max(), feed["item_history"].max())) + 1,
                Vc=int(max(feed["cates"].max(), feed["item_cate_history"].max())) + 1)
    params = O.init_params(dims, hp, kind, seed=1, dtype=torch.float64, scale_dense=4.0)
    tf = O.to_torch_feed(feed, dtype=torch.float64)
    bn, adam = O.init_bn_state(params), O.init_adam(params)
    new_p, new_bn, _, ls, grads, norms, out = O.train_step(params, bn, adam, 1, tf, hp, kind)
    assert all(np.isfinite(float(v)) for v in ls.values()) and float(ls["loss"]) > 0
    if kind in ("sli_rec", "a2svd"):
        w = out["w_asvd"]
        assert torch.allclose(w.sum(1), torch.ones_like(w[:, 0])) and float(w[:, -1].min()) > 0   # padded steps too
    ls2 = O.gradients(new_p, new_bn, tf, hp, kind)[0]
    assert float(ls2["loss"]) < float(ls["loss"])
    assert "sequential/embedding/user_embedding" not in grads


def test_shape_limits_are_reported_by_name(golden_hparams):
    """Configurations outside the kernels' shape limits are refused when the net is built, with the limits named
    (not as an 'unsupported shape' error code from a launch in the middle of the first step)."""
    from clsr_amd.net import CLSRNet
    from clsr_amd.seqnet import SeqNet

    dims = dict(Vu=10, Vi=20, Vc=5)
    for over, needle in ((dict(layer_sizes=[50, 25]), "MLP layer widths"),
                         (dict(max_seq_length=300), "max_seq_length"),
                         (dict(item_embedding_dim=30, cate_embedding_dim=10), "multiples of 4"),
                         (dict(att_fcn_layer_sizes=[512, 40]), "att_fcn_layer_sizes")):
        hp = copy.deepcopy(golden_hparams)
        for k, v in over.items():
            setattr(hp, k, v)
        with pytest.raises(NotImplementedError, match=needle):
            CLSRNet(hp, dims)
        hp.model_type = "DIN"
        with pytest.raises(NotImplementedError, match=needle):
            SeqNet(hp, dims, kind="din")
    hp = copy.deepcopy(golden_hparams)
    hp.hidden_size, hp.model_type = 200, "GRU4Rec"
    with pytest.raises(NotImplementedError, match="hidden_size"):
        SeqNet(hp, dims, kind="gru4rec")

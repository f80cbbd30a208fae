"""Every non-default value of a mode switch that selects other KERNELS runs as a training step here and is held to the
step's bars against the oracle (logits 1e-3 absolute -- the north_star tolerance --, loss terms 1e-4 relative): code that
no test runs rots.  The table of the switches is DESIGN.md section 11."""
import copy
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clsr_amd.net import CLSRNet  # noqa: E402


def _dims(hp):
    return dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))


def _feed(golden_dir, name, b=0):
    g = np.load(os.path.join(golden_dir, name))
    pre = "b%d_" % b
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


def _set(net, name, value):
    if name == "att_bwd":
        net.set_att_bwd(value)
    else:
        assert hasattr(net, name), name
        setattr(net, name, value)


VARIANTS = [
    ("fp32", dict(att_bwd="fp32")),                      # CLSR_ATT_BWD=fp32: fp32-MFMA attention backward + dW launches
    ("fp32", dict(att_bwd="x6")),                        # CLSR_ATT_BWD=x6: three pieces in the layer-0 kernel as well
    ("fp32", dict(rnn_products="fp32", rnn_fused_proj=False, rnn_act_tiled=False)),   # CLSR_RNN_PRODUCTS=fp32
    ("fp32", dict(rnn_fused_proj=False)),                # CLSR_NO_RNN_FUSED_PROJ=1: projection GEMM in front of the recurrences
    ("fp32", dict(enc_x6=True)),                         # CLSR_ENC_BWD=x6
    ("fp32", dict(early_user_update=False)),             # CLSR_NO_EARLY_USER_UPDATE=1: all four tables updated by the sweep at the end of the step
    ("fp32", dict(fold_hist_shares=False)),              # CLSR_NO_FOLD_SHARES=1: long-term d(hist) + prologue shares added by the segmented sums
    ("fp32", dict(heads_fused=False)),                   # CLSR_NO_HEADS_FUSED=1: the launch chain of the row-level heads
    ("fp32", dict(overlap=False)),                       # CLSR_NO_OVERLAP=1: one stream
    ("fp32", dict(use_plans=False)),                     # CLSR_NO_PLAN=1
    ("fp32", dict(det_grads=False)),                     # CLSR_NO_DET_GRADS=1: counting sort + float atomics
    ("fp32", dict(defer_dw=False)),                      # CLSR_DW_EAGER=1 (bench.py)
    ("fp32x3", dict(att_bwd="fp32")),
    ("fp32x3", dict(att_bwd="x6")),
    ("fp32x3", dict(rnn_products="fp32", rnn_fused_proj=False, rnn_act_tiled=False)),
    ("fp32x3", dict(rnn_products="x6")),
    ("bf16", dict(bf16_chain=False)),                    # CLSR_BF16_CHAIN=old
]


@pytest.mark.parametrize("precision,flips", VARIANTS, ids=["%s-%s" % (p, "+".join("%s=%s" % kv for kv in f.items())) for p, f in VARIANTS])
def test_switch_variant_runs_as_a_step_and_matches_the_oracle(golden_dir, golden_hparams, precision, flips):
    from oracle import clsr_oracle as O

    hp = copy.deepcopy(golden_hparams)
    dims = _dims(hp)
    params = O.init_params(dims, hp, seed=3, scale_dense=8.0)
    feed = _feed(golden_dir, "iterator_train_sa.npz", 0)
    net = CLSRNet(hp, dims, device="cuda:0", seed=0, precision=precision)
    for k, v in flips.items():
        _set(net, k, v)
    sd = dict(params)
    sd.update(O.init_bn_state(params))
    net.load_state_dict(sd)
    _, _, _, ls, _, _, out = O.train_step(params, O.init_bn_state(params), O.init_adam(params), 1, O.to_torch_feed(feed), hp)
    f = net.upload(feed, True)
    got = None
    for _ in range(1):
        got = net.train_step(f)
    torch.cuda.synchronize()
    err = float((got["logit"].cpu() - out["logit"].reshape(-1)).abs().max())
    assert err < 1e-3, "logit mismatch vs oracle: %g" % err
    gl = net.read_losses()
    for k in ("loss", "data_loss", "contrastive_loss", "discrepancy_loss", "regular_loss"):
        tol = 1e-4 if precision != "bf16" else 2e-3
        assert abs(gl[k] - float(ls[k])) <= tol * max(abs(float(ls[k])), 1e-6) + 1e-7, (k, gl[k], float(ls[k]))
    # the updated variables are finite and moved
    new = net.state_dict()
    moved = 0
    for k, v in new.items():
        assert bool(torch.isfinite(v).all()), k
        if k in params and not torch.equal(v, params[k].float()):
            moved += 1
    assert moved > 10

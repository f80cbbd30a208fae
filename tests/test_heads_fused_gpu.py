"""The row-level heads of the training step as two persistent launches with grid barriers (csrc/headsfused.hip) against
the chain of launches they replace (alpha gate -> fcn_alpha -> fusion -> logit_fcn -> softmax loss and their gradients:
reference models/sequential/clsr.py:239-275, models/base_model.py:653-708, 215-235): same losses, logits, gradients,
batch-norm moving statistics; bit-identical between two runs; at the golden batch and at the full batch of BASELINE
configs[1] (256 workgroups of 80 rows).  The oracle parity tests of tests/test_step_gpu.py run THROUGH this path (it is
the default): here the two host paths are compared with each other."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden_net(golden_dir, hp, fused):
    import pickle

    from clsr_amd.net import CLSRNet
    from oracle import clsr_oracle as O

    dims = dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))
    gfeed = np.load(os.path.join(golden_dir, "iterator_train_sa.npz"))
    feed = {k[3:]: gfeed[k] for k in gfeed.files if k.startswith("b0_")}
    params = O.init_params(dims, hp, seed=3, scale_dense=8.0)
    sd = dict(params)
    sd.update(O.init_bn_state(params))
    net = CLSRNet(hp, dims, device="cuda:0", seed=0)
    net.load_state_dict(copy.deepcopy(sd))
    net.heads_fused = fused
    net.capture_grads = True
    return net, net.upload(feed, True)


def _compare(a, b, la, lb, ca, cb, rtol):
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-6 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    assert float((a["logit"] - b["logit"]).abs().max()) <= 1e-5 * max(1.0, float(b["logit"].abs().max()))
    assert float((a["alpha"] - b["alpha"]).abs().max()) <= 1e-6
    # floor: biases in front of a batch-norm have an analytically ZERO gradient; what is compared there is fp32 summation
    # noise (as in tests/test_fullsize_gpu.py)
    gs = max(float(g.abs().max()) for g in cb["dense"].values())
    for name, g in cb["dense"].items():
        d = float((ca["dense"][name] - g).abs().max())
        assert d <= rtol * float(g.abs().max()) + 2e-5 * gs, (name, d, float(g.abs().max()))
    for k, g in cb["tables"].items():
        d = float((ca["tables"][k] - g).abs().max())
        assert d <= rtol * float(g.abs().max()) + 2e-5 * gs, (k, d)


def test_fused_heads_match_the_launch_chain_on_the_golden_batch(golden_dir, golden_hparams):
    from clsr_amd.ops import query

    res = []
    for fused in (True, False):
        net, f = _golden_net(golden_dir, golden_hparams, fused)
        B, G = f["B"], golden_hparams.train_num_ngs + 1
        assert net._heads_fused_ok(B, G) == fused, "the golden widths are the reference's defaults: the fused path must engage"
        if fused:
            assert query("clsr_heads_fused_parts", B, G) >= 2
        out = net.train_step(f)
        torch.cuda.synchronize()
        bn = {s + k: getattr(b, k).clone() for s, b in net.bn.items() for k in ("moving_mean", "moving_var")}
        res.append((dict(logit=out["logit"].clone(), alpha=out["alpha"].clone()), net.read_losses(),
                    copy.deepcopy(net.captured), bn))
    (a, la, ca, bna), (b, lb, cb, bnb) = res
    _compare(a, b, la, lb, ca, cb, rtol=2e-4)
    for k in bnb:
        assert torch.allclose(bna[k], bnb[k], rtol=1e-5, atol=1e-7), k


def test_fused_heads_are_bit_reproducible(golden_dir, golden_hparams):
    runs = []
    for _ in range(2):
        net, f = _golden_net(golden_dir, golden_hparams, True)
        for _ in range(3):
            net.train_step(f)
        torch.cuda.synchronize()
        st = {k: v.clone() for k, v in net.state_dict().items()}
        st.update({"g." + n: g.clone() for n, g in net.captured["dense"].items()})
        st["losses"] = net.losses.clone()
        runs.append(st)
    for k in runs[0]:
        if k == "losses":       # (the data loss is summed with one double atomic per workgroup: order-dependent in its last bits)
            assert torch.allclose(runs[0][k], runs[1][k], rtol=1e-12, atol=0)
        else:
            assert torch.equal(runs[0][k], runs[1][k]), k


@pytest.mark.parametrize("P", [4096, 1001, 37])
def test_fused_heads_full_batch(P):
    """BASELINE configs[1] widths at the benchmark batch (P = 4096 positives x 5 rows = 256 workgroups x 80 rows), a batch
    whose last workgroup is short, and a small one (one group per workgroup)."""
    import sys

    sys.path.insert(0, ROOT)
    from bench import build_hparams
    from clsr_amd.net import CLSRNet
    from clsr_amd.synthetic import CONFIGS, synthetic_feed

    cfg = dict(CONFIGS["taobao"])
    T, G = cfg["T"], 5
    feed = synthetic_feed(P, T, cfg["Vu"], cfg["Vi"], cfg["Vc"], G=G, lengths="lognormal", seed=11)
    hp = build_hparams(cfg, P)
    dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
    nets = [CLSRNet(hp, dims, seed=1), CLSRNet(hp, dims, seed=1)]
    nets[1].load_state_dict(nets[0].state_dict())
    res = []
    for net, fused in zip(nets, (True, False)):
        net.heads_fused = fused
        net.capture_grads = True
        f = net.upload(feed, True)
        assert net._heads_fused_ok(f["B"], G) == fused
        out = net.train_step(f)   # (ONE step from identical states: after an Adam step the elements whose gradient is summation
        #                           noise have moved by a full learning rate in either direction, and the states differ)
        torch.cuda.synchronize()
        res.append((dict(logit=out["logit"].clone(), alpha=out["alpha"].clone()), net.read_losses(),
                    copy.deepcopy(net.captured)))
    (a, la, ca), (b, lb, cb) = res
    _compare(a, b, la, lb, ca, cb, rtol=5e-4)


def test_unsupported_widths_keep_the_launch_chain():
    from clsr_amd.ops import query

    assert query("clsr_heads_fused_supported", 5120, 5, 40, 40, 161, 80, 40, 100, 64) == 1
    assert query("clsr_heads_fused_supported", 5120, 5, 48, 40, 185, 80, 40, 100, 64) == 0      # other embedding width
    assert query("clsr_heads_fused_supported", 5120, 5, 40, 40, 161, 80, 40, 200, 80) == 0      # other layer sizes
    assert query("clsr_heads_fused_supported", 5121, 5, 40, 40, 161, 80, 40, 100, 64) == 0      # ragged groups
    assert query("clsr_heads_fused_supported", 256 * 85, 5, 40, 40, 161, 80, 40, 100, 64) == 0  # > 80 rows per workgroup
    assert query("clsr_heads_fused_supported", 256 * 80, 5, 40, 40, 161, 80, 40, 100, 64) == 1


def test_no_barrier_timeout_after_the_steps_above():
    import sys

    sys.path.insert(0, ROOT)
    from bench import build_hparams
    from clsr_amd.net import CLSRNet
    from clsr_amd.ops import query
    from clsr_amd.synthetic import CONFIGS, synthetic_feed

    cfg = dict(CONFIGS["taobao"])
    P = cfg["P"]
    feed = synthetic_feed(P, cfg["T"], cfg["Vu"], cfg["Vi"], cfg["Vc"], G=5, lengths="lognormal", seed=5)
    net = CLSRNet(build_hparams(cfg, P), dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"]), seed=2)
    f = net.upload(feed, True)
    for _ in range(5):
        net.train_step(f)
    torch.cuda.synchronize()
    assert query("clsr_heads_fused_error", net._heads_ws().data_ptr()) == 0


@pytest.mark.parametrize("seed", range(10))
def test_fused_heads_random_batches(seed):
    """Random batches at the reference's widths -- positives 1 .. 400, rows per positive 2 .. 8, history length 3 .. 50,
    bpr / triplet, ragged lengths: the two persistent launches against the launch chain (logits, alpha, losses, every
    gradient) from the same state."""
    import sys

    sys.path.insert(0, ROOT)
    from bench import build_hparams
    from clsr_amd.net import CLSRNet
    from clsr_amd.ops import query
    from clsr_amd.synthetic import synthetic_feed

    rng = np.random.default_rng(100 + seed)
    P, G, T = int(rng.integers(1, 401)), int(rng.integers(2, 9)), int(rng.choice([3, 7, 20, 50]))
    cfg = dict(Vu=500, Vi=3000, Vc=60, Di=32, Dc=8, Du=40, H=40, T=T, P=P)
    feed = synthetic_feed(P, T, cfg["Vu"], cfg["Vi"], cfg["Vc"], G=G, lengths="lognormal", seed=seed)
    hp = build_hparams(cfg, P, train_num_ngs=G - 1, contrastive_loss=str(rng.choice(["bpr", "triplet"])),
                       contrastive_length_threshold=int(rng.integers(1, 6)))
    dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
    nets = [CLSRNet(hp, dims, seed=seed), CLSRNet(hp, dims, seed=seed)]
    nets[1].load_state_dict(nets[0].state_dict())
    res = []
    for net, fused in zip(nets, (True, False)):
        net.heads_fused = fused
        net.capture_grads = True
        f = net.upload(feed, True)
        assert net._heads_fused_ok(f["B"], G) == fused, (P, G, T)
        out = net.train_step(f)
        torch.cuda.synchronize()
        res.append((dict(logit=out["logit"].clone(), alpha=out["alpha"].clone()), net.read_losses(),
                    copy.deepcopy(net.captured)))
        if fused:
            assert query("clsr_heads_fused_error", net._heads_ws().data_ptr()) == 0
    (a, la, ca), (b, lb, cb) = res
    _compare(a, b, la, lb, ca, cb, rtol=5e-4)

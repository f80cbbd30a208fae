"""The TF-1.15 pin of the oracle and of the HIP path (oracle/tf115_pin.py, scripts/capture_tf115.py).

``tests/golden/tf115_clsr_step.npz`` holds what the REFERENCE's own CLSRModel computes under TensorFlow 1.15 on the
committed batch with the deterministic weight set F2.  It can only be produced outside the build container (no
TensorFlow here): until someone runs ``python scripts/capture_tf115.py --reference <CLSR checkout>`` and commits the
file, the comparisons below SKIP with "parity unpinned"; the harness itself (same variable inventory, same weights
from (name, shape) alone, the keys the script writes) is tested on every run so that it cannot rot."""
import ast
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import clsr_oracle as O
from oracle import tf115_pin as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.environ.get("CLSR_TF115_PIN") or os.path.join(ROOT, "tests", "golden", P.PIN_FILE)
UNPINNED = ("parity unpinned: %s is absent -- run scripts/capture_tf115.py in a TensorFlow-1.15 environment and commit "
            "its output" % os.path.relpath(PIN, ROOT))


def _dims(hp):
    return dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))


def _f2_params(hp, dtype):
    specs = O.param_specs(_dims(hp), hp)
    w = P.f2_weights([(n, s) for n, s, _ in specs])
    return type(O.init_params(dict(Vu=2, Vi=2, Vc=2), hp))((n, torch.from_numpy(w[n]).to(dtype)) for n, _, _ in specs)


def _oracle_run(hp, golden_dir):
    params = _f2_params(hp, torch.float64)
    feed = O.to_torch_feed(P.feed_arrays(np.load(os.path.join(golden_dir, "iterator_train_sa.npz"))), torch.float64)
    bn, adam = O.init_bn_state(params), O.init_adam(params)
    new_p, new_bn, _, ls, grads, norms, out = O.train_step(params, bn, adam, 1, feed, hp)
    feed_e = O.to_torch_feed(P.feed_arrays(np.load(os.path.join(golden_dir, "iterator_eval_sa.npz"))), torch.float64)
    pred = O.predict(new_p, new_bn, feed_e, hp)["pred"]
    return params, new_p, new_bn, ls, norms, out, pred


def test_pin_harness_is_consistent_with_the_oracle(golden_hparams, golden_dir):
    """F2 covers exactly the oracle's variable inventory (== the reference graph's trainables), is
    deterministic from (name, shape), loads into the oracle and gives a finite, non-degenerate step; the hparams
    the capture script passes are the ones of the golden fixture; the script parses and writes the keys used below."""
    hp = golden_hparams
    for k, v in P.HPARAMS.items():
        assert getattr(hp, k) == v, k
    specs = O.param_specs(_dims(hp), hp)
    a, b = P.f2_value(specs[3][0], specs[3][1]), P.f2_value(specs[3][0], specs[3][1])
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert not np.array_equal(P.f2_value("x/kernel", (4, 4)), P.f2_value("y/kernel", (4, 4)))
    params, new_p, new_bn, ls, norms, out, pred = _oracle_run(hp, golden_dir)
    assert all(torch.isfinite(v).all() for v in new_p.values()) and torch.isfinite(out["logit"]).all()
    assert float(out["logit"].std()) > 1e-3 and 0.0 < float(out["alpha"].min()) < float(out["alpha"].max()) < 1.0
    src = open(os.path.join(ROOT, "scripts", "capture_tf115.py")).read()
    ast.parse(src)
    for key in ('"logit"', '"alpha"', '"loss/"', '"grad/"', '"slices_norm/"', '"after/"', '"eval_pred"'):
        assert key in src, key
    assert set(P.CONFIRMS) == {"logit, alpha", "loss/*", "grad/*", "slices_norm/*", "after/*", "eval_pred"}


@pytest.mark.skipif(not os.path.exists(PIN), reason=UNPINNED)
def test_oracle_matches_the_tensorflow_reference(golden_hparams, golden_dir):
    ref = np.load(PIN)
    hp = golden_hparams
    params, new_p, new_bn, ls, norms, out, pred = _oracle_run(hp, golden_dir)
    np.testing.assert_allclose(out["logit"].detach().numpy().reshape(-1), ref["logit"].reshape(-1), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["alpha"].detach().numpy().reshape(-1), ref["alpha"].reshape(-1), rtol=1e-4, atol=1e-5)
    for k in ("loss", "data_loss", "regular_loss", "contrastive_loss", "discrepancy_loss"):
        np.testing.assert_allclose(float(ls[k]), float(ref["loss/" + k]), rtol=2e-5, atol=1e-7, err_msg=k)
    gscale = max(float(np.abs(ref[k]).max()) for k in ref.files if k.startswith("grad/"))
    for name, g in out["raw_grads"].items():
        if "grad/" + name in ref.files:
            np.testing.assert_allclose(g.numpy(), ref["grad/" + name], rtol=2e-3, atol=2e-6 * gscale, err_msg=name)
        if "slices_norm/" + name in ref.files:
            np.testing.assert_allclose(norms[name], float(ref["slices_norm/" + name]), rtol=1e-4, err_msg=name)
    for name, v in list(new_p.items()) + list(new_bn.items()):
        if "after/" + name in ref.files:
            np.testing.assert_allclose(v.numpy(), ref["after/" + name], rtol=1e-4, atol=2e-5, err_msg=name)
    np.testing.assert_allclose(pred.numpy().reshape(-1), ref["eval_pred"].reshape(-1), rtol=0, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(PIN), reason=UNPINNED)
def test_hip_step_matches_the_tensorflow_reference(golden_hparams, golden_dir):
    """The HIP path against the reference's own numbers at the north_star bar (logits within 1e-3)."""
    from clsr_amd.net import CLSRNet

    ref = np.load(PIN)
    hp = golden_hparams
    params = _f2_params(hp, torch.float32)
    net = CLSRNet(hp, _dims(hp), device="cuda:0", seed=0, dedup_histories=False)   # replicated: reference-exact clip
    sd = dict(params)
    sd.update(O.init_bn_state(params))
    net.load_state_dict(sd)
    feed = P.feed_arrays(np.load(os.path.join(golden_dir, "iterator_train_sa.npz")))
    got = net.train_step(net.upload(feed, True))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got["logit"].cpu().numpy().reshape(-1), ref["logit"].reshape(-1), rtol=0, atol=1e-3)
    np.testing.assert_allclose(got["alpha"].cpu().numpy().reshape(-1), ref["alpha"].reshape(-1), rtol=0, atol=1e-3)
    gl = net.read_losses()
    for k in ("loss", "data_loss", "regular_loss", "contrastive_loss", "discrepancy_loss"):
        np.testing.assert_allclose(gl[k], float(ref["loss/" + k]), rtol=1e-4, atol=1e-6, err_msg=k)
    after = net.state_dict()
    for name, v in after.items():
        if "after/" + name in ref.files:
            np.testing.assert_allclose(v.numpy(), ref["after/" + name], rtol=1e-3, atol=5e-5, err_msg=name)
    feed_e = P.feed_arrays(np.load(os.path.join(golden_dir, "iterator_eval_sa.npz")))
    pred = torch.sigmoid(net.forward(net.upload(feed_e, False), False)["logit"]).cpu().numpy()
    np.testing.assert_allclose(pred.reshape(-1), ref["eval_pred"].reshape(-1), rtol=0, atol=1e-3)

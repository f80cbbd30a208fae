"""The TF-1.15 pin of the oracle and of the HIP path (oracle/tf115_pin.py, scripts/capture_tf115.py).

``tests/golden/tf115_clsr_step.npz`` holds what the REFERENCE's own CLSRModel computes under TensorFlow 1.15 on the
committed batch with the deterministic weight set F2.  It can only be produced outside the build container (no
TensorFlow here): until someone runs ``python scripts/capture_tf115.py --reference <CLSR checkout>`` and commits the
file, the comparisons against it SKIP with "parity unpinned".

What runs on every CPU pass regardless (so that the harness cannot fail for the wrong reason the day a real pin file
arrives): a pin file FABRICATED from the oracle under TF's variable names goes through exactly the same comparison
code; the weights are loaded from the file (``before/<name>``), the two variable inventories are compared first, and a
variable that is renamed / missing on either side fails loudly with both name lists instead of being skipped."""
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
LOSS_KEYS = ("loss", "data_loss", "regular_loss", "contrastive_loss", "discrepancy_loss")


def _dims(hp):
    return dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))


def _ordered(hp, pairs):
    return type(O.init_params(dict(Vu=2, Vi=2, Vc=2), hp))(pairs)


def _f2_params(hp, dtype):
    specs = O.param_specs(_dims(hp), hp)
    w = P.f2_weights([(n, s) for n, s, _ in specs])
    return _ordered(hp, ((n, torch.from_numpy(w[n]).to(dtype)) for n, _, _ in specs))


def _params_from_pin(ref, hp, dtype):
    """(params, bn_state) with the weights the TF run really used (``before/<name>``), after checking that TF's list
    of trainables (``meta/variables``) and the oracle's variable inventory are the same set of names."""
    specs = O.param_specs(_dims(hp), hp)
    ours = [n for n, _, _ in specs]
    P.check_inventory([str(n) for n in ref["meta/variables"]], ours, "trainable variables")
    P.require_keys(ref, "before/", ours, "weights of the pinned run")
    pairs = []
    for n, shape, _ in specs:
        v = np.asarray(ref["before/" + n])
        assert tuple(v.shape) == tuple(shape), "shape of %s: reference %s, this repo %s" % (n, v.shape, tuple(shape))
        pairs.append((n, torch.from_numpy(v.astype(np.float64)).to(dtype)))
    params = _ordered(hp, pairs)
    bn = O.init_bn_state(params)
    P.require_keys(ref, "before/", list(bn), "batch-norm moving statistics of the pinned run")
    bn = type(bn)((n, torch.from_numpy(np.asarray(ref["before/" + n]).astype(np.float64)).to(dtype)) for n in bn)
    return params, bn


def _oracle_run(params, bn, hp, golden_dir):
    dt = next(iter(params.values())).dtype
    feed = O.to_torch_feed(P.feed_arrays(np.load(os.path.join(golden_dir, "iterator_train_sa.npz"))), dt)
    adam = O.init_adam(params)
    new_p, new_bn, _, ls, grads, norms, out = O.train_step(params, bn, adam, 1, feed, hp)
    feed_e = O.to_torch_feed(P.feed_arrays(np.load(os.path.join(golden_dir, "iterator_eval_sa.npz"))), dt)
    pred = O.predict(new_p, new_bn, feed_e, hp)["pred"]
    return new_p, new_bn, ls, norms, out, pred


def fabricate_pin(path, hp, golden_dir, rename=None, drop=()):
    """Write a pin file with the keys of scripts/capture_tf115.py from the ORACLE's own numbers (float32, TF names).
    ``rename(name) -> name`` plays a reference graph whose variable names differ; ``drop``: keys left out."""
    rename = rename or (lambda n: n)
    params = _f2_params(hp, torch.float64)
    bn = O.init_bn_state(params)
    new_p, new_bn, ls, norms, out, pred = _oracle_run(params, bn, hp, golden_dir)
    f32 = lambda t: np.asarray(t.detach().numpy(), dtype=np.float32)
    unused = [n for n in params if n.endswith("/user_embedding")]
    o = {"meta/tf_version": np.array("fabricated-from-oracle"),
         "meta/variables": np.array([rename(n) for n in params]),
         "meta/no_grad": np.array([rename(n) for n in unused], dtype=str)}
    for n, v in list(params.items()) + list(bn.items()):
        o["before/" + rename(n)] = f32(v)
    o["logit"], o["alpha"] = f32(out["logit"]), f32(out["alpha"])
    for k in LOSS_KEYS:
        o["loss/" + k] = np.asarray(float(ls[k]), dtype=np.float64)
    for n, g in out["raw_grads"].items():
        o["grad/" + rename(n)] = f32(g)
        if n in set(O.TABLES.values()):
            o["slices_norm/" + rename(n)] = np.asarray(norms[n], dtype=np.float64)
    for n, v in list(new_p.items()) + list(new_bn.items()):
        o["after/" + rename(n)] = f32(v)
    o["eval_pred"] = f32(pred)
    for k in drop:
        del o[k]
    np.savez_compressed(path, **o)
    return path


def _check_after(ref, after, lr, rtol, bn_rtol, bn_atol):
    """Variables after ONE Adam step against the pin file.  Adam's first step is ~lr * g / (|g| + eps): where a gradient
    is analytically zero (biases that feed a batch-norm, the softmax-invariant output bias) every implementation returns
    its own fp32 summation noise and the step is +-lr times noise / (noise + 1e-8) -- arbitrary.  So the UPDATE
    (after - before) is compared where the reference's own gradient is well above that noise, and merely bounded by the
    step size elsewhere; batch-norm moving statistics (no optimiser) are compared directly."""
    gnames = P.ref_names(ref, "grad/")
    floor = 4e-6 * max(float(np.abs(ref["grad/" + n]).max()) for n in gnames if not n.startswith("sequential/embedding/"))
    for name, v in after.items():
        want = np.asarray(ref["after/" + name], dtype=np.float64)
        got = np.asarray(v, dtype=np.float64)
        if "grad/" + name not in ref.files:        # moving statistics, variables without a gradient
            np.testing.assert_allclose(got, want, rtol=bn_rtol, atol=bn_atol, err_msg=name)
            continue
        before = np.asarray(ref["before/" + name], dtype=np.float64)
        sel = np.abs(np.asarray(ref["grad/" + name], dtype=np.float64)) > 100 * floor
        np.testing.assert_allclose((got - before)[sel], (want - before)[sel], rtol=rtol, atol=0.02 * lr, err_msg="adam update " + name)
        assert float(np.abs(got - before).max()) <= 1.05 * lr, name     # |step 1| = lr * |g| / (|g| + eps') <= lr


def check_oracle_against(pin_path, hp, golden_dir):
    """The oracle (CPU, float64) against a pin file: forward, losses, every gradient, slice norms, one Adam step,
    moving statistics, inference.  No key is optional."""
    ref = np.load(pin_path)
    params, bn = _params_from_pin(ref, hp, torch.float64)
    new_p, new_bn, ls, norms, out, pred = _oracle_run(params, bn, hp, golden_dir)
    np.testing.assert_allclose(out["logit"].detach().numpy().reshape(-1), ref["logit"].reshape(-1), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["alpha"].detach().numpy().reshape(-1), ref["alpha"].reshape(-1), rtol=1e-4, atol=1e-5)
    for k in LOSS_KEYS:
        np.testing.assert_allclose(float(ls[k]), float(ref["loss/" + k]), rtol=2e-5, atol=1e-7, err_msg=k)
    # gradients: every trainable of the reference graph except those TF itself reports as unconnected -- and the
    # oracle must agree on which those are
    no_grad = [str(n) for n in ref["meta/no_grad"]]
    P.check_inventory(P.ref_names(ref, "grad/") + no_grad, list(params), "gradients")
    P.check_inventory(no_grad, [n for n in params if n not in out["raw_grads"]], "variables without a gradient")
    gscale = max(float(np.abs(ref[k]).max()) for k in ref.files if k.startswith("grad/"))
    tables = set(O.TABLES.values())
    P.require_keys(ref, "slices_norm/", sorted(tables), "IndexedSlices norms")
    for name, g in out["raw_grads"].items():
        np.testing.assert_allclose(g.numpy(), ref["grad/" + name], rtol=2e-3, atol=2e-6 * gscale, err_msg=name)
        if name in tables:
            np.testing.assert_allclose(norms[name], float(ref["slices_norm/" + name]), rtol=1e-4, err_msg=name)
    after = list(new_p.items()) + list(new_bn.items())
    P.require_keys(ref, "after/", [n for n, _ in after], "variables after one train step")
    _check_after(ref, {n: v.numpy() for n, v in after}, float(hp.learning_rate), rtol=5e-3, bn_rtol=1e-4, bn_atol=2e-5)
    np.testing.assert_allclose(pred.numpy().reshape(-1), ref["eval_pred"].reshape(-1), rtol=0, atol=1e-4)


def check_hip_against(pin_path, hp, golden_dir):
    """The HIP path against a pin file at the north_star bar (logits within 1e-3)."""
    from clsr_amd.net import CLSRNet

    ref = np.load(pin_path)
    params, bn = _params_from_pin(ref, hp, torch.float32)
    net = CLSRNet(hp, _dims(hp), device="cuda:0", seed=0, dedup_histories=False)   # replicated: reference-exact clip
    P.check_inventory([str(n) for n in ref["meta/variables"]], list(net.P), "trainable variables of the HIP net")
    sd = dict(params)
    sd.update(bn)
    net.load_state_dict(sd)
    feed = P.feed_arrays(np.load(os.path.join(golden_dir, "iterator_train_sa.npz")))
    got = net.train_step(net.upload(feed, True))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got["logit"].cpu().numpy().reshape(-1), ref["logit"].reshape(-1), rtol=0, atol=1e-3)
    np.testing.assert_allclose(got["alpha"].cpu().numpy().reshape(-1), ref["alpha"].reshape(-1), rtol=0, atol=1e-3)
    gl = net.read_losses()
    for k in LOSS_KEYS:
        np.testing.assert_allclose(gl[k], float(ref["loss/" + k]), rtol=1e-4, atol=1e-6, err_msg=k)
    after = {n: v for n, v in net.state_dict().items() if not n.startswith("__adam__/")}
    P.require_keys(ref, "after/", list(after), "variables after one train step (HIP net)")
    _check_after(ref, {n: v.numpy() for n, v in after.items()}, float(hp.learning_rate), rtol=5e-3, bn_rtol=1e-3,
                 bn_atol=5e-5)
    feed_e = P.feed_arrays(np.load(os.path.join(golden_dir, "iterator_eval_sa.npz")))
    pred = torch.sigmoid(net.forward(net.upload(feed_e, False), False)["logit"]).cpu().numpy()
    np.testing.assert_allclose(pred.reshape(-1), ref["eval_pred"].reshape(-1), rtol=0, atol=1e-3)


@pytest.fixture(scope="module")
def fabricated_pin(tmp_path_factory, golden_hparams, golden_dir):
    return fabricate_pin(str(tmp_path_factory.mktemp("pin") / "fabricated.npz"), golden_hparams, golden_dir)


def test_pin_harness_is_consistent_with_the_oracle(golden_hparams, golden_dir):
    """F2 covers exactly the oracle's variable inventory (== the reference graph's trainables), is
    deterministic from (name, shape), loads into the oracle and gives a finite, non-degenerate step; the hparams
    the capture script passes are the ones of the golden fixture; the script parses and writes the keys used here."""
    hp = golden_hparams
    for k, v in P.HPARAMS.items():
        assert getattr(hp, k) == v, k
    specs = O.param_specs(_dims(hp), hp)
    a, b = P.f2_value(specs[3][0], specs[3][1]), P.f2_value(specs[3][0], specs[3][1])
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert not np.array_equal(P.f2_value("x/kernel", (4, 4)), P.f2_value("y/kernel", (4, 4)))
    params = _f2_params(hp, torch.float64)
    new_p, new_bn, ls, norms, out, pred = _oracle_run(params, O.init_bn_state(params), hp, golden_dir)
    assert all(torch.isfinite(v).all() for v in new_p.values()) and torch.isfinite(out["logit"]).all()
    assert float(out["logit"].std()) > 1e-3 and 0.0 < float(out["alpha"].min()) < float(out["alpha"].max()) < 1.0
    src = open(os.path.join(ROOT, "scripts", "capture_tf115.py")).read()
    ast.parse(src)
    for key in ('"logit"', '"alpha"', '"loss/"', '"grad/"', '"slices_norm/"', '"after/"', '"eval_pred"', '"before/"',
                '"meta/variables"', '"meta/no_grad"'):
        assert key in src, key
    assert set(P.CONFIRMS) == {"logit, alpha", "loss/*", "grad/*", "slices_norm/*", "after/*", "eval_pred"}


def test_time4lstm_variables_live_under_the_cell_scope(golden_hparams):
    """Time4LSTMCell is a plain RNNCell (rnn_cell_implement.py:46): TF r1.15 creates its variables under the layer's
    own scope, ``<dynamic_rnn scope>/time4lstm_cell/`` -- like gru_cell / lstm_cell / vec_att_gru_cell.  The oracle,
    the HIP net's inventory (== checkpoint keys) and F2 (weights from crc32(name)) must all use that name."""
    from clsr_amd.params import param_specs, sibling_scopes

    hp = golden_hparams
    t = "sequential/clsr/short_term/time4lstm/time4lstm_cell/"
    want = {t + n for n in ("_time_input_w1", "_time_input_bias1", "_time_input_w2", "_time_input_bias2",
                            "_time_kernel_w1", "_time_kernel_t1", "_time_bias1", "_time_kernel_w2", "_time_kernel_t2",
                            "_time_bias2", "_o_kernel_t1", "_o_kernel_t2", "kernel", "bias")}
    for specs in (O.param_specs(_dims(hp), hp), param_specs(_dims(hp), hp)):
        got = {n for n, _, _ in specs if "/time4lstm/" in n}
        assert got == want
    assert sibling_scopes("sli_rec")["t4"] == "sequential/sli_rec/rnn/time4lstm/time4lstm_cell/"


def test_fabricated_pin_passes_the_oracle_comparison(fabricated_pin, golden_hparams, golden_dir):
    check_oracle_against(fabricated_pin, golden_hparams, golden_dir)


def test_a_renamed_variable_fails_loudly(tmp_path, golden_hparams, golden_dir):
    """A reference graph that names one scope differently must FAIL the comparison (with both name lists), not be
    skipped: here the pin file spells the Time4LSTM scope the way this repo did before round 3."""
    old = lambda n: n.replace("/time4lstm/time4lstm_cell/", "/time4lstm/")
    pin = fabricate_pin(str(tmp_path / "renamed.npz"), golden_hparams, golden_dir, rename=old)
    with pytest.raises(AssertionError) as e:
        check_oracle_against(pin, golden_hparams, golden_dir)
    msg = str(e.value)
    assert "inventories differ" in msg
    assert "short_term/time4lstm/_time_input_w1" in msg and "short_term/time4lstm/time4lstm_cell/_time_input_w1" in msg


@pytest.mark.parametrize("key", ["grad/sequential/clsr/short_term/time4lstm/time4lstm_cell/kernel",
                                 "after/sequential/logit_fcn/nn_part/batch_normalization/moving_mean",
                                 "slices_norm/sequential/embedding/item_embedding",
                                 "before/sequential/clsr/long_term/attention_fcn/attention_mat"])
def test_a_missing_key_fails_loudly(tmp_path, golden_hparams, golden_dir, key):
    pin = fabricate_pin(str(tmp_path / "dropped.npz"), golden_hparams, golden_dir, drop=(key,))
    with pytest.raises(AssertionError) as e:
        check_oracle_against(pin, golden_hparams, golden_dir)
    assert key.split("/", 1)[1] in str(e.value)


@pytest.mark.skipif(not os.path.exists(PIN), reason=UNPINNED)
def test_oracle_matches_the_tensorflow_reference(golden_hparams, golden_dir):
    check_oracle_against(PIN, golden_hparams, golden_dir)


@pytest.mark.gpu
def test_fabricated_pin_passes_the_hip_comparison(fabricated_pin, golden_hparams, golden_dir):
    """The GPU leg of the harness on the fabricated file: weights from ``before/``, inventory check against the HIP
    net's own variable names, every ``after/`` key required."""
    check_hip_against(fabricated_pin, golden_hparams, golden_dir)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(PIN), reason=UNPINNED)
def test_hip_step_matches_the_tensorflow_reference(golden_hparams, golden_dir):
    check_hip_against(PIN, golden_hparams, golden_dir)


def test_checkpoints_with_the_old_time4lstm_names_still_load():
    """Round-1/2 checkpoints spelled the scope ``.../time4lstm/<var>``: the loader maps them onto the TF names."""
    from clsr_amd.net import CLSRNet

    old = {"sequential/clsr/short_term/time4lstm/kernel": 1, "sequential/sli_rec/rnn/time4lstm/_o_kernel_t1": 2,
           "sequential/clsr/short_term/time4lstm/time4lstm_cell/bias": 3, "sequential/embedding/item_embedding": 4}
    new = CLSRNet._alias_old_names(old)
    assert new == {"sequential/clsr/short_term/time4lstm/time4lstm_cell/kernel": 1,
                   "sequential/sli_rec/rnn/time4lstm/time4lstm_cell/_o_kernel_t1": 2,
                   "sequential/clsr/short_term/time4lstm/time4lstm_cell/bias": 3,
                   "sequential/embedding/item_embedding": 4}
    assert old["sequential/clsr/short_term/time4lstm/kernel"] == 1          # the caller's dict is left alone
    both = dict(old)
    both["sequential/clsr/short_term/time4lstm/time4lstm_cell/kernel"] = 9     # the same variable under both spellings
    with pytest.raises(ValueError, match="both spellings"):
        CLSRNet._alias_old_names(both)

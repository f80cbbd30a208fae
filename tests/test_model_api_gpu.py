"""GPU end-to-end test of the reference API surface on the committed synthetic slice
(BASELINE config 1 shape): fit / run_weighted_eval / predict / save + load_model / train() return."""
import copy
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hp(golden_hparams, tmp_path, **kw):
    hp = copy.deepcopy(golden_hparams)
    hp.MODEL_DIR = str(tmp_path / "model") + "/"
    hp.save_model = True
    hp.epochs = 3
    hp.learning_rate = 0.005
    for k, v in kw.items():
        setattr(hp, k, v)
    return hp


def test_fit_eval_predict_checkpoint_roundtrip(golden_dir, golden_hparams, tmp_path):
    import random

    from clsr_amd.clsr import CLSRModel, latest_checkpoint
    from clsr_amd.sequential_iterator import SASequentialIterator

    d = os.path.join(golden_dir, "data")
    train, valid, test = (os.path.join(d, n) for n in ("train_data", "valid_data", "test_data"))
    random.seed(7)
    model = CLSRModel(_hp(golden_hparams, tmp_path), SASequentialIterator, seed=11)
    # train() keeps the reference's 8-slot return convention
    feed = next(f for f in model.iterator.load_data_from_file(train, batch_num_ngs=4) if f)
    res = model.train(model.sess, feed)
    assert len(res) == 8 and res[0] is None and res[7] is None
    assert abs(res[2] - (res[3] + res[4] + res[5] + res[6])) < 1e-9
    first_loss = res[2]
    with pytest.raises(ValueError):
        model.fit(train, valid, valid_num_ngs=0)
    assert model.fit(train, valid, valid_num_ngs=4, eval_metric="wauc") is model
    assert model.best_epoch >= 1
    feed = next(f for f in model.iterator.load_data_from_file(train, batch_num_ngs=4) if f)
    assert model.train(model.sess, feed)[2] < first_loss  # the loss went down
    res = model.run_weighted_eval(test, num_ngs=9)
    for k in ("auc", "logloss", "mean_mrr", "ndcg@2", "hit@2", "wauc"):
        assert k in res and np.isfinite(res[k]), (k, res)
    res2 = model.run_eval(test, 9)
    assert res2["auc"] == res["auc"]
    out = tmp_path / "pred.txt"
    model.predict(test, str(out))
    lines = out.read_text().strip().split("\n")
    assert len(lines) == sum(1 for _ in open(test))
    assert all(0.0 <= float(x) <= 1.0 for x in lines)
    # checkpoint round trip: a fresh model restored from the best epoch scores identically
    ckpt = latest_checkpoint(model.hparams.MODEL_DIR)
    assert ckpt and ckpt.endswith("epoch_%d" % model.best_epoch)
    fresh = CLSRModel(_hp(golden_hparams, tmp_path), SASequentialIterator, seed=99)
    fresh.load_model(ckpt)
    model.load_model(ckpt)
    a = model.run_weighted_eval(valid, num_ngs=4)
    b = fresh.run_weighted_eval(valid, num_ngs=4)
    assert a == b
    with pytest.raises(IOError):
        fresh.load_model(str(tmp_path / "does_not_exist"))


def test_unsupported_configurations_fail_loudly(golden_hparams):
    from clsr_amd.clsr import CLSRModel
    from clsr_amd.sequential_iterator import SASequentialIterator

    hp = copy.deepcopy(golden_hparams)
    hp.enable_BN = False
    with pytest.raises(NotImplementedError):
        CLSRModel(hp, SASequentialIterator)
    hp = copy.deepcopy(golden_hparams)
    hp.train_num_ngs = None
    with pytest.raises(ValueError):
        CLSRModel(hp, SASequentialIterator)

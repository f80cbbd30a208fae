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


def test_fit_learns_a_learnable_task(tmp_path):
    """End-to-end training through the HIP path on a synthetic task WITH signal (every user browses one category,
    positives come from it): a few epochs of CLSRModel.fit must lift the ranking metrics far above chance."""
    from clsr_amd.clsr import CLSRModel
    from clsr_amd.deeprec_utils import prepare_hparams
    from clsr_amd.sequential_iterator import SASequentialIterator
    from clsr_amd.synthetic import make_tsv_dataset

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = make_tsv_dataset(str(tmp_path), n_users=400, n_items=1500, n_cates=12, n_train=6000, n_valid=300,
                             n_test=300, valid_ngs=4, test_ngs=9, max_hist=20, signal=True)
    hp = prepare_hparams(os.path.join(root, "clsr_amd", "config", "clsr.yaml"), user_vocab=paths["user_vocab"],
                         item_vocab=paths["item_vocab"], cate_vocab=paths["category_vocab"], max_seq_length=20,
                         batch_size=500, train_num_ngs=4, time_unit="s", contrastive_loss="triplet",
                         contrastive_length_threshold=5, is_clip_norm=1, embed_l2=1e-6, layer_l2=1e-6,
                         discrepancy_loss_weight=0.01, contrastive_loss_weight=0.1, learning_rate=0.005,
                         show_step=10 ** 9, save_model=False, MODEL_DIR=None, epochs=6, EARLY_STOP=10)
    model = CLSRModel(hp, SASequentialIterator, seed=7)
    before = model.run_weighted_eval(paths["test_data"], num_ngs=9)
    model.fit(paths["train_data"], paths["valid_data"], valid_num_ngs=4, eval_metric="group_auc")
    after = model.run_weighted_eval(paths["test_data"], num_ngs=9)
    assert before["auc"] < 0.62, before
    assert after["auc"] > 0.85 and after["group_auc"] > 0.85 and after["mean_mrr"] > 2.0 * before["mean_mrr"], \
        (before, after)


@pytest.mark.parametrize("cls_name,yaml_name", [("GRU4RecModel", "gru4rec.yaml"), ("DINModel", "din.yaml"),
                                                ("SLI_RECModel", "sli_rec.yaml"), ("A2SVDModel", "asvd.yaml"),
                                                ("DIENModel", "dien.yaml")])
def test_sibling_models_fit_eval_predict_checkpoint(cls_name, yaml_name, tmp_path):
    """The sibling models of the reference's quick-start through the same API: fit on a task with signal lifts the
    ranking metrics, train() keeps the base-class 5-slot return, predict / checkpoint round trip work."""
    import clsr_amd.clsr as M
    from clsr_amd.clsr import latest_checkpoint
    from clsr_amd.deeprec_utils import prepare_hparams
    from clsr_amd.sequential_iterator import SequentialIterator
    from clsr_amd.synthetic import make_tsv_dataset

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = make_tsv_dataset(str(tmp_path / "data"), n_users=400, n_items=1500, n_cates=12, n_train=5000, n_valid=200,
                             n_test=200, valid_ngs=4, test_ngs=9, max_hist=20, signal=True)
    hp = prepare_hparams(os.path.join(root, "clsr_amd", "config", yaml_name), user_vocab=paths["user_vocab"],
                         item_vocab=paths["item_vocab"], cate_vocab=paths["category_vocab"], max_seq_length=20,
                         batch_size=500, train_num_ngs=4, time_unit="s", is_clip_norm=1, embed_l2=1e-6, layer_l2=1e-6,
                         learning_rate=0.005, show_step=10 ** 9, save_model=True,
                         MODEL_DIR=str(tmp_path / "model") + "/", SUMMARIES_DIR=None, write_tfevents=False, epochs=4,
                         EARLY_STOP=10)
    cls = getattr(M, cls_name)
    model = cls(hp, SequentialIterator, seed=7)
    feed = next(f for f in model.iterator.load_data_from_file(paths["train_data"], batch_num_ngs=4) if f)
    res = model.train(model.sess, feed)
    assert len(res) == 5 and res[0] is None and res[4] is None and res[2] >= res[3] > 0
    before = model.run_eval(paths["test_data"], 9)
    assert model.fit(paths["train_data"], paths["valid_data"], valid_num_ngs=4, eval_metric="group_auc") is model
    after = model.run_eval(paths["test_data"], 9)
    assert before["auc"] < 0.65 and after["auc"] > 0.8 and after["group_auc"] > 0.8, (before, after)
    out = tmp_path / "pred.txt"
    model.predict(paths["test_data"], str(out))
    assert len(out.read_text().strip().split("\n")) == sum(1 for _ in open(paths["test_data"]))
    ckpt = latest_checkpoint(hp.MODEL_DIR)
    fresh = cls(hp, SequentialIterator, seed=123)
    fresh.load_model(ckpt)
    model.load_model(ckpt)
    assert fresh.run_eval(paths["valid_data"], 4) == model.run_eval(paths["valid_data"], 4)


def test_scoring_shares_the_history_work_of_a_group(golden_dir, golden_hparams):
    """Evaluation files hold 1 + num_ngs consecutive lines per positive with one history: scoring detects the
    groups and runs the history-level part once per group -- same predictions and metrics as row by row."""
    import random

    from clsr_amd.clsr import CLSRModel
    from clsr_amd.sequential_iterator import SASequentialIterator

    d = os.path.join(golden_dir, "data")
    test = os.path.join(d, "test_data")
    hp = copy.deepcopy(golden_hparams)
    hp.batch_size = 60                      # a multiple of the 1 + 9 lines per positive
    random.seed(3)
    shared = CLSRModel(hp, SASequentialIterator, seed=5)
    rowwise = CLSRModel(hp, SASequentialIterator, seed=5, dedup_histories=False)
    rowwise.net.load_state_dict(shared.net.state_dict())
    feed = next(iter(shared.iterator.load_data_from_file(test, batch_num_ngs=0)))
    arrays = shared._to_arrays(feed)
    assert arrays.get("hist_group") == 10 and arrays["item_history"].shape[0] * 10 == arrays["items"].shape[0]
    assert "hist_group" not in rowwise._to_arrays(feed)
    ua, pa, la = shared.eval_with_user(shared.sess, feed)
    ub, pb, lb = rowwise.eval_with_user(rowwise.sess, feed)
    assert np.array_equal(ua, ub) and np.array_equal(la, lb) and ua.shape[0] == 60
    np.testing.assert_allclose(pa, pb, rtol=2e-5, atol=2e-6)
    assert shared.run_weighted_eval(test, num_ngs=9) == rowwise.run_weighted_eval(test, num_ngs=9)
    # a batch size that cuts groups apart: the detection falls back to row-by-row scoring
    hp2 = copy.deepcopy(hp)
    hp2.batch_size = 64
    cut = CLSRModel(hp2, SASequentialIterator, seed=5)
    cut.net.load_state_dict(shared.net.state_dict())
    assert cut.run_weighted_eval(test, num_ngs=9) == shared.run_weighted_eval(test, num_ngs=9)

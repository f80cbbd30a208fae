"""CPU tests: clsr_amd.sequential_reviews.data_preprocessing against fixtures produced by the reference's
data_preprocessing on the same synthetic raw logs with the same seeds (scripts/make_golden_preprocessing.py;
SURVEY.md section 8f rank 3: the on-disk format either side of the hot path)."""
import hashlib
import json
import os
import pickle
import random

import numpy as np
import pytest

from clsr_amd.sequential_reviews import data_preprocessing
from clsr_amd.synthetic import make_raw_kuaishou_csv, make_raw_taobao_csv

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "preprocessing_golden.json")))


def _run(case, d):
    cfg = GOLD["cases"][case]
    raw = os.path.join(d, "UserBehavior.csv" if cfg["dataset"] == "taobao" else "kuaishou.csv")
    (make_raw_taobao_csv if cfg["dataset"] == "taobao" else make_raw_kuaishou_csv)(raw)
    p = {k: os.path.join(d, k) for k in ("train_data", "valid_data", "test_data")}
    v = {k: os.path.join(d, k + ".pkl") for k in ("user_vocab", "item_vocab", "category_vocab")}
    random.seed(GOLD["seed"])
    np.random.seed(GOLD["seed"])
    data_preprocessing(raw, os.path.join(d, ""), p["train_data"], p["valid_data"], p["test_data"], v["user_vocab"],
                       v["item_vocab"], v["category_vocab"], sample_rate=cfg["sample_rate"],
                       valid_num_ngs=cfg["valid_num_ngs"], test_num_ngs=cfg["test_num_ngs"],
                       dataset=cfg["dataset"], is_history_expanding=cfg["expand"])
    return cfg, p, v


@pytest.mark.parametrize("case", sorted(GOLD["cases"]))
def test_preprocessing_matches_reference_outputs(case, tmp_path):
    cfg, p, v = _run(case, str(tmp_path))
    for name, path in p.items():
        want = cfg["files"][name]
        data = open(path, "rb").read()
        assert data.decode().split("\n")[:2] == want["head"], name
        assert data.count(b"\n") == want["lines"], name
        assert hashlib.sha256(data).hexdigest() == want["sha256"], name
    for name, path in v.items():
        want = cfg["files"][name]
        items = list(pickle.load(open(path, "rb")).items())
        assert [list(kv) for kv in items[:4]] == want["head"], name
        assert len(items) == want["size"], name
        assert hashlib.sha256(json.dumps(items).encode()).hexdigest() == want["sha256"], name


def test_preprocessed_files_feed_the_iterator(tmp_path):
    """The files written here are what SASequentialIterator reads: groups of 1 + ngs lines sharing a history in
    valid / test, positives only in train, every training token in the vocabularies."""
    from clsr_amd.deeprec_utils import prepare_hparams
    from clsr_amd.sequential_iterator import SASequentialIterator

    cfg, p, v = _run("taobao_full", str(tmp_path))
    hp = prepare_hparams(os.path.join(os.path.dirname(__file__), "..", "clsr_amd", "config", "clsr.yaml"),
                         user_vocab=v["user_vocab"], item_vocab=v["item_vocab"], cate_vocab=v["category_vocab"],
                         max_seq_length=20, batch_size=32, train_num_ngs=4, time_unit="s",
                         contrastive_loss="triplet", contrastive_length_threshold=5, is_clip_norm=1, embed_l2=1e-6,
                         layer_l2=1e-6, discrepancy_loss_weight=0.01, contrastive_loss_weight=0.1, show_step=10 ** 9,
                         save_model=False, MODEL_DIR=None, epochs=1)
    it = SASequentialIterator(hp, None)
    random.seed(0)
    feeds = [f for f in it.load_data_from_file(p["train_data"], batch_num_ngs=4) if f]
    assert feeds and all(f["labels"].shape[0] % 5 == 0 for f in feeds)
    assert min(int(f["items"].min()) for f in feeds) >= 1          # train tokens are all in the vocabulary
    g = 1 + cfg["valid_num_ngs"]
    for f in it.load_data_from_file(p["valid_data"], batch_num_ngs=0):
        lab = f["labels"].reshape(-1)
        if lab.size % g == 0:
            assert np.all(lab.reshape(-1, g)[:, 0] == 1) and lab.sum() == lab.size // g
    with pytest.raises(ValueError):
        data_preprocessing("x", "", "a", "b", "c", "d", "e", "f", dataset="amazon")


def test_log_that_does_not_survive_the_filters_raises_like_the_reference(tmp_path):
    """A raw log too thin for the 5 % user sample + 10-core filters: the reference dies with IndexError when it
    indexes the last touch time (sequential_reviews.py:722); same exception type here (scripts/fuzz_preprocessing.py)."""
    import pytest

    from clsr_amd.sequential_reviews import data_preprocessing
    from clsr_amd.synthetic import make_raw_taobao_csv

    raw = str(tmp_path / "UserBehavior.csv")
    make_raw_taobao_csv(raw, seed=5, n_users=300, n_items=54, n_cates=21, events_per_user=12)
    d = str(tmp_path) + "/"
    random.seed(1)
    np.random.seed(1)
    with pytest.raises(IndexError):
        data_preprocessing(raw, d, d + "tr", d + "va", d + "te", d + "u.pkl", d + "i.pkl", d + "c.pkl", sample_rate=0.3,
                           valid_num_ngs=4, test_num_ngs=1, dataset="taobao", is_history_expanding=False)

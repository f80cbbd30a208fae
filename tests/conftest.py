import os
import sys


import pytest

# (a peer-to-peer all-reduce whose peer never arrives gives up after this many seconds instead of the default minute:
#  a broken test must not sit on the GPU box; inherited by the spawned rank processes)
os.environ.setdefault("CLSR_P2P_TIMEOUT_S", "10")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def golden_hparams():
    """hparams pointing at the committed synthetic TSV slice (BASELINE config 1 shape, T=10)."""
    from clsr_amd.deeprec_utils import prepare_hparams

    d = os.path.join(GOLDEN, "data")
    return prepare_hparams(
        os.path.join(ROOT, "clsr_amd", "config", "clsr.yaml"),
        user_vocab=os.path.join(d, "user_vocab.pkl"),
        item_vocab=os.path.join(d, "item_vocab.pkl"),
        cate_vocab=os.path.join(d, "category_vocab.pkl"),
        max_seq_length=10,
        batch_size=64,
        train_num_ngs=4,
        time_unit="s",
        contrastive_loss="triplet",
        contrastive_length_threshold=5,
        is_clip_norm=1,
        embed_l2=1e-6,
        layer_l2=1e-6,
        discrepancy_loss_weight=0.01,
        contrastive_loss_weight=0.1,
        pairwise_metrics=["mean_mrr", "ndcg@2;4;6", "hit@2;4;6"],
        weighted_metrics=["wauc"],
        epochs=2,
        EARLY_STOP=5,
        show_step=1000,
        save_model=False,
        MODEL_DIR=None,
        SUMMARIES_DIR=None,
        write_tfevents=False,
    )

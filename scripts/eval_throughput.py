#!/usr/bin/env python
"""Throughput of the scoring / evaluation path: CLSRModel.run_weighted_eval and predict on a synthetic test
file with 1 + 99 lines per positive (the reference's test protocol).   python scripts/eval_throughput.py [n_pos]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from clsr_amd.clsr import CLSRModel  # noqa: E402
from clsr_amd.deeprec_utils import prepare_hparams  # noqa: E402
from clsr_amd.sequential_iterator import SASequentialIterator  # noqa: E402
from clsr_amd.synthetic import make_tsv_dataset  # noqa: E402


def main():
    n_pos = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    d = "/tmp/clsr_eval_tsv"
    paths = make_tsv_dataset(d, n_users=20000, n_items=60000, n_cates=4000, n_train=4096, n_valid=64,
                             n_test=n_pos, test_ngs=99, max_hist=70)
    hp = prepare_hparams(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clsr_amd",
                                      "config", "clsr.yaml"),
                         user_vocab=paths["user_vocab"], item_vocab=paths["item_vocab"],
                         cate_vocab=paths["category_vocab"], max_seq_length=50, batch_size=4000, train_num_ngs=4,
                         time_unit="s", contrastive_loss="triplet", contrastive_length_threshold=5, is_clip_norm=1,
                         embed_l2=1e-6, layer_l2=1e-6, discrepancy_loss_weight=0.01, contrastive_loss_weight=0.1,
                         show_step=10 ** 9, save_model=False, MODEL_DIR=None, epochs=1)
    model = CLSRModel(hp, SASequentialIterator, seed=0)
    rows = n_pos * 100
    t = time.perf_counter()
    res = model.run_weighted_eval(paths["test_data"], num_ngs=99)
    print("first pass (parses the file): %.2f s  %s" % (time.perf_counter() - t, res))
    t = time.perf_counter()
    res2 = model.run_weighted_eval(paths["test_data"], num_ngs=99)
    dt = time.perf_counter() - t
    assert res2 == res
    print("run_weighted_eval: %.3f s for %d rows = %.0f rows/s" % (dt, rows, rows / dt))
    t = time.perf_counter()
    n = sum(1 for f in model.iterator.load_data_from_file(paths["test_data"], batch_num_ngs=0) if f)
    print("  iterator alone: %.3f s (%d batches)" % (time.perf_counter() - t, n))
    t = time.perf_counter()
    model.predict(paths["test_data"], os.path.join(d, "pred.txt"))
    dt = time.perf_counter() - t
    print("predict: %.3f s = %.0f rows/s" % (dt, rows / dt))
    if os.environ.get("CLSR_PROFILE"):
        cProfile.runctx("model.run_weighted_eval(paths['test_data'], num_ngs=99)", globals(), locals(), "/tmp/ev.prof")
        pstats.Stats("/tmp/ev.prof").sort_stats("cumtime").print_stats(18)


if __name__ == "__main__":
    main()

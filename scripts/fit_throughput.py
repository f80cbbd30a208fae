#!/usr/bin/env python
"""End-to-end throughput of CLSRModel.fit's training loop (TSV -> iterator -> H2D -> hipGraph step)
on a synthetic Taobao-shaped file at batch 4096 -- the host-pipeline number that sits next to bench.py's
resident-input number."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from clsr_amd.clsr import CLSRModel  # noqa: E402
from clsr_amd.deeprec_utils import prepare_hparams  # noqa: E402
from clsr_amd.sequential_iterator import SASequentialIterator  # noqa: E402
from clsr_amd.synthetic import make_tsv_dataset  # noqa: E402


def main():
    if os.environ.get("CLSR_SWITCH"):
        sys.setswitchinterval(float(os.environ["CLSR_SWITCH"]))
    d = "/tmp/clsr_fit_tsv"
    n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    paths = make_tsv_dataset(d, n_users=20000, n_items=60000, n_cates=4000, n_train=n_train, n_valid=64, n_test=64,
                             max_hist=70)
    hp = prepare_hparams(os.path.join(os.path.dirname(__file__), "..", "clsr_amd", "config", "clsr.yaml"),
                         user_vocab=paths["user_vocab"], item_vocab=paths["item_vocab"],
                         cate_vocab=paths["category_vocab"], max_seq_length=50, batch_size=4096, train_num_ngs=4,
                         time_unit="s", contrastive_loss="triplet", contrastive_length_threshold=5, is_clip_norm=1,
                         embed_l2=1e-6, layer_l2=1e-6, discrepancy_loss_weight=0.01, contrastive_loss_weight=0.1,
                         show_step=10 ** 9, save_model=False, MODEL_DIR=None, epochs=1)
    model = CLSRModel(hp, SASequentialIterator, seed=0, use_graph=bool(os.environ.get("CLSR_GRAPH")))
    random.seed(0)
    t0 = time.perf_counter()
    it = model.iterator.load_data_from_file(paths["train_data"], batch_num_ngs=4)
    model.batch_train(it, model.sess)            # epoch 1: parse + pad + graph capture
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        it = model.iterator.load_data_from_file(paths["train_data"], batch_num_ngs=4)
        model.batch_train(it, model.sess)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # break-down: iterator alone, and upload + step on pre-built feeds
    ta = time.perf_counter()
    feeds = [f for f in model.iterator.load_data_from_file(paths["train_data"], batch_num_ngs=4) if f]
    tb = time.perf_counter()
    for f in feeds:
        model.train(model.sess, f)
    torch.cuda.synchronize()
    tc = time.perf_counter()
    if os.environ.get("CLSR_PROFILE"):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for f in feeds[:4]:
            model._train_step(model._to_arrays(f, True))
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    # host cost of enqueueing one step (no sync inside)
    td = time.perf_counter()
    for f in feeds:
        model._train_step(model._to_arrays(f, True))
    te = time.perf_counter()
    torch.cuda.synchronize()
    tf_ = time.perf_counter()
    print("enqueue-only %.1f ms/step (then %.1f ms to drain)" % (1e3 * (te - td) / len(feeds), 1e3 * (tf_ - te)))
    print("iterator alone %.1f ms/batch; upload+step on prebuilt feeds %.1f ms/step" % (
        1e3 * (tb - ta) / len(feeds), 1e3 * (tc - tb) / len(feeds)))
    steps = 3 * (n_train // 4096)
    print("first epoch %.2fs (parse, pad, capture); steady state %.2f ms/step end-to-end = %.0f interactions/s"
          % (t1 - t0, 1e3 * (t2 - t1) / steps, 4096 * steps / (t2 - t1)))


if __name__ == "__main__":
    main()

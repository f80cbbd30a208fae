#!/usr/bin/env python
"""Drop-in for the reference's examples/00_quick_start/sequential.py (CLSR branch, :120-154, :307-376).

Same flags (absl is not available here, so argparse with identical names/defaults), same call
sequence: build hparams from clsr.yaml + flag overrides -> CLSRModel(hparams, SASequentialIterator)
-> fit -> reload best checkpoint -> run_weighted_eval -> optional predict.  The only change a
user of the reference makes is the import root (clsr_amd instead of reco_utils...).

    python examples/sequential.py --dataset taobao --data_path <dir with train_data/valid_data/test_data + vocab pkl>
    python examples/sequential.py --synthetic          # 1k-item synthetic slice (BASELINE config 1 shape)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from clsr_amd.clsr import (A2SVDModel, CLSRModel, DIENModel, DINModel, GRU4RecModel, SLI_RECModel,  # noqa: E402
                           latest_checkpoint)  # noqa: E402
from clsr_amd.deeprec_utils import prepare_hparams  # noqa: E402
from clsr_amd.sequential_iterator import SASequentialIterator, SequentialIterator  # noqa: E402

YAML = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clsr_amd", "config", "clsr.yaml")


def str2bool(v):
    return str(v).lower() in ("1", "true", "yes")


def get_flags():
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", default="taobao")
    p.add_argument("--gpu_id", type=int, default=0)
    p.add_argument("--val_num_ngs", type=int, default=4)
    p.add_argument("--test_num_ngs", type=int, default=99)
    p.add_argument("--batch_size", type=int, default=500)
    p.add_argument("--save_path", default="")
    p.add_argument("--contrastive_loss", default="triplet")
    p.add_argument("--contrastive_length_threshold", type=int, default=5)
    p.add_argument("--contrastive_recent_k", type=int, default=3)
    p.add_argument("--name", default="taobao-clsr-debug")
    p.add_argument("--model", default="CLSR")
    p.add_argument("--only_test", type=str2bool, default=False)
    p.add_argument("--write_prediction_to_file", type=str2bool, default=False)
    p.add_argument("--manual_alpha", type=str2bool, default=False)
    p.add_argument("--manual_alpha_value", type=float, default=0.5)
    p.add_argument("--interest_evolve", type=str2bool, default=True)
    p.add_argument("--predict_long_short", type=str2bool, default=True)
    p.add_argument("--is_clip_norm", type=int, default=1)
    p.add_argument("--sequential_model", default="time4lstm")
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--early_stop", type=int, default=5)
    p.add_argument("--data_path", default=os.path.join("..", "..", "tests", "resources", "deeprec", "sequential"))
    p.add_argument("--train_num_ngs", type=int, default=4)
    p.add_argument("--embed_l2", type=float, default=1e-6)
    p.add_argument("--layer_l2", type=float, default=1e-6)
    p.add_argument("--triplet_margin", type=float, default=1.0)
    p.add_argument("--discrepancy_loss_weight", type=float, default=0.01)
    p.add_argument("--contrastive_loss_weight", type=float, default=0.1)
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--show_step", type=int, default=500)
    p.add_argument("--sample_rate", type=float, default=1.0)
    p.add_argument("--synthetic", action="store_true", help="generate a 1k-item synthetic slice first")
    return p.parse_args()


def get_model(flags_obj, model_path, summary_path, user_vocab, item_vocab, cate_vocab, train_num_ngs, dist=None):
    if flags_obj.dataset == "kuaishou":
        pairwise_metrics, weighted_metrics, max_seq_length, time_unit = ["mean_mrr", "ndcg@1;2"], ["wauc"], 250, "ms"
    else:
        pairwise_metrics, weighted_metrics = ["mean_mrr", "ndcg@2;4;6", "hit@2;4;6"], ["wauc"]
        max_seq_length, time_unit = (10 if flags_obj.synthetic else 50), "s"
    device = "cuda:%d" % flags_obj.gpu_id
    siblings = {"SLIREC": ("sli_rec.yaml", SLI_RECModel), "GRU4REC": ("gru4rec.yaml", GRU4RecModel),
                "DIN": ("din.yaml", DINModel), "A2SVD": ("asvd.yaml", A2SVDModel), "DIEN": ("dien.yaml", DIENModel)}
    if flags_obj.model in siblings:     # reference :94-119, :157-205: same flags, the model's own yaml
        yaml_name, cls = siblings[flags_obj.model]
        extra = dict(manual_alpha=flags_obj.manual_alpha, manual_alpha_value=flags_obj.manual_alpha_value) \
            if flags_obj.model == "SLIREC" else {}
        hparams = prepare_hparams(
            os.path.join(os.path.dirname(YAML), yaml_name), embed_l2=flags_obj.embed_l2, layer_l2=flags_obj.layer_l2,
            learning_rate=flags_obj.learning_rate, epochs=flags_obj.epochs, EARLY_STOP=flags_obj.early_stop,
            batch_size=flags_obj.batch_size, show_step=flags_obj.show_step, MODEL_DIR=model_path,
            SUMMARIES_DIR=summary_path, user_vocab=user_vocab, item_vocab=item_vocab, cate_vocab=cate_vocab,
            need_sample=True, train_num_ngs=train_num_ngs, max_seq_length=max_seq_length,
            pairwise_metrics=pairwise_metrics, weighted_metrics=weighted_metrics, time_unit=time_unit, **extra)
        if dist is not None and dist.get_rank() != 0:
            hparams.save_model = False
        return cls(hparams, SequentialIterator, seed=None, device=device, dist=dist)
    if flags_obj.model != "CLSR":
        raise SystemExit("--model must be CLSR, SLIREC, GRU4REC, DIN, DIEN or A2SVD (the other quick-start models do not share "
                         "this path's kernels; SURVEY.md section 2)")
    hparams = prepare_hparams(
        YAML, embed_l2=flags_obj.embed_l2, layer_l2=flags_obj.layer_l2,
        contrastive_loss=flags_obj.contrastive_loss, triplet_margin=flags_obj.triplet_margin,
        discrepancy_loss_weight=flags_obj.discrepancy_loss_weight,
        contrastive_loss_weight=flags_obj.contrastive_loss_weight, learning_rate=flags_obj.learning_rate,
        epochs=flags_obj.epochs, EARLY_STOP=flags_obj.early_stop, manual_alpha=flags_obj.manual_alpha,
        manual_alpha_value=flags_obj.manual_alpha_value, interest_evolve=flags_obj.interest_evolve,
        predict_long_short=flags_obj.predict_long_short, is_clip_norm=flags_obj.is_clip_norm,
        contrastive_length_threshold=flags_obj.contrastive_length_threshold,
        contrastive_recent_k=flags_obj.contrastive_recent_k, batch_size=flags_obj.batch_size,
        show_step=flags_obj.show_step, MODEL_DIR=model_path, SUMMARIES_DIR=summary_path, user_vocab=user_vocab,
        item_vocab=item_vocab, cate_vocab=cate_vocab, need_sample=True, train_num_ngs=train_num_ngs,
        max_seq_length=max_seq_length, pairwise_metrics=pairwise_metrics, weighted_metrics=weighted_metrics,
        sequential_model=flags_obj.sequential_model, time_unit=time_unit)
    # data parallel: save_model stays on for EVERY rank -- CLSRModel.save_model is collective (rank 0 writes the
    # file atomically, then all ranks pass a barrier), so nobody loads a checkpoint that is still being written
    return CLSRModel(hparams, SASequentialIterator, seed=None, device=device, dist=dist)


def init_data_parallel(flags_obj):
    """One process per GPU (``python -m torch.distributed.run --nproc-per-node N examples/sequential.py ...``):
    RCCL process group, this rank's GPU, and the SAME ``random`` stream on every rank (all ranks iterate the same
    global batches of ``--batch_size`` positives and train on their share, see CLSRModel(dist=...))."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return None
    import random

    import torch
    import torch.distributed as dist

    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    flags_obj.gpu_id = local
    random.seed(20220425)
    return dist


def main():
    flags_obj = get_flags()
    dist = init_data_parallel(flags_obj)
    print("System version: {}".format(sys.version))
    print("start experiment")
    data_path = os.path.join(flags_obj.data_path, flags_obj.dataset)
    if flags_obj.synthetic:
        from clsr_amd.synthetic import make_tsv_dataset

        data_path = os.path.join(flags_obj.save_path or "/tmp", "clsr_synthetic")
        if dist is None or dist.get_rank() == 0:     # one writer; the others wait for the finished files
            make_tsv_dataset(data_path, n_train=2000, n_valid=100, n_test=100, valid_ngs=flags_obj.val_num_ngs,
                             test_ngs=flags_obj.test_num_ngs)
        if dist is not None:
            dist.barrier()
    train_file, valid_file, test_file = (os.path.join(data_path, n) for n in ("train_data", "valid_data", "test_data"))
    user_vocab, item_vocab, cate_vocab = (os.path.join(data_path, n) for n in
                                          ("user_vocab.pkl", "item_vocab.pkl", "category_vocab.pkl"))
    output_file = os.path.join(data_path, "output.txt")
    if not os.path.exists(train_file):
        # like the reference quick-start (:318-343): build the files from the raw behaviour log
        from clsr_amd.sequential_reviews import data_preprocessing

        reviews_file = os.path.join(data_path, {"taobao": "UserBehavior.csv", "kuaishou": "kuaishou.csv"}.get(
            flags_obj.dataset, flags_obj.dataset + ".csv"))
        if not os.path.exists(reviews_file):
            raise SystemExit("neither %s nor the raw log %s exists (or pass --synthetic)" % (train_file, reviews_file))
        if dist is None or dist.get_rank() == 0:
            data_preprocessing(reviews_file, os.path.join(data_path, ""), train_file, valid_file, test_file,
                               user_vocab, item_vocab, cate_vocab, sample_rate=flags_obj.sample_rate,
                               valid_num_ngs=flags_obj.val_num_ngs, test_num_ngs=flags_obj.test_num_ngs,
                               dataset=flags_obj.dataset)
        if dist is not None:
            dist.barrier()
    save_path = os.path.join(flags_obj.save_path, flags_obj.model, flags_obj.name)
    model_path, summary_path = os.path.join(save_path, "model/"), os.path.join(save_path, "summary/")
    model = get_model(flags_obj, model_path, summary_path, user_vocab, item_vocab, cate_vocab, flags_obj.train_num_ngs,
                      dist=dist)
    if flags_obj.only_test:
        model.load_model(latest_checkpoint(model_path))
        print(model.run_weighted_eval(test_file, num_ngs=flags_obj.test_num_ngs))
        return
    start_time = time.time()
    model = model.fit(train_file, valid_file, valid_num_ngs=flags_obj.val_num_ngs, eval_metric="wauc")
    print("Time cost for training is {0:.2f} mins".format((time.time() - start_time) / 60.0))
    if dist is not None:
        dist.barrier()                  # every rank has left fit(): the last best-epoch checkpoint is complete
    model.load_model(latest_checkpoint(model_path))
    res = model.run_weighted_eval(test_file, num_ngs=flags_obj.test_num_ngs)
    print(flags_obj.name)
    print(res)
    if flags_obj.write_prediction_to_file:
        model.predict(test_file, output_file)


if __name__ == "__main__":
    main()

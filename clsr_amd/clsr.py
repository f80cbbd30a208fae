"""CLSRModel / SequentialBaseModel -- the reference's user-facing model API on top of the HIP step.

Mirrors (same method names, arguments, return conventions, error behaviour):
  reco_utils/recommender/deeprec/models/base_model.py            BaseModel      (:18-71, :381-410)
  reco_utils/recommender/deeprec/models/sequential/sequential_base_model.py
                                                                  SequentialBaseModel (:19-48, :111-352)
  reco_utils/recommender/deeprec/models/sequential/clsr.py       CLSRModel      (:383-446)
so that ``examples/00_quick_start/sequential.py``'s call sequence
(``CLSRModel(hparams, SASequentialIterator, seed) -> fit -> load_model -> run_weighted_eval -> predict``)
runs unchanged against this package (see examples/sequential.py).

There is no TF session: ``self.sess`` is an opaque handle, ``train/eval_with_user/infer`` accept and
ignore it.  One training step = one hipGraph replay of the kernels in libclsr_hip.so (captured per
batch shape); feeds are copied into static device buffers first.
"""
import os
import time

import numpy as np
import torch

from clsr_amd import ops
from clsr_amd.deeprec_utils import cal_mean_alpha_metric, cal_metric, cal_weighted_metric, load_dict
from clsr_amd.net import CLSRNet

__all__ = ["BaseModel", "SequentialBaseModel", "CLSRModel", "latest_checkpoint"]


def _prefetch(gen, depth=3):
    """Run a feed generator in a background thread, ``depth`` batches ahead of the consumer
    (the numpy collate releases the GIL in its large copies, so it overlaps with the GPU step)."""
    import queue
    import threading

    q = queue.Queue(maxsize=depth)
    end = object()
    stop = threading.Event()

    def put(item):
        """Blocking put that gives up when the consumer has gone away (returns False then)."""
        while not stop.is_set():
            try:
                q.put(item, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def work():
        try:
            for item in gen:
                if not put(item):
                    return
            put(end)
        except BaseException as e:  # surface iterator errors in the consumer
            put(e)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is end:
                break
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        # the consumer is done or raised mid-epoch (kernel error, unshardable batch): stop the worker instead of
        # leaving it blocked on a full queue with batches alive and the global ``random`` stream advancing.  NOTE:
        # the iterator shuffles / samples negatives from the process-global ``random`` module in this thread -- do
        # not touch ``random`` from the main thread while an epoch is running, or the documented draw sequence of
        # the reference is lost.
        stop.set()
        while True:
            try:
                q.get_nowait()
            except queue.Empty:
                break
        th.join(timeout=5.0)

_CKPT_INDEX = "checkpoint"


def latest_checkpoint(model_dir):
    """Path of the most recent checkpoint written under ``model_dir`` (``tf.train.latest_checkpoint``
    stand-in used by the quick-start script, reference sequential.py:352,369); ``None`` if there is none."""
    idx = os.path.join(model_dir, _CKPT_INDEX)
    if not os.path.exists(idx):
        return None
    with open(idx) as f:
        name = f.read().strip()
    return name or None


class _Session(object):
    """Opaque stand-in for tf.Session (the reference passes ``self.sess`` around)."""

    def close(self):
        pass


class BaseModel(object):
    def __init__(self, hparams, iterator_creator, graph=None, seed=None):
        """Build iterator, parameters, optimiser state (reference BaseModel.__init__ :18-71)."""
        self.seed = seed
        if seed is not None:
            np.random.seed(seed)
        self.graph = graph
        self.iterator = iterator_creator(hparams, self.graph)
        self.train_num_ngs = hparams.train_num_ngs if "train_num_ngs" in hparams else None
        self.hparams = hparams
        self.sess = _Session()
        self._build_graph()

    def _build_graph(self):
        raise NotImplementedError

    def load_model(self, model_path=None):
        """Restore variables (+ BN moving stats, Adam slots); any failure raises IOError like the
        reference (base_model.py:394-410)."""
        act_path = self.hparams.load_saved_model
        if model_path is not None:
            act_path = model_path
        try:
            from safetensors.torch import load_file

            path = act_path if str(act_path).endswith(".safetensors") else str(act_path) + ".safetensors"
            sd = load_file(path)
            self.net.load_state_dict(sd, strict=True)
        except Exception:
            raise IOError("Failed to find any matching files for {0}".format(act_path))

    def save_model(self, save_path):
        """Write all variables under their TF names (safetensors) and update the directory index."""
        from safetensors.torch import save_file

        dist = getattr(self, "_dist", None)
        group = getattr(self, "_group", None)
        rank = 0
        if dist is not None:
            # rank WITHIN the model's own process group (a model may train on a sub-group, or several replica groups
            # may exist side by side): exactly one writer per group, and only that group meets at the barrier
            rank = self._dp.rank if getattr(self, "_dp", None) is not None else dist.get_rank(group)
        if rank == 0:   # replicas are identical: ONE writer; temp file + rename, index last (no torn checkpoints)
            path = str(save_path) + ".safetensors"
            d = os.path.dirname(path)
            if d:
                os.makedirs(d, exist_ok=True)
            # unique per writer: two groups saving to the same path never share a temporary file
            uniq = "%d.%d" % (os.getpid(), dist.get_rank() if dist is not None else 0)
            tmp = path + ".tmp." + uniq
            save_file({k: v.contiguous() for k, v in self.net.state_dict().items()}, tmp)
            os.replace(tmp, path)
            idx = os.path.join(d, _CKPT_INDEX)
            with open(idx + ".tmp." + uniq, "w") as f:
                f.write(str(save_path))
            os.replace(idx + ".tmp." + uniq, idx)
        if dist is not None:
            dist.barrier(group=group)      # nobody reads the checkpoint before the group's writer has finished
        return str(save_path)


class SequentialBaseModel(BaseModel):
    def __init__(self, hparams, iterator_creator, graph=None, seed=None, device="cuda:0", use_graph=False,
                 dedup_histories=True, dist=None, sync_bn=True, group=None, precision="fp32", table_dtype="fp32"):
        """Reference ``SequentialBaseModel.__init__`` (:19-48): requires ``train_num_ngs``.

        ``dist`` (an initialised ``torch.distributed`` module, one process per GPU) turns ``train`` / ``fit``
        into single-node data parallelism: every rank iterates the SAME global batches (same files, same
        ``random`` seed -- the in-batch negative sampling runs over the global batch like in the reference),
        trains on its contiguous share of the positives and exchanges gradients through
        :class:`clsr_amd.dp.DataParallel`; ``hparams.batch_size`` is the GLOBAL batch.  Evaluation is
        replicated (every rank scores the whole file and gets the same metrics)."""
        self.hparams = hparams
        self.need_sample = hparams.need_sample
        self.train_num_ngs = hparams.train_num_ngs
        if self.train_num_ngs is None:
            raise ValueError("Please confirm the number of negative samples for each positive instance.")
        self.min_seq_length = hparams.min_seq_length if "min_seq_length" in hparams else 1
        self.hidden_size = hparams.hidden_size if "hidden_size" in hparams else None
        self._device = device
        self._dist, self._sync_bn, self._dp, self._group = dist, sync_bn, None, group
        self._precision = precision
        self._table_dtype = table_dtype      # "bf16": embedding tables stored as bf16 (CLSRNet(table_dtype=...))
        self.dp_dropped_positives = self.dp_skipped_batches = 0
        self._use_graph = use_graph and dist is None
        self._dedup = dedup_histories
        self._graphs = {}
        self._static = {}
        self._stream = None
        self._overlap_upload = not os.environ.get("CLSR_NO_OVERLAP_UPLOAD")
        self.best_epoch = 0
        super(SequentialBaseModel, self).__init__(hparams, iterator_creator, graph=graph, seed=seed)

    # ------------------------------------------------------------------ construction
    def _build_graph(self):
        hp = self.hparams
        self.keep_prob_train = 1 - np.array(hp.dropout)
        self.keep_prob_test = np.ones_like(hp.dropout)
        self.embedding_keep_prob_train = 1.0 - hp.embedding_dropout
        self.embedding_keep_prob_test = 1.0
        dims = dict(Vu=len(load_dict(hp.user_vocab)), Vi=len(load_dict(hp.item_vocab)),
                    Vc=len(load_dict(hp.cate_vocab)))
        self.user_vocab_length, self.item_vocab_length, self.cate_vocab_length = dims["Vu"], dims["Vi"], dims["Vc"]
        self.net = self._make_net(hp, dims)

    def _make_net(self, hp, dims):
        return CLSRNet(hp, dims, device=self._device, seed=self.seed, dedup_histories=self._dedup,
                       precision=self._precision, table_dtype=self._table_dtype)

    # ------------------------------------------------------------------ device feeds / graphs
    def _to_arrays(self, feed_dict, training=False):
        """Accept the iterator's feed (keys are the iterator attributes == field names).  A training
        feed that carries the compact history-level arrays (sequential_iterator.LazyFeed) is passed on
        in that form, so the (1 + train_num_ngs)-fold repeated histories are never built or uploaded."""
        it = self.iterator
        compact = getattr(feed_dict, "compact", None) if training else None
        if compact is not None:
            arrays = {name: feed_dict[getattr(it, name)] for name in ("labels", "items", "cates")}
            arrays.update(compact)
            return arrays
        arrays = {name: feed_dict[getattr(it, name)] for name in
                  ("labels", "users", "items", "cates", "item_history", "item_cate_history", "mask",
                   "time_from_first_action", "time_to_now")}
        if not training and self._dedup:
            g = self._shared_history_group(arrays)
            if g > 1:
                arrays["users_rows"] = np.asarray(arrays["users"])
                # evaluation / test files hold 1 + num_ngs consecutive lines per positive with ONE history: gathers,
                # recurrences and the long-term attention run once per group, like in training
                for k in ("users", "item_history", "item_cate_history", "mask", "time_from_first_action",
                          "time_to_now"):
                    arrays[k] = np.ascontiguousarray(np.asarray(arrays[k])[::g])
                arrays["hist_group"] = g
        return arrays

    @staticmethod
    def _shared_history_group(arrays):
        """Largest G such that the batch is made of runs of exactly G consecutive rows with identical user, history,
        mask and time features (1: no sharing).  One vectorised comparison of neighbouring rows."""
        ih = np.asarray(arrays["item_history"])
        n = ih.shape[0]
        if n < 2:
            return 1
        same = np.ones(n - 1, dtype=bool)
        for k in ("item_history", "item_cate_history", "mask", "time_from_first_action", "time_to_now"):
            a = np.asarray(arrays[k])
            same &= (a[1:] == a[:-1]).all(axis=1)
        u = np.asarray(arrays["users"]).reshape(-1)
        same &= u[1:] == u[:-1]
        starts = np.flatnonzero(np.concatenate(([True], ~same)))
        runs = np.diff(np.concatenate((starts, [n])))
        g = int(runs.min())
        return g if g > 1 and n % g == 0 and bool(np.all(runs % g == 0)) else 1

    def _static_feed(self, feed, training, lookahead=False):
        """Copy a numpy feed into static device buffers keyed by (rows, T, mode, layout) through
        persistent pinned staging buffers (no per-step allocation).

        ``lookahead`` (training loop): two device arenas per shape are filled in turn, and the copy is issued on the
        stream of the weight-gradient kernels, which is idle during the forward pass -- staged BEFORE the previous
        step is launched, batch N+1 crosses PCIe underneath the forward pass of step N (a stream of its own would be
        the fifth concurrently active one: 7 ms/step)."""
        mask = feed["mask"]
        key = (int(mask.shape[0]), int(mask.shape[1]), bool(training), int(feed.get("hist_group", 0) or 0))
        if not lookahead:
            st = self._static.get(key)
            if st is None:
                st = self._static[key] = self.net.upload(feed, training)
                return key, st, True
            return key, self.net.upload(feed, training, into=st), False
        side = self.net._side_stream("@dw0")
        st = self._static.get(("2x",) + key)
        if st is None:
            st = self._static[("2x",) + key] = [[None, None], 0]
        i = st[1]
        st[1] ^= 1
        fresh = st[0][i] is None
        st[0][i] = self.net.upload(feed, training, into=st[0][i], copy_stream=side)
        return key, st[0][i], fresh

    @staticmethod
    def _mark_free(f):
        ev = torch.cuda.Event()
        ev.record()
        f["_free"] = ev

    def _stream_ctx(self):
        """All device work of the model runs on one private HIP stream: launches on the legacy
        default stream pay an implicit-synchronisation tax per kernel (measured 200 us vs 10 us)."""
        if self._stream is None:
            # the compute stream outranks the side streams of the step (their MFMA-saturated weight-gradient kernels
            # starve whatever shares a CU with them); see bench.py for the stream-count condition under data parallelism
            multi_rank = self._dist is not None and self._dist.get_world_size(self._group) > 1
            prio = int(os.environ.get("CLSR_MAIN_PRIORITY", "0" if multi_rank else "-1"))   # (more than one rank: bench.py)
            self._stream = torch.cuda.Stream(device=self.net.device, priority=prio)
        return torch.cuda.stream(self._stream)

    def _stage(self, feed, lookahead=False):
        """Host arrays of one training batch -> an uploaded (or uploading) device feed."""
        with self._stream_ctx():
            if self._dist is not None:
                from clsr_amd.dp import DataParallel, shard_feed

                if self._dp is None:
                    self._dp = DataParallel(self.net, self._dist, sync_bn=self._sync_bn, group=self._group)
                feed = shard_feed(feed, self._dp.rank, self._dp.world, self.train_num_ngs + 1)
            key, f, _ = self._static_feed(feed, True, lookahead=lookahead and not self._use_graph)
            return key, f

    def _run_staged(self, key, f):
        net = self.net
        with self._stream_ctx():
            ready = f.pop("_ready", None)
            if ready is not None:
                torch.cuda.current_stream().wait_event(ready)
            if self._dist is not None:
                self._dp.train_step(self._dp.prepare(f))
                self._mark_free(f)
                return
            if not self._use_graph:
                net.train_step(f)       # <1 ms of host time (replayed launch plan), hidden behind the GPU
                self._mark_free(f)
                return
            # optional hipGraph replay (one graph per batch shape).  hipGraphLaunch costs ~6.7 ms of
            # host time for this graph on ROCm 7.2, so it only pays when steps are enqueued back to back.
            g = self._graphs.get(key)
            if g is None:
                net.train_step(f)               # eager: THIS is the step for this batch (also allocates
                self._stream.synchronize()      # every workspace buffer the capture below will reference)
                ops.graph_begin()
                net.train_step(f)               # recorded only, nothing executes during capture
                self._graphs[key] = ops.graph_end()
                return
            ops.graph_launch(g)

    def _train_step(self, feed):
        self._run_staged(*self._stage(feed))

    # ------------------------------------------------------------------ per-step API
    def train(self, sess, feed_dict):
        """One optimisation step (reference CLSRModel.train, clsr.py:383-408).  Returns the same 8-list:
        [update, extra_update_ops, loss, data_loss, regular_loss, contrastive_loss, discrepancy_loss, summary]."""
        self._train_step(self._to_arrays(feed_dict, True))
        with self._stream_ctx():
            ls = self.net.read_losses()
        return [None, [], ls["loss"], ls["data_loss"], ls["regular_loss"], ls["contrastive_loss"],
                ls["discrepancy_loss"], None]

    def _score(self, feed_dict):
        feed = self._to_arrays(feed_dict)
        with self._stream_ctx():
            key, f, _ = self._static_feed(feed, False)
            out = self.net.forward(f, False)
            pred = torch.sigmoid(out["logit"]) if self.hparams.method == "classification" else out["logit"]
            pred = pred.detach().cpu().numpy().reshape(-1, 1)
            if "alpha" in out:
                out = dict(out, alpha=out["alpha"].detach().cpu())
        return feed, pred, out

    def eval(self, sess, feed_dict):
        feed, pred, _ = self._score(feed_dict)
        return pred, np.asarray(feed["labels"]).reshape(-1, 1)

    def eval_with_user(self, sess, feed_dict):
        """(users, pred [B,1], labels [B,1]) -- reference sequential_base_model.py:294-308."""
        feed, pred, _ = self._score(feed_dict)
        return (np.asarray(feed.get("users_rows", feed["users"])).astype(np.int32), pred,
                np.asarray(feed["labels"]).reshape(-1, 1))

    def eval_with_user_and_alpha(self, sess, feed_dict):
        feed, pred, out = self._score(feed_dict)
        alpha = out["alpha"].numpy().reshape(-1, 1)
        return (np.asarray(feed.get("users_rows", feed["users"])).astype(np.int32), pred,
                np.asarray(feed["labels"]).reshape(-1, 1), alpha)

    def infer(self, sess, feed_dict):
        """[pred] -- reference base_model.py:381-392."""
        return [self._score(feed_dict)[1]]

    # ------------------------------------------------------------------ loops
    def batch_train(self, file_iterator, train_sess):
        """One epoch of mini-batches (reference clsr.py:410-446); returns the summed loss."""
        step = 0
        net = self.net
        with self._stream_ctx():
            acc = torch.zeros(8, dtype=torch.float64, device=net.device)  # running loss sums stay on the device
        def run(staged):
            nonlocal step
            self._run_staged(*staged)                                         # no host sync per step
            with self._stream_ctx():
                ops.call("clsr_add_doubles", acc, net.losses, 8)
                step += 1
                if step % self.hparams.show_step == 0:
                    ls = net.read_losses()                              # synchronises (only when printing)
                    print("step {0:d} , total_loss: {1:.4f}, data_loss: {2:.4f}".format(step, ls["loss"],
                                                                                       ls["data_loss"]))

        # one batch of lookahead: batch N+1 is staged (its PCIe copy enqueued) BEFORE step N is launched
        pending = None
        world = self._dp_world()
        for batch_data_input in _prefetch(file_iterator):
            if batch_data_input:
                arrays = self._to_arrays(batch_data_input, True)
                if self._dist is not None:
                    # a last batch with fewer positives than ranks cannot be sharded: every rank sees the same
                    # global batch and skips it alike; a remainder (positives % world) is truncated by shard_feed
                    n_pos = arrays["labels"].shape[0] // (self.train_num_ngs + 1)
                    self.dp_dropped_positives += n_pos % world if n_pos >= world else n_pos
                    if n_pos < world:
                        self.dp_skipped_batches += 1
                        continue
                staged = self._stage(arrays, lookahead=self._overlap_upload)
                if pending is not None:
                    run(pending)
                pending = staged
        if pending is not None:
            run(pending)
        if self._dist is not None and self.dp_dropped_positives and self._dp is not None and self._dp.rank == 0:
            print("data parallel: %d positives dropped so far (%d batches skipped): global batches are truncated to "
                  "a multiple of %d ranks" % (self.dp_dropped_positives, self.dp_skipped_batches, world))
        with self._stream_ctx():
            total = float(acc[:4].sum().item())
            net.check_abort()       # end of an epoch: an aborted step (bounded wait gave up) must not reach evaluation / checkpoints
            return total

    def _dp_world(self):
        """Data-parallel world size of THIS model's process group (1 without ``dist``)."""
        if self._dist is None:
            return 1
        if self._dp is not None:
            return self._dp.world
        return self._dist.get_world_size(self._group)

    def fit(self, train_file, valid_file, valid_num_ngs, eval_metric="group_auc"):
        """Train with per-epoch validation, early stopping and best-epoch checkpoints
        (reference sequential_base_model.py:111-202)."""
        if not self.need_sample and self.train_num_ngs < 1:
            raise ValueError(
                "Please specify a positive integer of negative numbers for training without sampling needed.")
        if valid_num_ngs < 1:
            raise ValueError("Please specify a positive integer of negative numbers for validation.")
        if self.need_sample and self.train_num_ngs < 1:
            self.train_num_ngs = 1
        train_sess = self.sess
        eval_info = list()
        best_metric, self.best_epoch = 0, 0
        for epoch in range(1, self.hparams.epochs + 1):
            self.hparams.current_epoch = epoch
            file_iterator = self.iterator.load_data_from_file(
                train_file, min_seq_length=self.min_seq_length, batch_num_ngs=self.train_num_ngs)
            self.batch_train(file_iterator, train_sess)
            valid_res = self.run_weighted_eval(valid_file, valid_num_ngs)
            print("eval valid at epoch {0}: {1}".format(
                epoch, ",".join(["" + str(key) + ":" + str(value) for key, value in valid_res.items()])))
            eval_info.append((epoch, valid_res))
            progress = False
            early_stop = self.hparams.EARLY_STOP
            if valid_res[eval_metric] > best_metric:
                best_metric = valid_res[eval_metric]
                self.best_epoch = epoch
                progress = True
            else:
                if early_stop > 0 and epoch - self.best_epoch >= early_stop:
                    print("early stop at epoch {0}!".format(epoch))
                    break
            if self.hparams.save_model and self.hparams.MODEL_DIR:
                if not os.path.exists(self.hparams.MODEL_DIR):
                    os.makedirs(self.hparams.MODEL_DIR)
                if progress:
                    self.save_model(self.hparams.MODEL_DIR + "epoch_" + str(epoch))
        print(eval_info)
        print("best epoch: {0}".format(self.best_epoch))
        return self

    # ------------------------------------------------------------------ evaluation on the device
    def _device_eval(self, filename, num_ngs, weighted):
        """Score ``filename`` and compute the metrics on the device (clsr_amd/device_metrics.py): no per-batch host
        synchronisation, scores never leave HBM.  None when a requested metric has no device form (host path then)."""
        from clsr_amd import device_metrics as DM

        if os.environ.get("CLSR_HOST_METRICS") or not DM.supported(self.hparams, self.user_vocab_length):
            return None
        with self._stream_ctx():
            acc = DM.DeviceScores(self.net.device)

            def score(feeds):
                feed = feeds[0] if len(feeds) == 1 else {
                    k: (v if k == "hist_group" else np.concatenate([np.asarray(x[k]) for x in feeds], axis=0))
                    for k, v in feeds[0].items()}
                key, f, _ = self._static_feed(feed, False)
                out = self.net.forward(f, False)
                pred = torch.sigmoid(out["logit"]) if self.hparams.method == "classification" else out["logit"]
                acc.append(pred, f["labels"], f["users_rows"] if "users_rows" in f else f["users"])

            # the iterator's batches (hparams.batch_size LINES each) are scored several at a time: a forward pass is
            # ~100 launches whatever its size, and nothing here waits for the device between batches
            pend, rows, target = [], 0, int(os.environ.get("CLSR_EVAL_ROWS", "32768"))
            for batch_data_input in self.iterator.load_data_from_file(
                    filename, min_seq_length=self.min_seq_length, batch_num_ngs=0):
                if not batch_data_input:
                    continue
                feed = self._to_arrays(batch_data_input)
                if pend and (feed.get("hist_group", 0) != pend[0].get("hist_group", 0)
                             or feed["mask"].shape[1] != pend[0]["mask"].shape[1]):
                    score(pend)
                    pend, rows = [], 0
                pend.append(feed)
                rows += int(np.asarray(feed["labels"]).shape[0])
                if rows >= target:
                    score(pend)
                    pend, rows = [], 0
            if pend:
                score(pend)
            return DM.compute(acc, self.hparams, num_ngs + 1, weighted)

    def run_eval(self, filename, num_ngs):
        """auc/logloss + pairwise metrics over groups of ``num_ngs + 1`` lines
        (reference sequential_base_model.py:204-236)."""
        res = self._device_eval(filename, num_ngs, False)
        if res is not None:
            return res
        preds, labels = [], []
        for batch_data_input in self.iterator.load_data_from_file(
                filename, min_seq_length=self.min_seq_length, batch_num_ngs=0):
            if batch_data_input:
                step_pred, step_labels = self.eval(self.sess, batch_data_input)
                preds.extend(np.reshape(step_pred, -1))
                labels.extend(np.reshape(step_labels, -1))
        return self._metrics(None, preds, labels, num_ngs + 1)

    def _metrics(self, users, preds, labels, group):
        res = cal_metric(labels, preds, self.hparams.metrics)
        # the reference reshapes per batch (and silently needs batch_size % group == 0); accumulating
        # first gives the identical grouping without that constraint
        gp = np.reshape(np.asarray(preds), (-1, group))
        gl = np.reshape(np.asarray(labels), (-1, group))
        res.update(cal_metric(gl, gp, self.hparams.pairwise_metrics))    # 2-D arrays: vectorised, same result
        if users is not None:
            res.update(cal_weighted_metric(users, preds, labels, self.hparams.weighted_metrics))
        return res

    def run_weighted_eval(self, filename, num_ngs, calc_mean_alpha=False, manual_alpha=False):
        """run_eval + user-weighted metrics (wauc == the README's GAUC)
        (reference sequential_base_model.py:244-292)."""
        if not calc_mean_alpha:
            res = self._device_eval(filename, num_ngs, True)
            if res is not None:
                return res
        users, preds, labels, alphas = [], [], [], []
        for batch_data_input in self.iterator.load_data_from_file(
                filename, min_seq_length=self.min_seq_length, batch_num_ngs=0):
            if batch_data_input:
                if not calc_mean_alpha:
                    step_user, step_pred, step_labels = self.eval_with_user(self.sess, batch_data_input)
                else:
                    step_user, step_pred, step_labels, step_alpha = self.eval_with_user_and_alpha(
                        self.sess, batch_data_input)
                    alphas.extend(np.reshape(step_alpha, -1))
                users.extend(np.reshape(step_user, -1))
                preds.extend(np.reshape(step_pred, -1))
                labels.extend(np.reshape(step_labels, -1))
        res = self._metrics(users, preds, labels, num_ngs + 1)
        if calc_mean_alpha:
            if manual_alpha:
                alphas = alphas[0]
            res.update(cal_mean_alpha_metric(alphas, labels))
        return res

    def predict(self, infile_name, outfile_name):
        """One score per input line (reference sequential_base_model.py:326-347)."""
        with open(outfile_name, "w") as wt:
            for batch_data_input in self.iterator.load_data_from_file(infile_name, batch_num_ngs=0):
                if batch_data_input:
                    step_pred = np.reshape(self.infer(self.sess, batch_data_input), -1)
                    wt.write("\n".join(map(str, step_pred)))
                    wt.write("\n")
        return self


class CLSRModel(SequentialBaseModel):
    """Reference ``CLSRModel`` (clsr.py).  All graph pieces live in :class:`clsr_amd.net.CLSRNet`."""


class _SiblingModel(SequentialBaseModel):
    """A model of the reference that sits on the same ``SequentialBaseModel`` trunk as CLSR and is built from the same
    kernels (:class:`clsr_amd.seqnet.SeqNet`).  ``train`` returns the base-class 5-list
    ``[update, extra_update_ops, loss, data_loss, summary]`` (reference base_model.py:359-379)."""
    kind = None

    def _make_net(self, hp, dims):
        from clsr_amd.seqnet import SeqNet

        if self._precision != "fp32":
            # (a silent fp32 run would be reported as a speed-mode result)
            raise NotImplementedError("%s: precision=%r is not available for the sibling models -- only CLSRModel has "
                                      "the bf16 speed mode" % (type(self).__name__, self._precision))
        if self._table_dtype != "fp32":
            raise NotImplementedError("%s: table_dtype=%r is available for CLSRModel only" % (type(self).__name__, self._table_dtype))
        return SeqNet(hp, dims, kind=self.kind, device=self._device, seed=self.seed, dedup_histories=self._dedup)

    def train(self, sess, feed_dict):
        self._train_step(self._to_arrays(feed_dict, True))
        with self._stream_ctx():
            ls = self.net.read_losses()
        return [None, [], ls["loss"], ls["data_loss"], None]


class GRU4RecModel(_SiblingModel):
    """Reference ``GRU4RecModel`` (models/sequential/gru4rec.py)."""
    kind = "gru4rec"


class DINModel(_SiblingModel):
    """Reference ``DINModel`` (models/sequential/din.py)."""
    kind = "din"


class A2SVDModel(_SiblingModel):
    """Reference ``A2SVDModel`` (models/sequential/asvd.py)."""
    kind = "a2svd"


class DIENModel(_SiblingModel):
    """Reference ``DIENModel`` (models/sequential/dien.py)."""
    kind = "dien"


class SLI_RECModel(_SiblingModel):
    """Reference ``SLI_RECModel`` (models/sequential/sli_rec.py)."""
    kind = "sli_rec"

    def eval_with_user_and_alpha(self, sess, feed_dict):
        return super(SLI_RECModel, self).eval_with_user_and_alpha(sess, feed_dict)

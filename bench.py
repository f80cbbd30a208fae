#!/usr/bin/env python
"""Benchmark of the CLSR training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (forward, losses, backward, per-tensor clip, Adam incl.
the dense embedding-table sweeps, BN moving statistics) over one synthetic Taobao-shaped batch
(BASELINE.json configs[1]: batch 4096 positives x (1+4) rows, seq_len 50, emb_dim 40), inputs
already resident in HBM.  Rank 0 prints ONE JSON line:
metric = train interactions/sec (interaction = one positive train line).

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL), weak scaling (every rank
owns its own 4096-positive batch); `python bench.py --gpus N` re-executes itself under
torch.distributed.run, and the same file is the per-rank program when the driver launches it that way.
The gradient exchange (dense all-reduce or sparse touched-row exchange per table, see clsr_amd/dp.py)
overlaps the tail of the backward pass; DP steps are launched eagerly.

Extra objects on the line:
  roofline        the embedding-history gather (north-star kernel; HBM bound): algorithmic bytes per
                  launch / average launch duration measured here with HIP events, on the HBM-resident
                  100M-item catalogue table (N = 1); the cache-resident figure of the benchmarked config is
                  carried inside it.
  roofline_mfma   the most expensive kernel of the step (short-term attention layer-0 GEMM).
  cpu_baseline    the CPU oracle (torch, all host cores) on a bounded sample of the same workload.
  precision_modes step time of the other storage mode (fp32 = parity mode, bf16 = speed mode: bf16 storage of the
                  attention activations + bf16 MFMA, fp32 accumulation / statistics / optimiser).
  extra_workloads BASELINE configs[2] (kuaishou) and configs[4] (catalogue100m, single GPU), and the
                  reference-exact clip mode (history replication, `--exact-clip` makes it the headline).

Other workloads: --config kuaishou (configs[2]) | catalogue100m (configs[4], lazy Adam).
Experiment switches (environment): CLSR_FORCE_DP=1 (DP code path with one rank), CLSR_SPARSE_TABLES=auto|all|none,
CLSR_SPARSE_MODE=allgather|owner, CLSR_DP_GRAPH=1, CLSR_NO_OVERLAP=1 (no side stream), CLSR_DW_EAGER=1 (no batched dW reduction).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # four concurrently active streams per step: see clsr_amd/__init__.py

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIG_INDEX = {"taobao": "1", "kuaishou": "2", "catalogue100m": "4"}


def build_hparams(cfg, batch_size, **over):
    from clsr_amd.deeprec_utils import prepare_hparams

    kw = dict(
        user_vocab="synthetic", item_vocab="synthetic", cate_vocab="synthetic",
        max_seq_length=cfg["T"], batch_size=batch_size, train_num_ngs=4, time_unit="s",
        item_embedding_dim=cfg["Di"], cate_embedding_dim=cfg["Dc"], user_embedding_dim=cfg["Du"],
        hidden_size=cfg["H"], contrastive_loss="triplet", contrastive_length_threshold=5,
        contrastive_recent_k=3, is_clip_norm=1, embed_l2=1e-6, layer_l2=1e-6,
        discrepancy_loss_weight=0.01, contrastive_loss_weight=0.1, learning_rate=0.001,
        sequential_model="time4lstm", save_model=False, write_tfevents=False,
    )
    kw.update(over)
    return prepare_hparams(os.path.join(ROOT, "clsr_amd", "config", "clsr.yaml"), **kw)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def time_kernel(fn, iters=20, warm=3):
    """Average duration (seconds) of fn() launches on the current stream, via HIP events."""
    import torch

    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def cpu_baseline(cfg, seconds=25.0, P=None, warmup=1, steps=None):
    """The oracle (torch-CPU fp32 restatement of the reference graph, dense-Adam semantics) timed on the host cores.

    Protocol (BASELINE.md section 2 / SURVEY 8d): 5 warm-ups + >= 20 timed steps, median.  One step of the benchmarked
    batch (4096 positives x5 rows) costs ~36 s on the host, so the DEFAULT run is bounded: the benchmarked batch itself,
    ``warmup`` (1) untimed step, then THREE timed steps, MEDIAN step time (~2.5 min; a smaller batch is NOT representative:
    1024 positives per step measured 64.8 interactions/s against 113 for the full batch -- the per-step costs of the 50
    time steps do not shrink with the batch).  ``--cpu-warmup 5 --cpu-steps 20`` runs the full protocol (~15 min; that
    run is committed as profiles/r03_cpu_baseline_full_protocol.json: 113.25 interactions/s, median 36.17 s/step).  The
    oracle runs with its input-side RNN projections hoisted out of the T loop (oracle.FAST_RNN: same math, one batched
    product per encoder instead of T small ones)."""
    import torch
    from oracle import clsr_oracle as O
    from clsr_amd.synthetic import synthetic_feed

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)
    torch.set_num_threads(cores)
    P = P or cfg["P"]
    big = cfg["Vi"] >= 10_000_000   # catalogue configs: lazy Adam is mandatory, ids uniform
    hp = build_hparams(cfg, P, **({"optimizer": "lazyadam"} if big else {}))
    dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
    params = O.init_params(dims, hp, seed=0)
    bn, adam = O.init_bn_state(params), O.init_adam(params)
    feed = O.to_torch_feed(synthetic_feed(P, cfg["T"], cfg["Vu"], cfg["Vi"], cfg["Vc"], lengths="full"))
    O.FAST_RNN = True
    times = []
    try:
        tw = time.perf_counter()
        for i in range(warmup):
            params, bn, adam, _, _, _, _ = O.train_step(params, bn, adam, i + 1, feed, hp)
        log("cpu baseline: %d warm-up step(s) took %.1fs on %d threads" % (warmup, time.perf_counter() - tw, cores))
        t0 = time.perf_counter()
        while True:
            t1 = time.perf_counter()
            params, bn, adam, _, _, _, _ = O.train_step(params, bn, adam, warmup + len(times) + 1, feed, hp)
            times.append(time.perf_counter() - t1)
            n, dt = len(times), time.perf_counter() - t0
            if (steps is not None and n >= steps) or (steps is None and ((n >= 3 and dt > seconds) or n >= 20)):
                break
    finally:
        O.FAST_RNN = False
    med = float(np.median(times))
    full = warmup >= 5 and n >= 20 and P >= cfg["P"]
    return dict(value=round(P / med, 2), unit="interactions/s", cores=cores, kind="port",
                ms_per_step=round(med * 1e3, 1), timed_steps=n, positives_per_step=P,
                step_ms_min_max=[round(min(times) * 1e3, 1), round(max(times) * 1e3, 1)],
                sample="%d warm-up + %d timed steps, MEDIAN step time; each step = %d positives x5 rows of the benchmarked "
                       "workload (seq_len %d, same tables, same row group; the GPU step holds %d positives); "
                       "oracle/clsr_oracle.py, torch-CPU fp32, %d threads, input projections hoisted out of the T loop%s; "
                       "the reference's TF-1.15 CPU path cannot run here"
                       % (warmup, n, P, cfg["T"], cfg["P"], cores,
                          "" if full else "; bounded sample instead of the 5 + 20 steps of BASELINE.md section 2 (~15 min: "
                                          "--cpu-warmup 5 --cpu-steps 20; that run, 113.25 interactions/s: "
                                          "profiles/r03_cpu_baseline_full_protocol.json)"))


class Workload(object):
    """One (config, model, precision, clip mode) training workload with its feed resident in HBM."""

    def __init__(self, config, model="clsr", precision="fp32", dedup=True, lengths="full", rank=0, local_rank=0,
                 table_dtype="fp32"):
        import torch
        from clsr_amd.net import CLSRNet
        from clsr_amd.synthetic import CONFIGS, synthetic_feed

        self.name, self.model, self.precision, self.dedup = config, model, precision, dedup
        cfg = self.cfg = CONFIGS[config]
        self.P, self.T, self.G = cfg["P"], cfg["T"], 5
        self.big = cfg["Vi"] >= 10_000_000   # catalogue configs: lazy Adam is mandatory, ids uniform
        big = self.big
        dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
        dev = "cuda:%d" % local_rank
        if model == "clsr":
            self.hp = build_hparams(cfg, self.P, **({"optimizer": "lazyadam"} if big else {}))
            self.net = CLSRNet(self.hp, dims, device=dev, seed=0, dedup_histories=dedup, precision=precision,
                               table_dtype=table_dtype)
        else:
            from clsr_amd.seqnet import SeqNet

            self.hp = build_hparams(cfg, self.P, model_type=model, user_embedding_dim=16,
                                    attention_size=cfg["Di"] + cfg["Dc"], **({"optimizer": "lazyadam"} if big else {}))
            self.net = SeqNet(self.hp, dims, kind=model, device=dev, seed=0)
        if os.environ.get("CLSR_NO_OVERLAP"):
            self.net.overlap = False
        if os.environ.get("CLSR_DW_EAGER"):
            self.net.defer_dw = False
        self.feed = synthetic_feed(self.P, self.T, cfg["Vu"], cfg["Vi"], cfg["Vc"], G=self.G, lengths=lengths,
                                   seed=20220425 + rank, ids="uniform" if big else "zipf")
        self.f = self.net.upload(self.feed, True)
        self.stepper = None
        self.torch = torch

    def describe(self):
        cfg = self.cfg
        return ("BASELINE configs[%s]: %s %s train step, batch %d positives x5 rows (B=%d), seq_len %d, "
                "Di/Dc/Du/H=%d/%d/%d/%d, Vu/Vi/Vc=%d/%d/%d, %s%s" % (
                    CONFIG_INDEX.get(self.name, "?"), self.name,
                    {"clsr": "CLSR", "gru4rec": "GRU4Rec (sibling model)", "din": "DIN (sibling model)",
                     "sli_rec": "SLi-Rec (sibling model)", "a2svd": "A2SVD (sibling model)",
                     "dien": "DIEN (sibling model, relu)"}[self.model],
                    self.P, self.P * self.G, self.T, cfg["Di"], cfg["Dc"], cfg["Du"] if self.model == "clsr" else 16,
                    cfg["H"], cfg["Vu"], cfg["Vi"], cfg["Vc"],
                    "time4lstm + triplet, " if self.model == "clsr" else "",
                    "lazy Adam (row lists)" if self.big else "dense Adam"))

    def step(self):
        if self.stepper is None:
            self.net.train_step(self.f)
        else:
            self.stepper.train_step(self.f)

    def run(self, steps, warmup, dist=None, graph=False, host_losses=None):
        """``warmup`` untimed steps, then exactly ``steps`` timed ones bracketed by barrier + device sync on both
        sides; returns seconds (max over ranks)."""
        from clsr_amd import ops
        torch = self.torch

        stream = torch.cuda.current_stream()
        for i in range(2):
            self.step()
            stream.synchronize()
        # data-parallel runs stay eager: the RCCL watchdog thread of torch.distributed polls its events while a
        # stream capture is open, which can invalidate the capture (hipErrorCapturedEvent, seen on the catalogue
        # config); eager launches cost ~1-2 ms of host time per step and are hidden behind the device step
        self.use_graph = graph and (self.stepper is None or bool(os.environ.get("CLSR_DP_GRAPH")))
        if not self.use_graph:
            run = self.step
        elif self.stepper is None:
            ops.graph_begin()
            self.net.train_step(self.f)
            g = ops.graph_end()
            run = lambda: ops.graph_launch(g)
        else:
            run = self.stepper.capture(self.f)
        for _ in range(warmup):
            run()
        stream.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
            if host_losses is not None:
                host_losses.copy_(self.net.losses, non_blocking=True)  # what CLSRModel.train() returns each step
        stream.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if self.net is not None and hasattr(self.net, "read_losses"):
            self.net.read_losses()      # (raises if a grid barrier of the fused heads timed out: the numbers would be invalid)
        self.rank_seconds = [dt]
        if dist is not None:
            every = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(dist.get_world_size())]
            dist.all_gather(every, torch.tensor([dt], dtype=torch.float64, device="cuda"))
            self.rank_seconds = [float(t) for t in every]
            dt = max(self.rank_seconds)
        return dt

    def run_median(self, steps, warmup, reps=3):
        """Secondary workloads only (never the headline): the median of ``reps`` timed regions of ``steps`` steps.  A
        single 25-40 ms region right after the isolated-kernel loops of the previous workload came out 15-70 % slow in
        about one default run of four in round 6 (the isolated kernels timed in the same window were ~20 % slow too:
        the device's clocks, not this code); the median of three regions does not move with that."""
        ds = sorted(self.run(steps, warmup if i == 0 else 1) for i in range(reps))
        self.reps_ms = [round(d * 1e3 / steps, 4) for d in ds]
        return ds[len(ds) // 2]

    def free(self):
        # the blocks stay with torch's caching allocator (the next workload reuses them): memory handed back to the
        # driver is unmapped in the background and perturbs what is timed right after
        self.torch.cuda.synchronize()
        self.net = self.f = self.stepper = None


NSETS = 3      # id sets the isolated kernel timings rotate through: the touched working set (>= 3 x 184 MB at the catalogue)
               # exceeds the 256 MiB Infinity Cache, so that neither the timing nor FETCH_SIZE can be cache hits of the
               # previous launch


def rotating_id_sets(f, cfg, dev):
    """NSETS (item_history, item_cate_history) pairs of the feed's shape: the feed's own ids, then fresh uniform draws."""
    import torch

    sets = [(f["item_history"], f["item_cate_history"])]
    for j in range(1, NSETS):
        g = torch.Generator(device=dev).manual_seed(977 + j)
        ih = torch.randint(1, cfg["Vi"], tuple(f["item_history"].shape), generator=g, device=dev, dtype=torch.int32)
        ch = torch.randint(1, cfg["Vc"], tuple(f["item_cate_history"].shape), generator=g, device=dev, dtype=torch.int32)
        sets.append((ih, ch))
    return sets


def gather_roofline(net, f, cfg, G, feed, big):
    """HIP-event timing of the history gather of ``net`` on its own tables (SURVEY 8d byte formula), rotating through
    NSETS id sets with an output tensor each."""
    import torch

    from clsr_amd import ops

    T, Hn = cfg["T"], cfg["P"]
    D = cfg["Di"] + cfg["Dc"]
    hm, hr = net._buf("hist_mean", Hn, D), net._buf("hist_recent", Hn, D)
    tb16 = bool(getattr(net, "table_bf16", False))
    sets = rotating_id_sets(f, cfg, net.device)
    outs = [net._buf("hist", Hn, T, D)] + [torch.empty(Hn, T, D, device=net.device) for _ in range(NSETS - 1)]
    turn = [0]

    def gather():
        j = turn[0] % NSETS
        turn[0] += 1
        ih, ch = sets[j]
        if tb16:
            ops.call("clsr_gather_hist_fwd_h", net.tables["item"], net.tables["cate"], ih, ch, G * T, f["seq_len"], G, Hn, T,
                     cfg["Di"], cfg["Dc"], 3, outs[j], 0, hm, hr)
        else:
            ops.call("clsr_gather_hist_fwd", net.tables["item"], net.tables["cate"], ih, ch, G * T, f["seq_len"], G, Hn, T,
                     cfg["Di"], cfg["Dc"], 3, outs[j], hm, hr)

    t_gather = time_kernel(gather, iters=21)
    lens = np.asarray(feed["mask"]).sum(1)[::G]
    n_valid = float(lens.sum())
    # bytes_gather_fwd(n) = n*(Di+Dc)*(s_t + s_a) + 2*n*4 per gathered history row
    gbytes = n_valid * D * ((2 if tb16 else 4) + 4) + 2 * n_valid * 4
    return dict(bound="hbm", kernel="gather_hist_fwd_h_kernel (bf16 tables, fp32 hist)" if tb16 else "gather_hist_fwd_kernel", achieved=round(gbytes / t_gather / 1e9, 1),
                peak=8000.0, unit="GB/s", frac=round(gbytes / t_gather / 8e12, 4),
                bytes_per_launch=gbytes, us_per_launch=round(t_gather * 1e6, 2),
                id_sets_rotated=NSETS, working_set_bytes=NSETS * gbytes)


def embedding_rooflines(net, f, cfg, G, feed):
    """HIP-event timing of the other HBM-bound embedding kernels on the net's own (HBM-resident) tables, with the byte
    formulas of SURVEY 8d: the sorted segmented gradient of the history lookups (gather backward), the lazy-Adam row
    update of the touched rows, and the bf16-table forms (bf16 tables + bf16 hist / d(hist)).  After at least one
    training step on ``f`` (sorted id lists, involved-row lists and gradient buffers exist)."""
    import torch

    from clsr_amd import ops

    out = {}
    T, Hn = cfg["T"], cfg["P"]
    Di, Dc = cfg["Di"], cfg["Dc"]
    D, n = Di + Dc, cfg["P"] * cfg["T"]
    lens = np.asarray(feed["mask"]).sum(1)[::G]
    n_valid = float(lens.sum())
    dev = net.device
    # ---- gather backward: bytes_gather_bwd(n) = n*D*s_a (gradient read) + n*D*4 (fp32 row-gradient write) + 2*n*4
    # NSETS rotating sorted lists (the step's own, then fresh draws with the same structure: n unique-ish history ids + the
    # B target rows, P distinct ids x G) with a gradient tensor each: 3 x 175 MB touched per round > the Infinity Cache
    nk = n + (cfg["P"] * G if getattr(net, "det_grads", False) and net._det_merged(f) else 0)
    tg = net.tab_grad
    B = cfg["P"] * G
    ne = nk                                 # n history slices (+ B target rows when the net chains the two sites)
    merged = ne > n
    lists = [dict(ki=net._buf("sort.keys.item", nk, dtype=torch.int32), pi=net._buf("sort.perm.item", nk, dtype=torch.int32),
                  kc=net._buf("sort.keys.cate", nk, dtype=torch.int32), pc=net._buf("sort.perm.cate", nk, dtype=torch.int32))]
    for j in range(1, NSETS):
        g = torch.Generator(device=dev).manual_seed(4242 + j)
        ent = {}
        for tag_, V in (("i", cfg["Vi"]), ("c", cfg["Vc"])):
            ids = torch.randint(1, V, (n,), generator=g, device=dev, dtype=torch.int64)
            if merged:
                tgt = torch.randint(1, V, (cfg["P"],), generator=g, device=dev, dtype=torch.int64)
                ids = torch.cat([ids, tgt[torch.randint(0, cfg["P"], (B,), generator=g, device=dev)]])
            k_, p_ = torch.sort(ids, stable=True)
            ent["k" + tag_], ent["p" + tag_] = k_.int(), p_.int()
        lists.append(ent)
    dh32 = [torch.randn(Hn * T, D, device=dev) * 1e-3 for _ in range(NSETS)]
    dh16 = [d.to(torch.bfloat16) for d in dh32]
    dtarget = torch.randn(B, D, device=dev) * 1e-3

    # the step's own call of the item site while the contrastive loss is on: the long-term branch's d(hist) as a second
    # gradient tensor and the mean / recent-k shares of the history prologue ride in the walk (the full instantiation,
    # ss_chunks_kernel: 20 registers per entry in flight instead of 4)
    dh_lt = torch.randn(Hn * T, D, device=dev) * 1e-3
    dmean, drecent = torch.randn(Hn, D, device=dev) * 1e-3, torch.randn(Hn, D, device=dev) * 1e-3
    ss_full = torch.zeros(4, dtype=torch.float64, device=dev)

    def site_rows(d, L, full=False):
        bf = int(d.dtype == torch.bfloat16)
        if full:
            tail_i = (1, dtarget.data_ptr(), 0, n, D, 0) if merged else (0,)
            pad = lambda t: t + (0,) * (6 - len(t))
            tail_i = (1, dtarget.data_ptr(), ss_full[2:].data_ptr(), n, D, 0) if merged else (0,)      # (with the two squared norms)
            return [(d.data_ptr(), dh_lt.data_ptr(), dmean.data_ptr(), drecent.data_ptr(), L["ki"].data_ptr(), L["pi"].data_ptr(),
                     f["seq_len"].data_ptr(), tg["item"].data_ptr(), ss_full.data_ptr(), ne, bf, G, T, D, 0, Di, 3, Di, 0) + pad(tail_i) + (0, 0)]
        tail_i = (1, dtarget.data_ptr(), 0, n, D, 0) if merged else (0,)
        tail_c = (1, dtarget.data_ptr(), 0, n, D, Di) if merged else (0,)
        pad = lambda t: t + (0,) * (6 - len(t))
        return [(d.data_ptr(), 0, 0, 0, L["ki"].data_ptr(), L["pi"].data_ptr(), f["seq_len"].data_ptr(), tg["item"].data_ptr(), 0,
                 ne, bf, G, T, D, 0, Di, 3, Di, 0) + pad(tail_i) + (0, 0),
                (d.data_ptr(), 0, 0, 0, L["kc"].data_ptr(), L["pc"].data_ptr(), f["seq_len"].data_ptr(), tg["cate"].data_ptr(), 0,
                 ne, bf, G, T, D, Di, Dc, 3, Dc, 0) + pad(tail_c) + (0, 0)]

    def bwd(ds, only_item):
        jobs = []
        for d, L in zip(ds, lists):
            rows = site_rows(d, L, True) if only_item == "full" else site_rows(d, L)[:1] if only_item else site_rows(d, L)
            jobs.append((rows, torch.zeros(ops.segsum_workspace_bytes(rows), dtype=torch.uint8, device=dev)))
        turn = [0]

        def run():
            rows, ws = jobs[turn[0] % NSETS]
            turn[0] += 1
            ops.segsum_multi(rows, ws)
        return run

    def clear_grads():      # the timed launches wrote the gradient tables: put the touched rows back to zero
        tg["cate"].zero_()
        for L in lists:
            tg["item"].index_fill_(0, L["ki"].long(), 0.0)

    det = getattr(net, "det_grads", False)
    # the HBM claim is about the ITEM table (38 GB, rows of Di * 4 bytes, uniform ids); the category table (1.3 MB) is cache
    # resident and its site runs beside the item site on another stream in the step -- timed here as a second entry
    for tag, ds, sa, only_item in (("gather_bwd", dh32, 4, True), ("gather_bwd_bf16_dhist", dh16, 2, True),
                                   ("gather_bwd_item_and_category_one_stream", dh32, 4, False),
                                   ("gather_bwd_step_form", dh32, 4, "full")):
        if not det:
            out[tag] = dict(skipped="CLSR_NO_DET_GRADS: the counting-sort + atomics path is not measured here")
            continue
        t = time_kernel(bwd(ds, only_item), iters=21)
        W = Di if only_item else D
        # n slices of W values read (s_a bytes) and written once (fp32) + (key, slice) index pairs; the B target rows' slices
        # ride in the same list (fp32 read + fp32 write)
        nbytes = n_valid * W * sa + n_valid * W * 4 + 2 * n_valid * 4 + ((B * W * (4 + 4) + 2 * B * 4) if merged else 0)
        if not only_item:
            nbytes += 2 * n_valid * 4 + (2 * B * 4 if merged else 0)      # (the second site's index pairs)
        if only_item == "full":
            nbytes += n_valid * W * 4 + 2 * Hn * W * 4                    # (second gradient tensor + the mean / recent rows)
        out[tag] = dict(bound="hbm", kernel="ss_chunks_lean_kernel (csrc/segsum.hip: deterministic segmented sums in ONE launch, history "
                                            "+ target slices of a table in one sorted list, every row stored once)",
                        achieved=round(nbytes / t / 1e9, 1), peak=8000.0, unit="GB/s", frac=round(nbytes / t / 8e12, 4),
                        bytes_per_launch=nbytes, us_per_launch=round(t * 1e6, 2), id_sets_rotated=NSETS,
                        formula="n*W*%d (gradient read) + n*W*4 (fp32 row-gradient write) + 2*n*4 (+ the target rows' slices)"
                                % sa, columns=W, sites_merged=bool(merged))
        if tag == "gather_bwd":
            out[tag].update(traffic=167.7e6, traffic_source="profiles/r06_item_embed_kernel_trace.md, counters in KiB (52 dispatches of ss_chunks_lean_kernel over "
                                                            "three rotating lists, fp32 and bf16 d(hist) alternating: WRITE_SIZE 89.1 MB + 2 x FETCH_SIZE 39.3 MB; kernel "
                                                            "time avg 39.4 us)")
        elif tag == "gather_bwd_step_form":
            out[tag].update(kernel="ss_chunks_kernel (the full instantiation: what the item site of a training step with the contrastive "
                                   "loss runs -- a second gradient tensor and the mean / recent-k shares added in the walk, the squared norms of both "
                                   "sites folded by ss_fold_kernel behind it)",
                            formula="n*W*4 (d(hist)) + n*W*4 (long-term d(hist)) + n*W*4 (row-gradient write) + 2*n*4 + 2*Hn*W*4")
        elif tag == "gather_bwd_item_and_category_one_stream":
            out[tag].update(traffic=209.5e6, traffic_source="profiles/r06_embed_kernel_trace.md (52 dispatches, counters in KiB, kernel time avg 56.1 us)")
        clear_grads()
    del dh32, dh16
    # ---- bf16 tables + bf16 hist: bytes_gather_fwd(n) = n*D*(2 + 2) + 2*n*4
    try:
        it_h, ct_h = net.tables["item"].to(torch.bfloat16), net.tables["cate"].to(torch.bfloat16)
        sets = rotating_id_sets(f, cfg, dev)
        hist_h = [torch.empty(Hn * T, D, device=dev, dtype=torch.bfloat16) for _ in range(NSETS)]
        hm, hr = net._buf("hist_mean", Hn, D), net._buf("hist_recent", Hn, D)
        turn = [0]

        def fwd_h():
            j = turn[0] % NSETS
            turn[0] += 1
            ops.call("clsr_gather_hist_fwd_h", it_h, ct_h, sets[j][0], sets[j][1], G * T, f["seq_len"], G, Hn, T, Di, Dc, 3,
                     hist_h[j], 1, hm, hr)

        t = time_kernel(fwd_h, iters=21)
        nbytes = n_valid * D * (2 + 2) + 2 * n_valid * 4
        out["gather_fwd_bf16_tables"] = dict(bound="hbm", kernel="gather_hist_fwd_h_kernel (bf16 tables, bf16 hist)",
                                             achieved=round(nbytes / t / 1e9, 1), peak=8000.0, unit="GB/s",
                                             frac=round(nbytes / t / 8e12, 4), bytes_per_launch=nbytes,
                                             us_per_launch=round(t * 1e6, 2), formula="n*D*(2 + 2) + 2*n*4")
    except RuntimeError as e:
        out["gather_fwd_bf16_tables"] = dict(skipped=str(e)[:120])
        it_h = None
    # ---- lazy-Adam rows: touched_rows * Drow * (3 reads + 3 writes + gradient read + gradient clear) * 4
    try:
        tb, m, v, fl = net.tables["item"], net.tab_m["item"], net.tab_v["item"], net.tab_flags["item"]
        fl.index_fill_(0, lists[0]["ki"].long(), 1)      # (the step cleared its involved-row marks: the rows of this batch again)
        ids, count, cap = net._involved_list("item")
        nrows = int(count[0].item())
        ss = torch.ones(1, dtype=torch.float64, device=dev)
        state0 = net.adam_state.clone()
        state0[3] = 0.0                          # learning rate 0: the timed launches leave the weights where they are

        def adam():
            ops.call("clsr_table_adam_rows", tb, tg["item"], m, v, fl, ids, count, cap, Di, ss, 1, 1, 2.0, state0,
                     0.9, 0.999, 1e-8)

        t = time_kernel(adam)
        nbytes = float(nrows) * Di * 8 * 4
        out["adam_rows"] = dict(bound="hbm", kernel="table_adam_rows_kernel<4,2> (item table, %d touched rows of %d B)"
                                % (nrows, Di * 4), achieved=round(nbytes / t / 1e9, 1), peak=8000.0, unit="GB/s",
                                frac=round(nbytes / t / 8e12, 4), bytes_per_launch=nbytes, us_per_launch=round(t * 1e6, 2),
                                formula="touched_rows*Drow*(param, m, v read + write; gradient read + clear)*4",
                                note="one launch touches 691 MB of rows: nothing of it survives in the 256 MiB Infinity Cache until the next",
                                traffic=704.5e6, traffic_source="profiles/r06_item_embed_kernel_trace.md, counters in KiB (24 dispatches, 224 995 rows: "
                                                                "WRITE_SIZE 352.7 MB + 2 x FETCH_SIZE 175.9 MB; algorithmic 691 MB)")
        if it_h is not None:
            t = time_kernel(lambda: ops.call("clsr_table_adam_rows_h", it_h, tg["item"], m, v, fl, ids, count, cap, Di, ss, 1,
                                             1, 2.0, state0, 0.9, 0.999, 1e-8))
            nbytes = float(nrows) * Di * (6 * 4 + 2 * 2)
            out["adam_rows_bf16_table"] = dict(bound="hbm", kernel="table_adam_rows_h_kernel<2> (bf16 item rows, fp32 moments)",
                                               achieved=round(nbytes / t / 1e9, 1), peak=8000.0, unit="GB/s",
                                               frac=round(nbytes / t / 8e12, 4), bytes_per_launch=nbytes,
                                               us_per_launch=round(t * 1e6, 2),
                                               formula="touched_rows*Drow*(m, v read + write, gradient read + clear: 6*4; "
                                                       "bf16 parameter read + write: 2*2)")
    except (RuntimeError, KeyError, AttributeError) as e:
        out["adam_rows"] = dict(skipped="%s: %s" % (type(e).__name__, str(e)[:120]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="taobao")
    ap.add_argument("--lengths", default="full", choices=["full", "lognormal"])
    ap.add_argument("--model", default="clsr", choices=["clsr", "gru4rec", "din", "sli_rec", "a2svd", "dien"],
                    help="clsr = the BASELINE metric (default); the sibling models run the same step machinery "
                         "(clsr_amd/seqnet.py) and report the same metric for comparison")
    ap.add_argument("--precision", default=os.environ.get("CLSR_PRECISION", "fp32"), choices=["fp32", "fp32x3", "bf16"],
                    help="fp32 = the reference's arithmetic: every product at fp32 accuracy (parity mode, headline); fp32x3 = fp32 "
                         "storage with two-piece split-bf16 products (2^-16 per term) in the recurrences / attention backward / "
                         "encoder tail; bf16 = speed mode: bf16 storage of the attention activations + bf16 MFMA with fp32 "
                         "accumulation, statistics and optimiser")
    ap.add_argument("--table-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: embedding tables stored as bf16 (SURVEY 8d 'bf16 tables'); gradients / Adam moments fp32")
    ap.add_argument("--exact-clip", action="store_true",
                    help="time the reference-exact clip mode (histories replicated like the reference iterator: "
                         "tf.clip_by_norm sees the un-summed replica slices) instead of the de-duplicated step")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as ONE captured hipGraph instead of launching it eagerly")
    ap.add_argument("--no-graph", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-catalogue", action="store_true",
                    help="skip the HBM-resident (100M-item catalogue) measurements (gather roofline + configs[4] step)")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_workloads / precision_modes")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--cpu-warmup", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=None)
    ap.add_argument("--cpu-P", type=int, default=None, help="positives per CPU-baseline step (default: the benchmarked batch)")
    ap.add_argument("--cpu-only", action="store_true", help="print only the cpu_baseline object (no GPU work)")
    ap.add_argument("--local-bn", action="store_true",
                    help="(N>1) per-rank batch-norm statistics with averaged moving stats (no mid-step collectives) "
                         "instead of the default global statistics (single-device parity)")
    ap.add_argument("--sync-bn", action="store_true", help="(default for N>1; kept for old command lines)")
    args = ap.parse_args()
    if args.cpu_only:
        from clsr_amd.synthetic import CONFIGS as _C

        print(json.dumps(cpu_baseline(_C[args.config], seconds=args.cpu_seconds, warmup=args.cpu_warmup,
                                      steps=args.cpu_steps, P=args.cpu_P)), flush=True)
        return

    import torch
    from clsr_amd import ops
    from clsr_amd.synthetic import CONFIGS

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: re-exec under torch.distributed.run (one process per GPU, RCCL);
        # rank 0 of that job prints the one JSON line
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("--gpus %d: this node exposes %d GPU(s)" % (args.gpus, have))
        import socket

        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # CLSR_BENCH_TRANSPORT=staged (test rig: tests/test_bench_ranks_gpu.py): the ranks of the job share ONE device and their
    # collectives are staged through the host (gloo) -- RCCL cannot connect two ranks on one GPU.  Everything else of the
    # per-rank program (stepper, barriers, per-rank times, the JSON line) is the code the driver's N > 1 runs execute.
    staged = os.environ.get("CLSR_BENCH_TRANSPORT") == "staged"
    if staged:
        local_rank = int(os.environ.get("CLSR_BENCH_DEVICE", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    force_dp = bool(os.environ.get("CLSR_FORCE_DP"))   # exercise the DP code path with a single rank
    if world > 1 or force_dp:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        pg_opts = None
        if os.environ.get("CLSR_NCCL_HIGH_PRIORITY"):     # experiment: RCCL's internal stream at high priority
            pg_opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        if staged:
            from clsr_amd.dp import HostStagedDist

            dist.init_process_group("gloo")
            dist = HostStagedDist(dist)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=pg_opts)
        # proof that RCCL really connects the ranks the line claims: a sum of ones over the job
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        rccl_ranks = int(ones.item())
        if rccl_ranks != world and not force_dp:
            raise SystemExit("RCCL all-reduce of ones gave %d, WORLD_SIZE is %d" % (rccl_ranks, world))
    sync_bn = not args.local_bn

    cfg = CONFIGS[args.config]
    wl = Workload(args.config, args.model, args.precision, dedup=not args.exact_clip, lengths=args.lengths, rank=rank,
                  local_rank=local_rank, table_dtype=args.table_dtype)
    net, f, feed = wl.net, wl.f, wl.feed
    use_plans = bool(getattr(net, "use_plans", False))
    P, T, G, big = wl.P, wl.T, wl.G, wl.big
    log("net built")
    if dist is not None:
        from clsr_amd.dp import DataParallel

        wl.stepper = DataParallel(net, dist, sync_bn=sync_bn, sparse_tables=os.environ.get("CLSR_SPARSE_TABLES", "auto"),
                                  sparse_mode=os.environ.get("CLSR_SPARSE_MODE", "allgather"))
        wl.stepper.prepare(f)

    # The step's compute stream outranks its side streams: the weight-gradient kernels beside it are MFMA-saturated and
    # starve whatever shares a CU with them (r03: 3.72 -> 3.63 ms).  Safe under data parallelism only because the step
    # keeps THREE streams of its own (net.stream_alias): with four, RCCL's stream was a fifth hardware queue and the
    # step went from 4.0 to 6.6 ms.  With MORE than one rank the default stays at equal priorities: RCCL's kernels (a few
    # persistent workgroups on a normal-priority stream) would have to find their CUs between the dispatches of a
    # high-priority stream, and that has never run in this build environment -- the measured world-1 result (4.03 ms both
    # ways) says nothing about it.  CLSR_MAIN_PRIORITY overrides (0 = normal, -1 = high).
    multi_rank = wl.stepper is not None and world > 1
    prio = int(os.environ.get("CLSR_MAIN_PRIORITY", "0" if multi_rank else "-1"))
    stream = torch.cuda.Stream(priority=prio)
    host_losses = torch.zeros(8, dtype=torch.float64).pin_memory()
    extra, modes = [], {}
    with torch.cuda.stream(stream):
        dt = wl.run(args.steps, args.warmup, dist=dist, graph=args.graph and not args.no_graph,
                    host_losses=host_losses)
        use_graph = wl.use_graph
        ms = dt * 1e3 / args.steps
        value = world * P * args.steps / dt
        log("timed %d steps: %.3f ms/step" % (args.steps, ms))

        # ---- rooflines of two kernels, measured live with HIP events on this stream
        B, Hn = P * G, P
        D = cfg["Di"] + cfg["Dc"]
        roof = gather_roofline(net, f, cfg, G, feed, big)
        # PMC pass committed in profiles/r02_gather_hist_fwd_pmc_hbm_traffic.csv (same shape): WRITE_SIZE
        # 33.3 MB + 2 x FETCH_SIZE 12.5 MB (gfx950 wide-load correction); reads of the 8 MB tables hit cache
        roof.update(traffic=213.5e6 if big else 59.75e6, traffic_source="counters in KiB; profiles/r06_gather_pmc_{WRITE,FETCH}_SIZE.csv (rotating id sets; first 23 dispatches: this config; last 23: the catalogue)",
                    note=("tables (%.1f MB) are L2/Infinity-Cache resident at this config; the HBM claim needs the "
                          "100M-item config" % ((cfg["Vi"] * cfg["Di"] + cfg["Vc"] * cfg["Dc"]) * 4 / 1e6))
                    if not big else "38 GB item table, uniform ids: every row read is an HBM read")
        roof_mfma, roof_bwd = None, None
        if args.model == "clsr":
            roof_mfma = net.bench_att_layer0(f, time_kernel)
            roof_bwd = net.bench_att_l1_bwd(f, time_kernel) if hasattr(net, "bench_att_l1_bwd") else None
        products_note = net.precision_note() if args.model == "clsr" else None

        single = world == 1 and dist is None and args.model == "clsr" and args.config == "taobao"
        if single and not args.no_extra:
            # ---- the other precision mode of the same workload (named secondary object; `dtype` stays the headline's)
            for other, td in (("bf16", "fp32"), ("bf16", "bf16"), ("fp32x3", "fp32"), ("fp32", "fp32")):
                if other == args.precision and td == args.table_dtype:
                    continue
                tag = other + ("+bf16_tables" if td == "bf16" else "")
                try:
                    w2 = Workload(args.config, args.model, other, dedup=not args.exact_clip, lengths=args.lengths,
                                  table_dtype=td)
                    n2 = max(15, args.steps)
                    d2 = w2.run_median(n2, 5)
                    modes[tag] = dict(ms_per_step=round(d2 * 1e3 / n2, 4), interactions_per_s=round(P * n2 / d2, 1),
                                      timed="median of 3 regions of %d steps" % n2, regions_ms=w2.reps_ms,
                                      note=w2.net.precision_note() + ("; embedding tables stored as bf16 (fp32 gradients and "
                                                                      "Adam moments)" if td == "bf16" else ""))
                    if td == "fp32" and other != "fp32x3":
                        modes[tag]["layer0"] = w2.net.bench_att_layer0(w2.f, time_kernel)
                        l1b = w2.net.bench_att_l1_bwd(w2.f, time_kernel)
                        if l1b is not None:
                            modes[tag]["layer1_bwd"] = l1b
                    log("precision %s: %.3f ms/step" % (tag, d2 * 1e3 / n2))
                    w2.free()
                except NotImplementedError as e:
                    modes[tag] = dict(skipped=str(e)[:200])
            # ---- reference-exact clip mode (or, under --exact-clip, the de-duplicated default)
            w3 = Workload(args.config, args.model, args.precision, dedup=args.exact_clip, lengths=args.lengths)
            n3 = 5 if not args.exact_clip else 10
            d3 = w3.run_median(n3, 2)
            extra.append(dict(workload=w3.describe() + (", histories replicated x5 like the reference iterator "
                                                         "(reference-exact tf.clip_by_norm of the embedding IndexedSlices)"
                                                         if not args.exact_clip else ", histories de-duplicated"),
                              history_dedup=bool(args.exact_clip), ms_per_step=round(d3 * 1e3 / n3, 4),
                              interactions_per_s=round(P * n3 / d3, 1), steps=n3, regions_ms=w3.reps_ms))
            log("history_dedup=%s: %.3f ms/step" % (args.exact_clip, d3 * 1e3 / n3))
            w3.free()
            # ---- BASELINE configs[2]
            w4 = Workload("kuaishou", "clsr", args.precision)
            d4 = w4.run_median(10, 5)
            extra.append(dict(workload=w4.describe(), ms_per_step=round(d4 * 100.0, 4),
                              interactions_per_s=round(w4.P * 10 / d4, 1), steps=10, regions_ms=w4.reps_ms))
            log("kuaishou: %.3f ms/step" % (d4 * 100.0))
            w4.free()
            if args.precision == "fp32":      # (the two-piece split products: comparable with round 5's 3.655 ms)
                w4b = Workload("kuaishou", "clsr", "fp32x3")
                d4b = w4b.run_median(10, 5)
                extra.append(dict(workload=w4b.describe() + ", precision fp32x3 (two-piece split-bf16 products)", precision="fp32x3",
                                  ms_per_step=round(d4b * 100.0, 4), interactions_per_s=round(w4b.P * 10 / d4b, 1), steps=10, regions_ms=w4b.reps_ms))
                log("kuaishou fp32x3: %.3f ms/step" % (d4b * 100.0))
                w4b.free()
            # ---- the same Taobao-shaped batch with realistic (log-normal) history lengths
            other_len = "lognormal" if args.lengths == "full" else "full"
            w6 = Workload(args.config, args.model, args.precision, dedup=not args.exact_clip, lengths=other_len)
            d6 = w6.run_median(10, 3)
            lens6 = np.asarray(w6.feed["mask"]).sum(1)[::w6.G]
            extra.append(dict(workload=w6.describe() + " (%s lengths: mean %.1f of %d steps)" % (other_len, float(lens6.mean()), w6.T),
                              ms_per_step=round(d6 * 100.0, 4), interactions_per_s=round(w6.P * 10 / d6, 1), steps=10, regions_ms=w6.reps_ms))
            log("%s lengths: %.3f ms/step" % (other_len, d6 * 100.0))
            w6.free()
        if world == 1 and dist is None and not args.no_catalogue and not big:
            # BASELINE configs[4] on one GPU: 100M items x 96 floats (38 GB, uniform ids: no cache reuse, every row
            # read is an HBM read) -- the HBM claim of the gather kernel is made HERE (at configs[1]
            # the 8 MB of tables sit in the L2 / Infinity Cache), and the whole step is timed on the same net
            wl.free()
            net = f = None
            try:
                w5 = Workload("catalogue100m", "clsr", args.precision)
                log("catalogue net built")
                big_roof = gather_roofline(w5.net, w5.f, w5.cfg, w5.G, w5.feed, True)
                cache_resident = dict(roof)
                roof = dict(big_roof, workload="BASELINE configs[4] catalogue: 100M items, rows 384 B + 128 B, uniform "
                                               "ids, 4096 histories x 50 steps (38 GB table: every row read is an HBM read)",
                            traffic=213.5e6, traffic_source="counters in KiB; profiles/r06_gather_pmc_WRITE_SIZE.csv + r06_gather_pmc_FETCH_SIZE.csv, the 23 catalogue "
                                                            "dispatches over three rotating id sets (WRITE_SIZE 109.05 MB + 2 x FETCH_SIZE 52.25 MB = 1.01 x the "
                                                            "algorithmic bytes; kernel trace of the same command: profiles/r06_gather_hist_fwd_kernel_trace.md, avg 40.9 us)",
                            in_step_us=36.4, in_step_source="profiles/r06_cat_timeline.txt (rocprofv3 trace of the catalogue step: the gather "
                                                            "dispatch at t = 13.4 us, behind ~9.4 ms of other traffic since the previous step's gather)",
                            cache_resident_at_benchmarked_config=cache_resident)
                try:
                    w5.step()          # (sorted id lists / gradient buffers of the step exist)
                    torch.cuda.synchronize()
                    roof["more"] = embedding_rooflines(w5.net, w5.f, w5.cfg, w5.G, w5.feed)
                    for k_, v_ in roof["more"].items():
                        log("roofline %s: %s" % (k_, {kk: v_.get(kk) for kk in ("achieved", "frac", "us_per_launch", "skipped")}))
                except RuntimeError as e:
                    roof["more"] = dict(skipped=str(e)[:160])
                if not args.no_extra:
                    d5 = w5.run(5, 2)
                    extra.append(dict(workload=w5.describe(), ms_per_step=round(d5 * 200.0, 4),
                                      interactions_per_s=round(w5.P * 5 / d5, 1), steps=5))
                    log("catalogue100m: %.3f ms/step" % (d5 * 200.0))
                w5.free()
                if not args.no_extra and args.precision == "fp32":
                    # the same workload with the two-piece split products (what round 5 called "fp32": comparable with its 8.46 ms)
                    w5b = Workload("catalogue100m", "clsr", "fp32x3")
                    d5b = w5b.run(5, 2)
                    extra.append(dict(workload=w5b.describe() + ", precision fp32x3 (two-piece split-bf16 products)",
                                      precision="fp32x3", ms_per_step=round(d5b * 200.0, 4),
                                      interactions_per_s=round(w5b.P * 5 / d5b, 1), steps=5))
                    log("catalogue100m fp32x3: %.3f ms/step" % (d5b * 200.0))
                    w5b.free()
            except RuntimeError as e:   # not enough free HBM on this device
                roof["hbm_resident_skipped"] = str(e)[:120]

    if rank == 0:
        # (measured LAST: handing its 8 GiB back to the driver is unmapped in the background and slowed whatever was
        # timed in the following ~0.5 s -- two of four default runs of round 6 showed a secondary workload at 3.5-4.1
        # ms instead of 2.33)
        # the measured copy rate of this device next to the 8 TB/s datasheet figure (1 GiB device-to-device
        # copy: 1 GiB read + 1 GiB written per launch)
        copy_peak = None
        try:
            from clsr_amd import ops as _ops

            # the repo's own 16-byte copy kernel (csrc/optim.hip: clsr_copy_words, non-temporal, 4 words in flight per
            # lane, one pass per lane) over 4 GiB -- past the 256 MiB Infinity Cache; swept in scripts/copy_sweep.py:
            # 5.0-5.4 TB/s at 1 GiB, 6.1 TB/s at 4 GiB (a torch copy_ of 1 GiB: 5.2 TB/s, below the gather itself)
            nb_ = 4 << 30
            src_, dst_ = torch.empty(nb_ // 4, device="cuda"), torch.empty(nb_ // 4, device="cuda")
            t_copy = time_kernel(lambda: _ops.call("clsr_copy_words", dst_, src_.data_ptr(), nb_), iters=10, warm=2)
            copy_peak = round(2.0 * nb_ / t_copy / 1e9, 1)
            for r_ in (roof, roof.get("cache_resident_at_benchmarked_config")):
                if r_ is not None:
                    r_["measured_copy_peak_GBps"] = copy_peak
                    r_["frac_of_measured_copy_peak"] = round(r_["achieved"] / copy_peak, 4)
            del src_, dst_
        except RuntimeError:
            pass
        out = {
            "metric": "train interactions/sec @ batch %d seq_len %d" % (P, T), "value": round(value, 1),
            "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32 (every product at fp32 accuracy: fp32-input MFMAs or three bf16 pieces per operand; the two-piece "
                              "split-product step is precision_modes.fp32x3)",
                      "fp32x3": "f32 storage, two-piece split-bf16 products (2^-16 per term) in the recurrences, the attention "
                                "backward and the encoder tail -- NOT the reference's fp32 products (that step: --precision fp32)",
                      "bf16": "bf16"}[args.precision] + (" (bf16 embedding tables)" if args.table_dtype == "bf16" else ""),
            "data": "synthetic",
            "config": {"workload": wl.describe() + " (%s lengths)" % args.lengths,
                       "global_batch": world * P, "seq_len": T,
                       "parallelism": "dp%d" % world if world > 1 else "single",
                       "hipgraph": use_graph, "launch_plan": use_plans and not use_graph,
                       "history_dedup": not args.exact_clip,
                       "history_dedup_note": "forward, losses and summed gradients are identical to the reference's "
                                             "replicated computation; the one deviation is tf.clip_by_norm of the "
                                             "item/cate embedding IndexedSlices (norm of the summed replica slices "
                                             "instead of the un-summed ones), observable only while that clip is "
                                             "active; --exact-clip / extra_workloads times the replicated step",
                       "precision": args.precision, "table_dtype": args.table_dtype,
                       "batch_norm": ("sync" if sync_bn else "per-rank") if world > 1 else "single-device"},
            "rows_per_s": round(value * G, 1),
            "roofline": roof, "roofline_mfma": roof_mfma,
            "loss": float(host_losses[:4].sum()),
        }
        if dist is not None:
            out["config"]["rccl_ranks"] = rccl_ranks
            out["config"]["ms_per_step_by_rank"] = [round(t * 1e3 / args.steps, 4) for t in wl.rank_seconds]
            out["config"]["sparse_tables"] = getattr(wl.stepper, "last_sparse", None)
            out["config"]["sync_bn_statistics_through"] = getattr(wl.stepper, "stats_transport", None)
        if roof_mfma is None:
            del out["roofline_mfma"]
        if args.model == "clsr" and roof_bwd is not None:
            out["roofline_att_bwd"] = roof_bwd
        if products_note:
            out["config"]["products"] = products_note
        if modes:
            out["precision_modes"] = modes
        if extra:
            out["extra_workloads"] = extra
        if world == 1 and not args.no_cpu_baseline and not big and args.model == "clsr":
            out["cpu_baseline"] = cpu_baseline(cfg, seconds=args.cpu_seconds, warmup=args.cpu_warmup,
                                               steps=args.cpu_steps, P=args.cpu_P)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

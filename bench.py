#!/usr/bin/env python
"""Benchmark of the CLSR training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (forward, losses, backward, per-tensor clip, Adam incl.
the dense embedding-table sweeps, BN moving statistics) over one synthetic Taobao-shaped batch
(BASELINE.json configs[1]: batch 4096 positives x (1+4) rows, seq_len 50, emb_dim 40), inputs
already resident in HBM, replayed as one hipGraph.  Rank 0 prints ONE JSON line:
metric = train interactions/sec (interaction = one positive train line).

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL), weak scaling (every rank
owns its own 4096-positive batch); the gradient exchange (dense all-reduce or sparse touched-row
all-gather per table, see clsr_amd/dp.py) runs between backward and update; DP steps are launched eagerly.

Extra objects on the line:
  roofline      the embedding-history gather (north-star kernel; HBM bound): algorithmic bytes per
                launch / average launch duration measured here with HIP events, on the HBM-resident
                100M-item catalogue table (N = 1); the cache-resident figure of the benchmarked config is
                carried inside it.
  roofline_mfma the most expensive kernel of the step (short-term attention layer-0 fp32-MFMA GEMM).
  cpu_baseline  the CPU oracle (torch, all host cores) on a bounded sample of the same workload.

Other workloads: --config kuaishou (configs[2]) | catalogue100m (configs[4], lazy Adam).
Experiment switches (environment): CLSR_FORCE_DP=1 (DP code path with one rank), CLSR_SPARSE_TABLES=auto|all|none,
CLSR_DP_GRAPH=1, CLSR_NO_OVERLAP=1 (no side stream), CLSR_DW_EAGER=1 (no batched dW reduction).
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


def cpu_baseline(cfg, seconds=15.0, P=256):
    """The oracle (torch-CPU fp32 restatement of the reference graph, dense-Adam semantics) timed on
    the host cores on a bounded sample of the same workload (P positives instead of 4096)."""
    import torch
    from oracle import clsr_oracle as O
    from clsr_amd.synthetic import synthetic_feed

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)
    torch.set_num_threads(cores)
    big = cfg["Vi"] >= 10_000_000   # catalogue configs: lazy Adam is mandatory, ids uniform
    hp = build_hparams(cfg, P, **({"optimizer": "lazyadam"} if big else {}))
    dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
    params = O.init_params(dims, hp, seed=0)
    bn, adam = O.init_bn_state(params), O.init_adam(params)
    feed = O.to_torch_feed(synthetic_feed(P, cfg["T"], cfg["Vu"], cfg["Vi"], cfg["Vc"], lengths="full"))
    tw = time.perf_counter()
    O.train_step(params, bn, adam, 1, feed, hp)  # warm-up
    log("cpu baseline warm-up step took %.1fs on %d threads" % (time.perf_counter() - tw, cores))
    n, t0 = 0, time.perf_counter()
    while True:
        params, bn, adam, _, _, _, _ = O.train_step(params, bn, adam, n + 2, feed, hp)
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds or n >= 50:
            break
    return dict(value=round(P * n / dt, 2), unit="interactions/s", cores=cores, kind="port",
                sample="%d steps of batch %d positives x5 rows, seq_len %d, same tables (oracle/clsr_oracle.py, "
                       "torch-CPU fp32, %d threads); the reference's TF-1.15 CPU path cannot run here"
                       % (n, P, cfg["T"], cores))


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
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as ONE captured hipGraph instead of launching it eagerly (slower on "
                         "ROCm 7.2 once the step uses four streams: 5.15 vs 4.63 ms)")
    ap.add_argument("--no-graph", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-catalogue", action="store_true",
                    help="skip the HBM-resident (100M-item catalogue) measurement of the gather kernel")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--sync-bn", action="store_true",
                    help="(N>1) global batch-norm statistics (single-device parity; eager, slower); default is "
                         "per-rank statistics with averaged moving stats")
    args = ap.parse_args()

    import torch
    from clsr_amd import ops
    from clsr_amd.net import CLSRNet
    from clsr_amd.synthetic import CONFIGS, synthetic_feed

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    dist = None
    force_dp = bool(os.environ.get("CLSR_FORCE_DP"))   # exercise the DP code path with a single rank
    if world > 1 or force_dp:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    cfg = CONFIGS[args.config]
    P, T, G = cfg["P"], cfg["T"], 5
    big = cfg["Vi"] >= 10_000_000   # catalogue configs: lazy Adam is mandatory, ids uniform
    hp = build_hparams(cfg, P, **({"optimizer": "lazyadam"} if big else {}))
    dims = dict(Vu=cfg["Vu"], Vi=cfg["Vi"], Vc=cfg["Vc"])
    if args.model == "clsr":
        net = CLSRNet(hp, dims, device="cuda:%d" % local_rank, seed=0)
    else:
        from clsr_amd.seqnet import SeqNet

        hp = build_hparams(cfg, P, model_type=args.model, user_embedding_dim=16,
                           attention_size=cfg["Di"] + cfg["Dc"], **({"optimizer": "lazyadam"} if big else {}))
        net = SeqNet(hp, dims, kind=args.model, device="cuda:%d" % local_rank, seed=0)
    if os.environ.get("CLSR_NO_OVERLAP"):
        net.overlap = False
    if os.environ.get("CLSR_DW_EAGER"):
        net.defer_dw = False
    log("net built")
    feed = synthetic_feed(P, T, cfg["Vu"], cfg["Vi"], cfg["Vc"], G=G, lengths=args.lengths, seed=20220425 + rank,
                          ids="uniform" if big else "zipf")
    f = net.upload(feed, True)
    if dist is not None:
        from clsr_amd.dp import DataParallel

        stepper = DataParallel(net, dist, sync_bn=args.sync_bn,
                               sparse_tables=os.environ.get("CLSR_SPARSE_TABLES", "auto"))
        stepper.prepare(f)
    else:
        stepper = None

    stream = torch.cuda.Stream()
    host_losses = torch.zeros(8, dtype=torch.float64).pin_memory()
    with torch.cuda.stream(stream):
        def eager_step():
            if stepper is None:
                net.train_step(f)
            else:
                stepper.train_step(f)

        for i in range(2):
            eager_step()
            stream.synchronize()
            log("eager step %d done" % i)
        # data-parallel runs stay eager: the RCCL watchdog thread of torch.distributed polls its events while a
        # stream capture is open, which can invalidate the capture (hipErrorCapturedEvent, seen on the catalogue
        # config); eager launches cost ~2 ms of host time per step and are hidden behind the device step
        use_graph = args.graph and not args.no_graph and (stepper is None or bool(os.environ.get("CLSR_DP_GRAPH")))
        if not use_graph:
            run = eager_step
        elif stepper is None:
            ops.graph_begin()
            net.train_step(f)
            graph = ops.graph_end()
            run = lambda: ops.graph_launch(graph)
        else:
            run = stepper.capture(f)
        log("step captured" if use_graph else "eager mode")
        for _ in range(args.warmup):
            run()
        stream.synchronize()
        log("warmup done")
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
            host_losses.copy_(net.losses, non_blocking=True)  # what CLSRModel.train() returns each step
        stream.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        ms = dt * 1e3 / args.steps
        value = world * P * args.steps / dt
        log("timed %d steps: %.3f ms/step" % (args.steps, ms))

        # ---- rooflines of two kernels, measured live with HIP events on this stream
        B, Hn = P * G, P
        D = cfg["Di"] + cfg["Dc"]
        hist = net._buf("hist", Hn, T, D)
        hm, hr = net._buf("hist_mean", Hn, D), net._buf("hist_recent", Hn, D)

        def gather():
            ops.call("clsr_gather_hist_fwd", net.tables["item"], net.tables["cate"], f["item_history"],
                     f["item_cate_history"], G * T, f["seq_len"], G, Hn, T, cfg["Di"], cfg["Dc"], 3, hist, hm, hr)

        t_gather = time_kernel(gather)
        lens = np.asarray(feed["mask"]).sum(1)[::G]
        n_valid = float(lens.sum())
        # bytes_gather_fwd(n) = n*(Di+Dc)*(s_t + s_a) + 2*n*4 per gathered history row
        gbytes = n_valid * D * (4 + 4) + 2 * n_valid * 4
        roof = dict(bound="hbm", kernel="gather_hist_fwd_kernel", achieved=round(gbytes / t_gather / 1e9, 1),
                    peak=8000.0, unit="GB/s", frac=round(gbytes / t_gather / 8e12, 4),
                    # PMC pass committed in profiles/r01_gather_hist_fwd_pmc_hbm_traffic.csv (same shape): WRITE_SIZE
                    # 33.3 MB + 2 x FETCH_SIZE 12.5 MB (gfx950 wide-load correction); reads of the 8 MB tables hit cache
                    traffic=208.8e6 if big else 58.3e6, traffic_source="profiles/r01_gather_hist_fwd_pmc_hbm_traffic.csv",
                    bytes_per_launch=gbytes, us_per_launch=round(t_gather * 1e6, 2),
                    note=("tables (%.1f MB) are L2/Infinity-Cache resident at this config; the HBM claim needs the "
                          "100M-item config" % ((cfg["Vi"] * cfg["Di"] + cfg["Vc"] * cfg["Dc"]) * 4 / 1e6))
                    if not big else "38 GB item table, uniform ids: every row read is an HBM read")
        # the measured copy rate of this device next to the 8 TB/s datasheet figure (1 GiB device-to-device
        # copy: 1 GiB read + 1 GiB written per launch)
        try:
            src_, dst_ = torch.empty(1 << 28, device="cuda"), torch.empty(1 << 28, device="cuda")
            t_copy = time_kernel(lambda: dst_.copy_(src_), iters=10, warm=2)
            roof["measured_copy_peak_GBps"] = round(2.0 * (1 << 30) / t_copy / 1e9, 1)
            roof["frac_of_measured_copy_peak"] = round(roof["achieved"] / roof["measured_copy_peak_GBps"], 4)
            del src_, dst_
        except RuntimeError:
            pass
        if world == 1 and not args.no_catalogue and not big:
            # same kernel on BASELINE configs[4]'s catalogue (100M items x 96 floats + 10k categories x 32, uniform
            # ids: no cache reuse): the table is 38 GB, so every row read is an HBM read
            cat = CONFIGS["catalogue100m"]
            try:
                it = torch.empty(cat["Vi"], cat["Di"], device="cuda").zero_()   # touch every page once
                ct = torch.randn(cat["Vc"], cat["Dc"], device="cuda")
                ii = torch.randint(1, cat["Vi"], (Hn, T), device="cuda", dtype=torch.int32)
                ci = torch.randint(1, cat["Vc"], (Hn, T), device="cuda", dtype=torch.int32)
                ln = torch.full((Hn,), T, device="cuda", dtype=torch.int32)
                Db = cat["Di"] + cat["Dc"]
                hb = torch.empty(Hn, T, Db, device="cuda")
                hmb, hrb = torch.empty(Hn, Db, device="cuda"), torch.empty(Hn, Db, device="cuda")
                t_big = time_kernel(lambda: ops.call("clsr_gather_hist_fwd", it, ct, ii, ci, T, ln, 1, Hn, T,
                                                     cat["Di"], cat["Dc"], 3, hb, hmb, hrb))
                bbytes = Hn * T * (Db * 8 + 8)
                # the HBM claim is made on this measurement (at configs[1] the 8 MB of tables sit in the
                # L2 / Infinity Cache); the cache-resident figure of the benchmarked config stays alongside
                cache_resident = dict(roof)
                roof = dict(
                    bound="hbm", kernel="gather_hist_fwd_kernel",
                    workload="BASELINE configs[4] catalogue: 100M items, rows 384 B + 128 B, uniform ids, "
                             "4096 histories x 50 steps (38 GB table: every row read is an HBM read)",
                    achieved=round(bbytes / t_big / 1e9, 1), peak=8000.0, unit="GB/s",
                    frac=round(bbytes / t_big / 8e12, 4),
                    traffic=208.8e6, traffic_source="profiles/r01_gather_hist_fwd_pmc_hbm_traffic.csv "
                                                    "(WRITE_SIZE 106.5 MB + 2 x FETCH_SIZE 51.1 MB)",
                    bytes_per_launch=float(bbytes), us_per_launch=round(t_big * 1e6, 2),
                    cache_resident_at_benchmarked_config=cache_resident)
                if "measured_copy_peak_GBps" in cache_resident:
                    roof["measured_copy_peak_GBps"] = cache_resident["measured_copy_peak_GBps"]
                    roof["frac_of_measured_copy_peak"] = round(roof["achieved"] / roof["measured_copy_peak_GBps"], 4)
                del it, ct, hb
                torch.cuda.empty_cache()
            except RuntimeError as e:   # not enough free HBM on this device
                roof["hbm_resident_skipped"] = str(e)[:120]
        roof_mfma = None
        if args.model == "clsr":
            Qs, A0 = cfg["Du"] + D, 80
            a_s, q_s = net._buf("st.a", Hn * T, Qs), net._buf("st.q", B, Qs)
            U, V, z0 = net._buf("st.U", Hn * T, A0), net._buf("st.V", B, A0), net._buf("st.z0", B * T, A0)
            Wt, Kp = net.packed["st.Wp"]

            def z0_gemm():
                ops.call("clsr_pgemm", a_s, Qs, T, G, q_s, Qs, None, None, 1, Wt, Kp, None, U, A0, V, A0, z0, A0, 0,
                         None, B * T, Qs, A0)

            t_mm = time_kernel(z0_gemm)
            flops = 2.0 * B * T * Qs * A0
            roof_mfma = dict(bound="mfma", kernel="pgemm_fast_kernel<5,MUL,UV,false> (short-term attention layer 0)",
                             achieved=round(flops / t_mm / 1e12, 2), peak=157.3, unit="TFLOP/s",
                             frac=round(flops / t_mm / 157.3e12, 4), us_per_launch=round(t_mm * 1e6, 2),
                             note="fp32-input MFMA (v_mfma_f32_16x16x4_f32); peak = dense fp32 matrix rate")

    out = None
    if rank == 0:
        out = {
            "metric": "train interactions/sec @ batch %d seq_len %d" % (P, T), "value": round(value, 1),
            "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%s]: %s %s train step, batch %d positives x5 rows "
                                   "(B=%d), seq_len %d (%s lengths), Di/Dc/Du/H=%d/%d/%d/%d, Vu/Vi/Vc=%d/%d/%d, "
                                   "%s%s" % (
                                       {"taobao": "1", "kuaishou": "2", "catalogue100m": "4"}.get(args.config, "?"),
                                       args.config, {"clsr": "CLSR", "gru4rec": "GRU4Rec (sibling model)",
                                                     "din": "DIN (sibling model)",
                                                     "sli_rec": "SLi-Rec (sibling model)",
                                                     "a2svd": "A2SVD (sibling model)",
                                                     "dien": "DIEN (sibling model, relu)"}[args.model],
                                       P, P * G, T, args.lengths, cfg["Di"], cfg["Dc"],
                                       cfg["Du"] if args.model == "clsr" else 16, cfg["H"],
                                       cfg["Vu"], cfg["Vi"], cfg["Vc"],
                                       "time4lstm + triplet, " if args.model == "clsr" else "",
                                       "lazy Adam (row lists)" if big else "dense Adam"),
                       "global_batch": world * P, "seq_len": T,
                       "parallelism": "dp%d" % world if world > 1 else "single",
                       "hipgraph": use_graph, "launch_plan": bool(getattr(net, "use_plans", False)) and not use_graph,
                       "history_dedup": True,
                       "batch_norm": ("sync" if args.sync_bn else "per-rank") if world > 1 else "single-device"},
            "rows_per_s": round(value * G, 1),
            "roofline": roof, "roofline_mfma": roof_mfma,
            "loss": float(host_losses[:4].sum()),
        }
        if roof_mfma is None:
            del out["roofline_mfma"]
        if world == 1 and not args.no_cpu_baseline and not big and args.model == "clsr":
            out["cpu_baseline"] = cpu_baseline(cfg, seconds=args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Single-node data parallelism for the CLSR step: one process per GPU, torch.distributed with the
"nccl" backend (== RCCL over xGMI on ROCm).  The reference has no multi-device code at all
(SURVEY.md section 5/8e); this is new surface.

Per step every rank runs forward + backward on ITS shard of the global batch (whole groups of
1 + train_num_ngs rows stay together), then ONE exchange step sums the gradient state:

  dense_grad      flat fp32 buffer of the ~124 k dense parameters           (all-reduce SUM)
  tab_grad_flat   dense gradient tables of the four embedding tables        (all-reduce SUM)
                  -- at Taobao scale (21 MB) a dense all-reduce is simpler and as fast as a
                  segmented sparse exchange; the sparse row exchange only pays for huge catalogues
  tab_flags_flat  "involved row" byte maps (tf.unique id sets)               (all-reduce MAX == OR)
  small           squared norms of the IndexedSlices pieces + loss numerators (all-reduce SUM)

Tables whose touched rows are few compared with the vocabulary are exchanged SPARSELY instead (the
"segmented sparse reduce" of BASELINE.json:north_star): the byte map is compacted into an ascending id
list, the listed rows are packed, (count, ids, rows) are all-gathered, and every rank adds the lists of
all ranks in rank order into its cleared rows -- bit-identical sums on every replica, and the traffic is
O(touched rows) instead of O(vocabulary).  ``sparse_tables="auto"`` picks per table and per batch shape
(sparse when the table has at least 64 MB and  4 * rows_bound * world < vocabulary: a dense table rides in the ONE
flat all-reduce, a sparse one
costs three collectives of its own, so it has to save real traffic);  "all" / "none" force one path.

and every rank applies the identical clip + Adam update, so replicas never diverge.  Loss
normalisers are global: the softmax data loss is scaled by 1/(P * world) and the contrastive
denominator (rows longer than the threshold) is summed over ranks before the step.

Batch-norm: ``sync_bn=True`` sums the per-feature partial statistics of every BN layer across ranks
(forward sums and backward sums) so the result equals a single-device run on the global batch;
``sync_bn=False`` ("local BN") keeps per-rank statistics (faster: no mid-step collectives, graph
capturable) and averages the moving statistics once per step.
"""
import os

import torch

from clsr_amd import ops

__all__ = ["DataParallel", "HostStagedDist", "allreduce_step_buffers", "allgather_row_lists", "touched_rows_bound"]


class HostStagedDist(object):
    """torch.distributed look-alike that stages GPU tensors through the host (a gloo process group): for rigs whose ranks
    cannot run RCCL between them -- several ranks sharing ONE GPU (tests/test_dp_gpu.py, ``CLSR_BENCH_TRANSPORT=staged`` of
    bench.py).  The blocking device-to-host copy on the issuing stream gives every collective the "ordered after that
    stream's work so far" semantics RCCL has.  Not a production transport."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def get_world_size(self, group=None):
        return self._d.get_world_size()

    def get_rank(self, group=None):
        return self._d.get_rank()

    class _Done(object):
        def wait(self):
            pass

    def all_reduce(self, t, op=None, group=None, async_op=False):
        c = t.detach().cpu()             # blocking copy on the CURRENT stream: ordered after its work so far
        self._d.all_reduce(c, op=self.ReduceOp.SUM if op is None else op)
        t.copy_(c)
        return self._Done() if async_op else None

    def all_gather(self, outs, t, group=None):
        cs = [o.detach().cpu() for o in outs]
        self._d.all_gather(cs, t.detach().cpu())
        for o, c in zip(outs, cs):
            o.copy_(c)

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
        co, ci = out.detach().cpu(), inp.detach().cpu().contiguous()
        self._d.all_to_all_single(co, ci, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes)
        out.copy_(co)

    def barrier(self, group=None):
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()

    def broadcast(self, t, src=0, group=None):
        c = t.detach().cpu()
        self._d.broadcast(c, src=src)
        t.copy_(c)


def allreduce_step_buffers(dist, dense_grad, tab_grad_flat, tab_flags_flat, small, group=None, grad_flat=None):
    """The exchange step.  Works on CPU tensors (gloo) and GPU tensors (nccl/RCCL) alike.  ``grad_flat``: the one
    buffer both gradient views live in (CLSRNet.grad_flat) -- then they travel in ONE collective (three per step
    instead of four; each costs a launch + ring latency whatever its size)."""
    if grad_flat is not None:
        dist.all_reduce(grad_flat, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.all_reduce(dense_grad, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tab_grad_flat, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(tab_flags_flat, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(small, op=dist.ReduceOp.SUM, group=group)


def allgather_row_lists(dist, count, ids, rows, counts_all, ids_all, rows_all, group=None):
    """All-gather one rank's (count[2], ids[cap], rows[cap, C]) into the [world, ...] buffers.  Works on
    CPU tensors (gloo) and GPU tensors (nccl/RCCL); the lists are padded to ``cap`` so no host sync is
    needed to learn the counts."""
    world = counts_all.shape[0]
    dist.all_gather([counts_all[r] for r in range(world)], count, group=group)
    dist.all_gather([ids_all[r] for r in range(world)], ids, group=group)
    dist.all_gather([rows_all[r] for r in range(world)], rows, group=group)


def touched_rows_bound(table, shape, vocab):
    """Upper bound on the rows of ``table`` one rank touches in a step of shape (B, T, G, Hn)."""
    B, T, G, Hn = shape
    bound = Hn if table.startswith("user") else Hn * T + B
    return int(min(vocab, bound))


_HIST_KEYS = ("users", "item_history", "item_cate_history", "mask", "time_from_first_action", "time_to_now")


def shard_feed(feed, rank, world, group_size):
    """Contiguous block of whole groups for ``rank`` out of a global training feed (numpy arrays): the row
    layout of the reference iterator, or the compact layout (``hist_group``: history-level arrays hold one row
    per positive).  A global batch whose number of positives is not a multiple of ``world`` is truncated (the
    last partial batch of an epoch)."""
    hg = int(feed.get("hist_group", 0) or 0)
    if hg:
        if hg != group_size:
            raise ValueError("compact feed of group %d sharded with group size %d" % (hg, group_size))
        n = feed["mask"].shape[0]
        per = n // world
        if per == 0:
            raise ValueError("global batch of %d positives cannot feed %d ranks" % (n, world))
        out = {"hist_group": hg}
        for k, v in feed.items():
            if k == "hist_group":
                continue
            step = 1 if k in _HIST_KEYS else hg
            out[k] = v[rank * per * step:(rank + 1) * per * step]
        return out
    B = feed["labels"].shape[0]
    if B % group_size:
        raise ValueError("global batch rows (%d) must be a multiple of the group size (%d)" % (B, group_size))
    per = (B // group_size // world) * group_size      # whole groups per rank; a remainder is dropped like above
    if per == 0:
        raise ValueError("global batch of %d positives cannot feed %d ranks" % (B // group_size, world))
    return {k: v[rank * per:(rank + 1) * per] for k, v in feed.items()}


class DataParallel(object):
    """Data-parallel stepper around a CLSRNet / SeqNet.

    ``train_step`` = backward on this rank's shard with the exchange OVERLAPPED: the net reports, from inside the
    step and on the stream where it happens, when a piece of the gradient state is final (``net.dp_hooks``), and that
    piece's collective is issued right there with ``async_op=True`` -- RCCL's stream waits for exactly that stream's
    work so far and the step's own streams keep going:
      flags_ready   start of the step (row marks depend on the feed only)  -> byte-map all-reduce under the forward
      table_ready   per table, on the stream that finished it              -> SPARSE tables: their row exchange starts there
      dense_ready   after the batched weight-gradient reduction            -> noted (the stream it is final on)
    ``_finish`` then issues, in THIS order: every run of adjacent dense gradient tables as ONE all-reduce (the compute
    stream has joined the streams that produced them), the 0.5 MB of dense gradients (final ~100 us later, on the
    weight-gradient stream: its collective is queued BEHIND the tables' so that the 20 MB of tables do not wait for
    it -- collectives of a process group run in issue order), the 24 doubles of norms / loss numerators and, with
    per-rank BN, the moving statistics; the compute stream waits for all of them and the identical clip + Adam update
    follows.  Round 2 issued the dense all-reduce and every table's own all-reduce from the hooks: seven collectives at
    the tail, the first of which (dense) held back the others (profiles/r03_dp_world1_timeline.txt).
    Nets without hooks (or ``overlap=False``) run the same collectives back to back after the backward pass."""

    def __init__(self, net, dist, sync_bn=True, group=None, sparse_tables="auto", overlap=True, sparse_mode="allgather",
                 p2p_stats=True):
        if sparse_tables not in ("auto", "all", "none"):
            raise ValueError("sparse_tables must be 'auto', 'all' or 'none'")
        if sparse_mode not in ("allgather", "owner"):
            raise ValueError("sparse_mode must be 'allgather' or 'owner'")
        self.sparse_mode = sparse_mode
        self.net, self.dist, self.group = net, dist, group
        self.sparse_tables = sparse_tables
        self._sparse_bufs = {}
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sync_bn = bool(sync_bn)
        self.overlap = bool(overlap)
        self.coalesce = True        # dense gradients + dense gradient tables in one collective (False: one each)
        net.dp_world = self.world
        # recorded launch plans keep raw communicator pointers: a new stepper on the same net must never replay the plans
        # of an earlier one (a freed communicator's address can be handed out again)
        net.dp_generation = getattr(net, "dp_generation", 0) + 1
        if hasattr(net, "_step_plans"):
            net._step_plans.clear()
        net.dp_stats_hook = self._sum_stats if self.sync_bn else None
        # sync-BN statistics through the peer-to-peer communicator (clsr_amd/p2p.py) when one can be set up: ONE kernel per
        # all-reduce on the issuing stream instead of a torch.distributed call.  CLSR_P2P_STATS=0: always torch.distributed.
        self.comm, self.stats_transport = None, "torch.distributed"
        self.heads_comm = None
        if self.sync_bn and p2p_stats and os.environ.get("CLSR_P2P_STATS", "1") != "0" and self.world <= 8 \
                and torch.cuda.is_available() and str(net.device).startswith("cuda"):
            try:
                from clsr_amd import p2p
                self.comm = p2p.from_process_group(self.rank, self.world, group)
                # self-test before the step depends on it: sums of (rank + 1) * (i + 1) on two streams (two channels),
                # checked on every rank
                ok, why = self._p2p_self_test(), ""
            except Exception as e:      # (no IPC between these devices / processes: the process group does it)
                ok, why = False, str(e)[:80]
            # ... and the verdict is agreed on by ALL ranks: one rank alone on the other transport would wait for
            # collectives its peers never issue
            votes = [None] * self.world
            import torch.distributed as tdist      # (``dist`` may be a host-staged wrapper of the test rigs)
            tdist.all_gather_object(votes, bool(ok), group=group)
            if all(votes):
                net.dp_comm = self.comm.handle
                self.comm.set_abort(net.adam_state[4:])     # a wait that gives up aborts the step (CLSRNet.check_abort)
                self.stats_transport = "p2p"
                # the same primitives (IPC-mapped uncached buffers, system-scope atomics) carry the statistics of the fused
                # heads launches: their group sums are pushed to every rank from inside the launches (csrc/headsfused.hip)
                # ... unless two ranks share ONE device (test rigs): each launch wants a compute unit per workgroup for its
                # whole duration, and two of them waiting for each other cannot both be resident beyond 256 workgroups in all
                # (CLSR_HEADS_COMM_SHARED=1: small batches on a shared device, tests/test_p2p_gpu.py)
                props = torch.cuda.get_device_properties(net.device)
                ident = "%s/%s" % (getattr(props, "uuid", None), getattr(props, "pci_bus_id", net.device))
                idents = [None] * self.world
                tdist.all_gather_object(idents, ident, group=group)
                shared = len(set(idents)) < self.world and os.environ.get("CLSR_HEADS_COMM_SHARED") != "1"
                if os.environ.get("CLSR_HEADS_COMM", "1") != "0" and getattr(net, "heads_fused", False) and not shared:
                    try:
                        self.heads_comm = p2p.heads_comm_from_process_group(self.rank, self.world, group)
                        hok = self.heads_comm.self_test(net.device)
                    except Exception:
                        self.heads_comm, hok = None, False
                    hv = [None] * self.world
                    tdist.all_gather_object(hv, bool(hok), group=group)
                    if all(hv):
                        net.heads_comm = self.heads_comm.handle
                        self.stats_transport = "p2p + fused heads"
                    elif self.heads_comm is not None:
                        self.heads_comm.close()
                        self.heads_comm = None
            else:
                if self.comm is not None:
                    try:
                        self.comm.close()
                    except Exception:
                        pass
                self.comm, net.dp_comm = None, None
                self.stats_transport = "torch.distributed (p2p unavailable: %s)" % (
                    why or "self-test failed on ranks %s" % [r for r, v in enumerate(votes) if not v])
        net.dp_hooks = self if self.overlap else None
        # sumsq_tab (16) + losses (8) travel together
        self.small = net.stats24       # the net's own 24 doubles: summed in place, no staging copies
        self.sparse_min_bytes = 64 << 20   # "auto": tables below this size always go dense
        self._graphs = None
        self._works, self._done = [], set()
        self._dense_stream = None
        self.last_sparse = []
        self.trace = None              # tests: list that receives (event, detail) tuples in issue order
        # asynchronous copy of the step's abort flag (net.adam_state[4]), looked at one step late: no synchronisation
        self._abort_host, self._steps_done = None, 0
        if torch.cuda.is_available() and str(net.device).startswith("cuda") and hasattr(net, "adam_state"):
            self._abort_host = torch.zeros(1, dtype=torch.float64).pin_memory()
        self.broadcast_parameters()

    def broadcast_parameters(self):
        """Replicas start from rank 0's variables (dense buffer, tables, BN moving stats, Adam slots)."""
        net, dist = self.net, self.dist
        tensors = [net.dense, net.dense_m, net.dense_v, net.adam_state]
        tensors += list(net.tables.values()) + list(net.tab_m.values()) + list(net.tab_v.values())
        tensors.append(net.bn_moving)
        for t in tensors:
            dist.broadcast(t, src=0, group=self.group)

    # ---- collectives: issued on (ordered after) ``stream``, never waited for here
    def _on(self, stream):
        import contextlib

        return torch.cuda.stream(stream) if (stream is not None and torch.cuda.is_available()) else contextlib.nullcontext()

    def _allreduce(self, t, op, stream, what):
        if self.trace is not None:
            self.trace.append(("collective", what))
        with self._on(stream):
            w = self.dist.all_reduce(t, op=op, group=self.group, async_op=True)
        self._works.append(w)

    def _sum_stats(self, t, stream=None):
        """SyncBN: the statistics are needed by the very next launch -- a blocking (stream-ordered) all-reduce."""
        with self._on(stream):
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def prepare(self, f):
        """Make the per-batch loss normalisers global (call once per uploaded feed)."""
        self.dist.all_reduce(f["denom"], op=self.dist.ReduceOp.SUM, group=self.group)
        return f

    def _is_sparse(self, name):
        if self.sparse_tables != "auto":
            return self.sparse_tables == "all"
        V, C = self.net.tables[name].shape
        if V * C * 4 < self.sparse_min_bytes:
            # a small table rides in a dense all-reduce for a few tens of microseconds; the sparse route costs
            # ~10 small launches + three collectives per table (measured: +0.1 ms per 6 MB user table)
            return False
        return 4 * touched_rows_bound(name, self.net.last_shape, V) * self.world < V

    # ---- hooks (called by the net from inside the step; ``stream`` = where the data became final)
    def flags_ready(self, stream):
        net = self.net
        self._done.add("flags")
        names = list(net.tab_grad)
        sparse = [n for n in names if self._is_sparse(n)]
        self.last_sparse = sparse
        with self._on(stream):
            for n in sparse:      # the list of THIS rank's touched rows, before any merging
                self._compact_local(n)
        # runs of consecutive dense tables share one collective (their byte maps are adjacent in the flat buffer)
        i = 0
        while i < len(names):
            if names[i] in sparse:
                i += 1
                continue
            j = i
            while j + 1 < len(names) and names[j + 1] not in sparse:
                j += 1
            f0, f1 = net.tab_foff[names[i]], net.tab_foff[names[j]] + net.tab_shape[names[j]][0]
            self._allreduce(net.tab_flags_flat[f0:f1], self.dist.ReduceOp.MAX, stream, "flags")
            i = j + 1

    def dense_ready(self, stream):
        """The dense gradients are final on ``stream``; their all-reduce is issued by ``_finish`` behind the tables'."""
        self._dense_stream = stream
        self._done.add("dense-final")

    def table_ready(self, stream, name):
        net = self.net
        if name in self._done or name not in net.tab_grad:
            return
        if "flags" not in self._done:          # a net that never reported its flags: exchange them first
            self.flags_ready(stream)
        if name in self.last_sparse:
            self._done.add(name)
            with self._on(stream):
                self._exchange_rows(name)
        # (a dense table travels with its neighbours in ONE all-reduce issued by _finish)

    def _dense_table_runs(self):
        """[(first name, element offset, element count)] of the runs of ADJACENT dense tables in tab_grad_flat."""
        net = self.net
        names = list(net.tab_grad)
        runs, i = [], 0
        while i < len(names):
            if names[i] in self.last_sparse:
                i += 1
                continue
            j = i
            while j + 1 < len(names) and names[j + 1] not in self.last_sparse:
                j += 1
            g0 = net.tab_goff[names[i]]
            g1 = net.tab_goff[names[j]] + net.tab_shape[names[j]][0] * net.tab_shape[names[j]][1]
            runs.append((names[i:j + 1], g0, g1 - g0))
            i = j + 1
        return runs

    def _compact_local(self, name):
        net, W = self.net, self.world
        grad, flags = net.tab_grad[name], net.tab_flags[name]
        V, C = grad.shape
        cap = touched_rows_bound(name, net.last_shape, V)   # per rank: the flags are not merged yet
        key = (name, cap)
        b = self._sparse_bufs.get(key)
        if b is None:
            dev, i32 = net.device, torch.int32
            nws = ops.query("clsr_flags_compact_workspace_bytes", V)
            b = self._sparse_bufs[key] = dict(
                ws=torch.empty(nws, dtype=torch.uint8, device=dev), nws=nws,
                count=torch.zeros(2, dtype=i32, device=dev), ids=torch.zeros(cap, dtype=i32, device=dev),
                rows=torch.zeros(cap, C, device=dev), counts_all=torch.zeros(W, 2, dtype=i32, device=dev),
                ids_all=torch.zeros(W, cap, dtype=i32, device=dev), rows_all=torch.zeros(W, cap, C, device=dev))
        self._cur_sparse = getattr(self, "_cur_sparse", {})
        self._cur_sparse[name] = (b, cap)
        ops.call("clsr_flags_compact", flags, V, b["ids"], cap, b["count"], b["ws"], b["nws"],
                 stream=torch.cuda.current_stream().cuda_stream)

    def _exchange_rows_owner(self, name):
        """Owner-routed segmented sparse reduce: every rank OWNS a contiguous range of rows
        [V*r/W, V*(r+1)/W).  (1) the local ascending (ids, rows) list is cut at the range bounds -- contiguous
        slices, no permutation -- and routed to the owners with one all-to-all (each pair of GPUs uses its own xGMI
        link); (2) the owner adds the W incoming slices into its range of the dense gradient table in rank order (ids
        are unique inside a slice: no atomics, the same fp32 sums on every run); (3) it compacts the now globally
        marked rows of its range and (4) the reduced lists are all-gathered and written into every replica.  Rows
        touched by several ranks (popular items) cross the fabric once per toucher on the way in and ONCE on the way
        out, instead of world times in the all-gather exchange; with uniform ids over a 100M-row table there is
        almost nothing to merge and the all-gather exchange moves the same bytes in one hop less.
        Costs two host synchronisations per table and step (slice sizes must be known to size the transfers)."""
        net, W, r, dist = self.net, self.world, self.rank, self.dist
        grad, flags = net.tab_grad[name], net.tab_flags[name]
        V, C = grad.shape
        b, cap = self._cur_sparse[name]
        dev, i32 = net.device, torch.int32
        if self.trace is not None:
            self.trace.append(("collective", "owner-rows:" + name))
        s = torch.cuda.current_stream().cuda_stream
        bounds = b.get("bounds")
        if bounds is None:
            bounds = b["bounds"] = torch.tensor([V * o // W for o in range(W + 1)], dtype=i32, device=dev)
            b["offs"] = torch.zeros(W + 1, dtype=i32, device=dev)
            b["rcnt"] = torch.zeros(W, dtype=i32, device=dev)
            b["cnt2"], b["cnt2_all"] = torch.zeros(2, dtype=i32, device=dev), torch.zeros(W, 2, dtype=i32, device=dev)
        lo, hi = V * r // W, V * (r + 1) // W
        ops.call("clsr_rows_pack", grad, b["ids"], b["count"], cap, C, b["rows"], stream=s)
        ops.call("clsr_range_offsets", b["ids"], b["count"], bounds, W, b["offs"], stream=s)
        scnt = b["offs"][1:] - b["offs"][:-1]
        dist.all_to_all_single(b["rcnt"], scnt.contiguous(), group=self.group)
        offs_h, rcnt_h = b["offs"].cpu().tolist(), b["rcnt"].cpu().tolist()          # host sync 1: slice sizes
        ins = [offs_h[o + 1] - offs_h[o] for o in range(W)]
        n_in = sum(rcnt_h)
        ids_in = torch.empty(max(n_in, 1), dtype=i32, device=dev)
        rows_in = torch.empty(max(n_in, 1), C, device=dev)
        dist.all_to_all_single(ids_in[:n_in], b["ids"][:offs_h[W]], output_split_sizes=rcnt_h, input_split_sizes=ins,
                               group=self.group)
        dist.all_to_all_single(rows_in[:n_in], b["rows"][:offs_h[W]], output_split_sizes=rcnt_h, input_split_sizes=ins,
                               group=self.group)
        # local contributions are in flight / packed: clear them, then the owner sums the W slices in rank order
        ops.call("clsr_rows_unpack", b["ids"], None, b["count"], cap, C, 0, grad, None, stream=s)
        o0 = 0
        for src in range(W):
            if rcnt_h[src]:
                ops.call("clsr_rows_unpack", ids_in[o0:], rows_in[o0:], b["rcnt"][src:], rcnt_h[src], C, 1, grad, flags,
                         stream=s)
            o0 += rcnt_h[src]
        # the owner's reduced list: every marked row of its range (its own marks + what the slices marked)
        cap2 = max(min(hi - lo, cap * W), 1)
        key = ("own", name, cap2)
        ob = self._sparse_bufs.get(key)
        if ob is None:
            nws = ops.query("clsr_flags_compact_workspace_bytes", hi - lo)
            ob = self._sparse_bufs[key] = dict(ws=torch.empty(nws, dtype=torch.uint8, device=dev), nws=nws,
                                               ids=torch.zeros(cap2, dtype=i32, device=dev),
                                               rows=torch.zeros(cap2, C, device=dev))
        ops.call("clsr_flags_compact_off", flags[lo:hi], hi - lo, lo, ob["ids"], cap2, b["cnt2"], ob["ws"], ob["nws"],
                 stream=s)
        ops.call("clsr_rows_pack", grad, ob["ids"], b["cnt2"], cap2, C, ob["rows"], stream=s)
        dist.all_gather([b["cnt2_all"][o] for o in range(W)], b["cnt2"], group=self.group)
        cnt2_h = b["cnt2_all"][:, 0].cpu().tolist()                                     # host sync 2: reduced sizes
        mx = max(max(cnt2_h), 1)
        ids_all = torch.empty(W, mx, dtype=i32, device=dev)
        rows_all = torch.empty(W, mx, C, device=dev)
        pad_i, pad_r = ob["ids"], ob["rows"]
        if mx > cap2:                                                                   # (a peer owns more rows)
            pad_i, pad_r = torch.zeros(mx, dtype=i32, device=dev), torch.zeros(mx, C, device=dev)
            pad_i[:cap2].copy_(ob["ids"])
            pad_r[:cap2].copy_(ob["rows"])
        dist.all_gather([ids_all[o] for o in range(W)], pad_i[:mx], group=self.group)
        dist.all_gather([rows_all[o] for o in range(W)], pad_r[:mx], group=self.group)
        for o in range(W):
            if o != r and cnt2_h[o]:      # (this rank's own range already holds the sums)
                ops.call("clsr_rows_unpack", ids_all[o], rows_all[o], b["cnt2_all"][o], cnt2_h[o], C, 1, grad, flags,
                         stream=s)

    def _exchange_rows(self, name):
        """Sparse exchange of one embedding table's gradient (see the module docstring); runs on the current stream."""
        if self.sparse_mode == "owner" and self.world > 1:
            return self._exchange_rows_owner(name)
        net, W = self.net, self.world
        grad, flags = net.tab_grad[name], net.tab_flags[name]
        V, C = grad.shape
        b, cap = self._cur_sparse[name]
        if self.trace is not None:
            self.trace.append(("collective", "rows:" + name))
        s = torch.cuda.current_stream().cuda_stream
        ops.call("clsr_rows_pack", grad, b["ids"], b["count"], cap, C, b["rows"], stream=s)
        allgather_row_lists(self.dist, b["count"], b["ids"], b["rows"], b["counts_all"], b["ids_all"],
                            b["rows_all"], self.group)
        # own rows are cleared, then the lists of ALL ranks are added in rank order: the same fp32 sums,
        # in the same order, on every replica
        ops.call("clsr_rows_unpack", b["ids"], None, b["count"], cap, C, 0, grad, None, stream=s)
        for r in range(W):
            ops.call("clsr_rows_unpack", b["ids_all"][r], b["rows_all"][r], b["counts_all"][r], cap, C, 1, grad,
                     flags, stream=s)

    # ---- the phases of a step
    def _backward(self, f):
        self._works, self._done = [], set()
        self.net.train_step(f, apply=False)

    def close(self):
        """Release the peer-to-peer communicators (mapped exchange buffers of the peers, own allocations).  Every rank calls
        it at the same point, after its last step; the net goes back to the launch chain / the process group."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.net.dp_comm = None
        self.net.heads_comm = None
        if hasattr(self.net, "_step_plans"):
            self.net._step_plans.clear()       # (they hold the communicators' raw pointers)
        self.net.dp_generation = getattr(self.net, "dp_generation", 0) + 1
        for name in ("heads_comm", "comm"):
            c = getattr(self, name, None)
            if c is not None:
                try:
                    c.close()
                finally:
                    setattr(self, name, None)
        self.stats_transport = "torch.distributed"

    def _p2p_self_test(self):
        """Two small all-reduces on two streams through the communicator against the expected sums."""
        dev = self.net.device
        exp = float(sum(r + 1 for r in range(self.world)))
        a = torch.arange(1, 65, dtype=torch.float64, device=dev) * (self.rank + 1)
        b = torch.arange(1, 9, dtype=torch.float64, device=dev) * (self.rank + 1)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        self.comm.all_reduce(a, 64)
        with torch.cuda.stream(side):
            self.comm.all_reduce(b, 8)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.comm.reset_channels()      # (the step's streams take the channels in THEIR order of first use)
        if self.comm.error() != 0:
            return False
        return bool(torch.equal(a, torch.arange(1, 65, dtype=torch.float64, device=dev) * exp)
                    and torch.equal(b, torch.arange(1, 9, dtype=torch.float64, device=dev) * exp))

    def _finish(self):
        """Everything that has not been exchanged yet, then the compute stream waits for all collectives."""
        net, dist = self.net, self.dist
        if self.trace is not None:
            self.trace.append(("finish", sorted(self._done)))
        stream = ops.current_stream() if torch.cuda.is_available() else None
        if "flags" not in self._done:
            self.flags_ready(stream)
        for name in net.tab_grad:          # sparse tables that never reported
            self.table_ready(stream, name)
        runs = self._dense_table_runs()
        flat = getattr(net, "grad_flat", None)
        bn_done = False
        if (self.coalesce and flat is not None and len(runs) == 1 and runs[0][1] == 0 and "dense" not in self._done
                and not any(k in self._done for k in runs[0][0]) and net.dense_grad.data_ptr() == flat.data_ptr()):
            # every table is dense (BASELINE configs[1]-[3]): [dense gradients | gradient tables (| moving statistics of a
            # per-rank batch-norm)] are ONE range of the flat gradient buffer and travel in ONE collective -- all of them are
            # issued here, behind the backward pass, whatever their order: every collective costs a launch and the ring's
            # latency whatever its size (world 1, CLSR_FORCE_DP: three stream hand-overs fewer at the end of the step)
            names, g0, n = runs[0]
            t0 = (net.tab_grad_flat.data_ptr() - flat.data_ptr()) // flat.element_size()
            end = flat.numel() if not self.sync_bn else t0 + g0 + n
            ds = self._dense_stream if "dense-final" in self._done else None
            if ds is not None and stream is not None and ds is not stream:
                stream.wait_stream(ds)                  # (the dense gradients became final on the weight-gradient stream)
            self._done.update(names)
            self._done.add("dense")
            bn_done = not self.sync_bn
            self._allreduce(flat[:end], dist.ReduceOp.SUM, stream, "dense+tables:" + "+".join(names))
        for names, g0, n in runs:
            if not all(k in self._done for k in names):     # (only a second _finish of the same step finds them done)
                self._done.update(names)
                self._allreduce(net.tab_grad_flat[g0:g0 + n], dist.ReduceOp.SUM, stream, "tables:" + "+".join(names))
        if "dense" not in self._done:
            self._done.add("dense")
            ds = self._dense_stream if "dense-final" in self._done else stream
            self._allreduce(net.dense_grad, dist.ReduceOp.SUM, ds, "dense")
        if self.comm is not None and self.small.numel() <= 64:
            # the 24 doubles through the peer-mapped buffers (one small launch on the compute stream, no process-group call)
            if self.trace is not None:
                self.trace.append(("collective", "small (p2p)"))
            with self._on(stream):
                self.comm.all_reduce(self.small, self.small.numel())
        else:
            self._allreduce(self.small, dist.ReduceOp.SUM, stream, "small")
        if not self.sync_bn and not bn_done:
            # keep the (non-trainable) moving statistics identical on every replica: their average
            self._allreduce(net.bn_moving, dist.ReduceOp.SUM, stream, "bn_moving")
        with self._on(stream):
            for w in self._works:
                w.wait()
            if not self.sync_bn:
                net.bn_moving.mul_(1.0 / self.world)
        self._works = []

    def _exchange(self):
        self._finish()

    def _update(self):
        self.net._apply_updates()

    ARM_AFTER_STEPS = 5     # good steps before the cross-rank waits take their production bound (csrc/p2p.hip)

    def _abort_check(self):
        """One step late, without a synchronisation: the asynchronous copy of the abort flag made after the previous step."""
        if self._abort_host is not None and float(self._abort_host[0]) != 0.0:
            self.net.check_abort()         # (an earlier step gave up on the device: raises StepAborted)

    def _abort_poll(self):
        if self._abort_host is not None:
            self._abort_host.copy_(self.net.adam_state[4:5], non_blocking=True)
        self._steps_done += 1
        if self._steps_done == self.ARM_AFTER_STEPS and self.comm is not None:
            # every rank counts the same steps: the warm-up bound (60 s: allocations, plan recording, lazy module loads
            # make the first steps drift apart by seconds) gives way to the production bound on all of them alike
            ops.query("clsr_p2p_arm_timeout", 1)

    def train_step(self, f):
        self._abort_check()
        self._backward(f)
        self._finish()
        self._update()
        self._abort_poll()

    def capture(self, f):
        """Two hipGraphs (backward | update) around the eager RCCL exchange; returns run().
        With sync_bn the BN collectives sit inside the backward phase, so that phase stays eager; the overlapped
        exchange issues collectives from inside the backward phase too, so graphs need ``overlap=False``."""
        if self.sync_bn or self.overlap:
            return lambda: self.train_step(f)
        ops.graph_begin()
        self._backward(f)
        g1 = ops.graph_end()
        ops.graph_begin()
        self._update()
        g2 = ops.graph_end()
        self._graphs = (g1, g2)

        def run():
            self._abort_check()
            ops.graph_launch(g1)
            self._exchange()
            ops.graph_launch(g2)
            self._abort_poll()

        return run

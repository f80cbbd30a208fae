"""CPU tests of the data-parallel exchange step with the gloo backend, world_size 2."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clsr_amd.dp import allgather_row_lists, allreduce_step_buffers, shard_feed, touched_rows_bound


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    dense = torch.randn(1000, generator=g)
    tabs = torch.randn(4096, generator=g)
    flags = (torch.rand(512, generator=g) < 0.1).to(torch.uint8)
    small = torch.randn(24, generator=g, dtype=torch.float64)
    ref = [t.clone() for t in (dense, tabs, flags, small)]
    allreduce_step_buffers(dist, dense, tabs, flags, small)
    # gather what the other rank had to check the reduction semantics
    other = [torch.zeros_like(t) for t in ref]
    for i, t in enumerate(ref):
        lst = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        other[i] = lst
    ok = torch.allclose(dense, sum(other[0])) and torch.allclose(tabs, sum(other[1])) \
        and torch.equal(flags, torch.maximum(other[2][0], other[2][1])) and torch.allclose(small, sum(other[3]))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_exchange_step_gloo_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_shard_feed_keeps_groups_together():
    B, G = 40, 5
    feed = {"labels": np.arange(B).reshape(B, 1), "users": np.repeat(np.arange(B // G), G)}
    a, b = shard_feed(feed, 0, 2, G), shard_feed(feed, 1, 2, G)
    assert a["labels"].shape[0] == b["labels"].shape[0] == 20
    assert set(a["users"]) & set(b["users"]) == set()
    np.testing.assert_array_equal(np.concatenate([a["labels"], b["labels"]]), feed["labels"])
    # 8 positives over 3 ranks: two whole groups each, the remainder is dropped (last partial batch of an epoch),
    # exactly like the compact layout
    parts = [shard_feed(feed, r, 3, G) for r in range(3)]
    assert [p["labels"].shape[0] for p in parts] == [10, 10, 10]
    np.testing.assert_array_equal(np.concatenate([p["labels"] for p in parts]), feed["labels"][:30])
    for bad_world, bad_g in ((9, G), (2, 3)):      # more ranks than positives; rows that are not whole groups
        try:
            shard_feed(feed, 0, bad_world, bad_g)
            assert False
        except ValueError:
            pass


def _rows_worker(rank, world, port, out):
    """Sparse row exchange, host logic only: numpy stand-ins for the pack / unpack kernels."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, C, cap = 300, 8, 64
    g = torch.Generator().manual_seed(7 + rank)
    n = int(torch.randint(10, cap, (1,), generator=g))
    touched = torch.sort(torch.randperm(V, generator=g)[:n]).values
    grad = torch.zeros(V, C)
    grad[touched] = torch.randn(n, C, generator=g)
    # pack (what clsr_flags_compact + clsr_rows_pack produce)
    count = torch.tensor([n, 0], dtype=torch.int32)
    ids = torch.zeros(cap, dtype=torch.int32)
    ids[:n] = touched.to(torch.int32)
    rows = torch.zeros(cap, C)
    rows[:n] = grad[touched]
    counts_all = torch.zeros(world, 2, dtype=torch.int32)
    ids_all = torch.zeros(world, cap, dtype=torch.int32)
    rows_all = torch.zeros(world, cap, C)
    allgather_row_lists(dist, count, ids, rows, counts_all, ids_all, rows_all)
    # unpack in rank order (what clsr_rows_unpack does)
    total = grad.clone()
    total[touched] = 0
    for r in range(world):
        k = int(counts_all[r, 0])
        total[ids_all[r, :k].long()] += rows_all[r, :k]
    dense = grad.clone()
    dist.all_reduce(dense)
    lst = [torch.zeros_like(total) for _ in range(world)]
    dist.all_gather(lst, total)
    out[rank] = bool(torch.allclose(total, dense, atol=1e-6) and torch.equal(lst[0], lst[1]))
    dist.destroy_process_group()


def test_sparse_row_exchange_gloo_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rows_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_touched_rows_bound():
    shape = (20480, 50, 5, 4096)
    assert touched_rows_bound("user_long", shape, 36915) == 4096
    assert touched_rows_bound("item", shape, 64138) == 64138          # bound exceeds the vocabulary
    assert touched_rows_bound("item", shape, 100_000_000) == 4096 * 50 + 20480


def test_shard_feed_compact_layout():
    """Compact (history-level) training feeds are sharded by positives; row-level arrays follow with G rows each."""
    n, G, T = 10, 5, 4
    feed = {"hist_group": G, "labels": np.arange(n * G).reshape(n * G, 1), "items": np.arange(n * G),
            "cates": np.arange(n * G), "users": np.arange(n), "mask": np.ones((n, T)),
            "item_history": np.arange(n * T).reshape(n, T), "item_cate_history": np.zeros((n, T)),
            "time_from_first_action": np.zeros((n, T)), "time_to_now": np.zeros((n, T))}
    parts = [shard_feed(feed, r, 3, G) for r in range(3)]      # 10 positives over 3 ranks: 3 each, 1 dropped
    for r, p in enumerate(parts):
        assert p["hist_group"] == G and p["users"].tolist() == [3 * r, 3 * r + 1, 3 * r + 2]
        assert p["items"].tolist() == list(range(15 * r, 15 * r + 15)) and p["labels"].shape == (15, 1)
        assert p["item_history"].shape == (3, T) and p["item_history"][0, 0] == 3 * r * T
    try:
        shard_feed(feed, 0, 2, G + 1)
        raise AssertionError("group mismatch accepted")
    except ValueError:
        pass


class _StubNet(object):
    """What DataParallel touches of a net, on CPU tensors."""

    def __init__(self):
        self.device = torch.device("cpu")
        shapes = dict(item=(50, 8), cate=(7, 4), user_long=(20, 8), user_short=(20, 8))
        self.tab_shape = shapes
        self.tab_goff, self.tab_foff, g, f = {}, {}, 0, 0
        for k, (V, C) in shapes.items():
            self.tab_goff[k], self.tab_foff[k] = g, f
            g += V * C
            f += (V + 15) // 16 * 16
        self.grad_flat = torch.zeros(100 + g + 8)
        self.dense_grad = self.grad_flat[:100]
        self.tab_grad_flat = self.grad_flat[100:100 + g]
        self.bn_moving = self.grad_flat[100 + g:]
        self.tab_flags_flat = torch.zeros(f, dtype=torch.uint8)
        self.tables = {k: torch.zeros(s) for k, s in shapes.items()}
        self.tab_grad = {k: self.tab_grad_flat[self.tab_goff[k]:self.tab_goff[k] + s[0] * s[1]].view(s)
                         for k, s in shapes.items()}
        self.tab_flags = {k: self.tab_flags_flat[self.tab_foff[k]:self.tab_foff[k] + s[0]] for k, s in shapes.items()}
        self.tab_m, self.tab_v = {}, {}
        self.dense = self.dense_m = self.dense_v = torch.zeros(1)
        self.adam_state = torch.zeros(4, dtype=torch.float64)
        self.stats24 = torch.zeros(24, dtype=torch.float64)
        self.last_shape = (10, 5, 5, 2)
        self.dp_world, self.dp_stats_hook, self.dp_hooks = 1, None, None
        self.updated = 0
        self.script = []

    def train_step(self, f, apply=True):
        for name, args in self.script:             # the hooks a real step would fire, in the scripted order
            getattr(self.dp_hooks, name)(None, *args)

    def _apply_updates(self):
        self.updated += 1


class _LogDist(object):
    class ReduceOp(object):
        SUM, MAX = "sum", "max"

    class _Work(object):
        def __init__(self, log, tag):
            self.log, self.tag = log, tag

        def wait(self):
            self.log.append(("wait", self.tag))

    def __init__(self):
        self.log = []

    def get_world_size(self, group=None):
        return 2

    def get_rank(self, group=None):
        return 1

    def broadcast(self, t, src=0, group=None):
        pass

    def all_reduce(self, t, op=None, group=None, async_op=False):
        self.log.append(("all_reduce", t.data_ptr(), t.numel(), op))
        return self._Work(self.log, t.data_ptr()) if async_op else None


def test_overlapped_exchange_bookkeeping_every_piece_exactly_once_and_waited_for():
    """Host logic of DataParallel: whatever subset of hooks a step fires (all of them, some, none -- a net without
    hooks), every piece of gradient state is all-reduced exactly once: the byte maps first, the adjacent dense gradient
    tables as ONE collective, the dense gradients BEHIND the tables (collectives of a group run in issue order and the
    dense gradients are final last), the 24 doubles last; the update runs only after every collective has been waited
    for."""
    from clsr_amd.dp import DataParallel

    scripts = [
        [("flags_ready", ()), ("dense_ready", ()), ("table_ready", ("cate",)), ("table_ready", ("user_long",)),
         ("table_ready", ("user_short",)), ("table_ready", ("item",))],
        [("dense_ready", ()), ("table_ready", ("cate",))],                    # flags never reported: forced before the tables
        [],                                                                   # a net without hooks
        [("table_ready", ("item",)), ("table_ready", ("item",))],             # a duplicate report is ignored
    ]
    for script in scripts:
      for coalesce in (True, False):
        net, d = _StubNet(), _LogDist()
        dp = DataParallel(net, d, sync_bn=True, sparse_tables="none")
        dp.coalesce = coalesce
        assert net.dp_hooks is dp and net.dp_world == 2
        net.script = script
        dp.train_step({})
        ar = [e for e in d.log if e[0] == "all_reduce"]
        ptrs = [e[1] for e in ar]
        last = list(net.tab_shape)[-1]                      # (the byte maps travel without the last table's padding)
        if coalesce:
            # every table dense: [dense gradients | gradient tables] of the flat buffer in ONE collective (not the moving
            # statistics behind them: sync BN)
            expect = {net.grad_flat.data_ptr(): 100 + net.tab_grad_flat.numel(), net.stats24.data_ptr(): 24,
                      net.tab_flags_flat.data_ptr(): net.tab_foff[last] + net.tab_shape[last][0]}
            order = [net.tab_flags_flat.data_ptr(), net.grad_flat.data_ptr(), net.stats24.data_ptr()]
        else:
            expect = {net.dense_grad.data_ptr(): 100, net.stats24.data_ptr(): 24,
                      net.tab_flags_flat.data_ptr(): net.tab_foff[last] + net.tab_shape[last][0],
                      net.tab_grad_flat.data_ptr(): net.tab_grad_flat.numel()}       # the four tables: one collective
            order = [net.tab_flags_flat.data_ptr(), net.tab_grad_flat.data_ptr(), net.dense_grad.data_ptr(),
                     net.stats24.data_ptr()]
        assert sorted(ptrs) == sorted(expect), script          # every piece exactly once (sync BN: no moving stats)
        assert all(expect[e[1]] == e[2] for e in ar)
        assert ptrs == order, script
        waits = [e for e in d.log if e[0] == "wait"]
        assert len(waits) == len(ar) and d.log.index(waits[0]) > d.log.index(ar[-1])
        assert net.updated == 1
    # a sparse table in the middle splits the dense tables into two runs, each one collective
    net, d = _StubNet(), _LogDist()
    dp = DataParallel(net, d, sync_bn=True, sparse_tables="none")
    dp.last_sparse = ["cate"]
    runs = dp._dense_table_runs()
    assert [r[0] for r in runs] == [["item"], ["user_long", "user_short"]]
    assert runs[0][1:] == (0, 50 * 8) and runs[1][1:] == (net.tab_goff["user_long"], 2 * 20 * 8)
    # per-rank batch-norm: the moving statistics are averaged as well
    net, d = _StubNet(), _LogDist()
    net.bn_moving += 2.0
    dp = DataParallel(net, d, sync_bn=False, sparse_tables="none", overlap=False)
    assert net.dp_hooks is None
    dp.train_step({})
    # (the whole flat buffer in one collective: dense gradients, gradient tables, moving statistics)
    assert any(e[0] == "all_reduce" and e[1] == net.grad_flat.data_ptr() and e[2] == net.grad_flat.numel() for e in d.log)
    assert not any(e[0] == "all_reduce" and e[1] == net.bn_moving.data_ptr() for e in d.log)
    assert torch.allclose(net.bn_moving, torch.ones(8))       # (identity all-reduce) * 1 / world
    net, d = _StubNet(), _LogDist()
    net.bn_moving += 2.0
    dp = DataParallel(net, d, sync_bn=False, sparse_tables="none", overlap=False)
    dp.coalesce = False
    dp.train_step({})
    assert any(e[0] == "all_reduce" and e[1] == net.bn_moving.data_ptr() for e in d.log)
    assert torch.allclose(net.bn_moving, torch.ones(8))

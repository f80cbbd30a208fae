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

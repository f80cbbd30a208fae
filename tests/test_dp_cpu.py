"""CPU tests of the data-parallel exchange step with the gloo backend, world_size 2."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clsr_amd.dp import allreduce_step_buffers, shard_feed


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
    try:
        shard_feed(feed, 0, 3, G)
        assert False
    except ValueError:
        pass

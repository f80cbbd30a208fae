"""The small-message all-reduce of csrc/p2p.hip (the synchronised batch-norm statistics of
tf.layers.batch_normalization, models/base_model.py:673-679, summed across the data-parallel ranks): two processes that
share the one GPU of the test box map each other's exchange buffer through hipIpc handles and all-reduce through the
C ABI -- every rank must hold the fp64 sum, bit-identical on both, for a long sequence of calls of different lengths."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clsr_amd import p2p

    torch.cuda.set_device(0)
    comm = p2p.from_process_group(rank, world)
    g = torch.Generator().manual_seed(5)          # the same stream of values on every rank: rank r uses row r
    res, exp = [], []
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for it in range(200):
            n = (1, 2, 80, 160, 200, 256)[it % 6]
            vals = torch.randn(world, n, generator=g, dtype=torch.float64)
            x = vals[rank].clone().cuda()
            comm.all_reduce(x)
            res.append(x)
            e = torch.zeros(n, dtype=torch.float64)
            for r in range(world):                # rank order: the kernel's summation order
                e = e + vals[r]
            exp.append(e)
        s.synchronize()
    ok = all(torch.equal(r.cpu(), e) for r, e in zip(res, exp))
    err = comm.error()
    dist.barrier()
    comm.close()
    if rank == 0:
        out["ok"], out["err"] = bool(ok), int(err)
    else:
        out["ok1"] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_small_allreduce_between_processes_on_one_gpu(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_worker, args=(world, _port(), out), nprocs=world, join=True)
    assert out["ok"] and out["err"] == 0
    if world > 1:
        assert out["ok1"]


def _dp_worker(rank, world, port, hp, dims, feed, sd, out, use_p2p):
    import torch.distributed as dist

    from clsr_amd.dp import DataParallel, shard_feed
    from clsr_amd.net import CLSRNet
    from test_dp_gpu import _init

    d, dev = _init("staged", rank, world, port)
    net = CLSRNet(hp, dims, device=dev, seed=rank)
    if rank == 0:
        net.load_state_dict(sd)
    dp = DataParallel(net, d, sync_bn=True, sparse_tables="none", p2p_stats=use_p2p)
    f = dp.prepare(net.upload(shard_feed(feed, rank, world, hp.train_num_ngs + 1), True))
    for _ in range(3):                 # (the third step replays the recorded launch plan)
        dp.train_step(f)
    torch.cuda.synchronize()
    if rank == 0:
        out["transport"] = dp.stats_transport
        out["state"] = {k: v.cpu().numpy() for k, v in net.state_dict().items()}
        out["err"] = dp.comm.error() if dp.comm is not None else 0
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_through_the_p2p_communicator_equals_the_process_group(golden_dir, golden_hparams):
    """Two ranks on one GPU, three data-parallel steps with synchronised batch-norm: the statistics summed by
    clsr_allreduce_small give the same variables as the statistics summed by the process group."""
    import pickle

    import numpy as np
    import torch.multiprocessing as mp

    from oracle import clsr_oracle as O

    hp = golden_hparams
    dims = dict(Vu=len(pickle.load(open(hp.user_vocab, "rb"))), Vi=len(pickle.load(open(hp.item_vocab, "rb"))),
                Vc=len(pickle.load(open(hp.cate_vocab, "rb"))))
    g = np.load(os.path.join(golden_dir, "iterator_train_sa.npz"))
    feed = {k[3:]: g[k] for k in g.files if k.startswith("b0_")}
    params = O.init_params(dims, hp, seed=3, scale_dense=8.0)
    sd = dict(params)
    sd.update(O.init_bn_state(params))
    ctx = mp.get_context("spawn")
    states = []
    for use_p2p in (True, False):
        out = ctx.Manager().dict()
        mp.spawn(_dp_worker, args=(2, _port(), hp, dims, feed, sd, out, use_p2p), nprocs=2, join=True)
        assert out["err"] == 0
        assert out["transport"] == ("p2p" if use_p2p else "torch.distributed"), out["transport"]
        states.append(out["state"])
    a, b = states
    for k in a:
        # (fp64 sums of two addends are the same in both transports; Adam's noise on analytically-zero gradients aside)
        np.testing.assert_allclose(a[k], b[k], rtol=2e-4, atol=2e-6, err_msg=k)

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


def _dp_worker(rank, world, port, hp, dims, feed, sd, out, use_p2p, fused_heads=True):
    import torch.distributed as dist

    from clsr_amd.dp import DataParallel, shard_feed
    from clsr_amd.net import CLSRNet
    from test_dp_gpu import _init

    os.environ["CLSR_HEADS_COMM_SHARED"] = "1"      # (two ranks on ONE device: fine at this batch, 2 x 8 workgroups)
    d, dev = _init("staged", rank, world, port)
    net = CLSRNet(hp, dims, device=dev, seed=rank)
    net.heads_fused = bool(fused_heads)
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
        out["losses"] = net.read_losses()       # (raises if a grid barrier of the fused heads timed out)
        out["fused"] = bool(net._heads_fused_ok(f["B"], hp.train_num_ngs + 1))
    dist.barrier()
    dp.close()                      # (unmaps the peers' exchange buffers, frees the own ones)
    assert dp.comm is None and net.dp_comm is None and net.heads_comm is None
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
    # p2p statistics + the fused heads launches (their group sums pushed to the peer from inside the launch), p2p statistics
    # with the launch chain, everything through the process group
    for use_p2p, fused in ((True, True), (True, False), (False, False)):
        out = ctx.Manager().dict()
        mp.spawn(_dp_worker, args=(2, _port(), hp, dims, feed, sd, out, use_p2p, fused), nprocs=2, join=True)
        assert out["err"] == 0
        want = "p2p + fused heads" if (use_p2p and fused) else "p2p" if use_p2p else "torch.distributed"
        assert out["transport"] == want, out["transport"]
        assert out["fused"] == (use_p2p and fused)
        states.append((out["state"], out["losses"]))
    (a, la), (b, lb), (c, lc) = states
    for k in la:
        assert abs(la[k] - lc[k]) <= 2e-6 * max(1.0, abs(lc[k])) and abs(lb[k] - lc[k]) <= 2e-6 * max(1.0, abs(lc[k])), k
    for k in c:
        # (fp64 sums of two addends are the same in both transports; Adam's noise on analytically-zero gradients aside)
        np.testing.assert_allclose(b[k], c[k], rtol=2e-4, atol=2e-6, err_msg=k)
        # fused heads vs the chain: different summation orders upstream, and Adam normalises every gradient -- elements whose
        # gradient is small against the noise move by a visible fraction of a step in either path (the gradients themselves
        # are compared to 2e-4 in tests/test_heads_fused_gpu.py).  What THIS test is about are the cross-rank sums: the
        # batch-norm moving statistics are those sums (a rank using only its own rows would be off by tens of percent), so
        # they are compared tightly, the trained variables on the scale of the three steps.
        if "moving_variance" in k:
            np.testing.assert_allclose(a[k], c[k], rtol=2e-4, atol=1e-6, err_msg="fused heads: " + k)
        elif "moving_mean" in k:       # (shifted by the noise of the bias in front of the batch-norm)
            np.testing.assert_allclose(a[k], c[k], rtol=2e-4, atol=0.2 * 3 * float(hp.learning_rate), err_msg="fused heads: " + k)
        else:
            # (biases in front of a batch-norm or of a softmax have an analytically ZERO gradient: pure noise, a full step
            # of either sign per update)
            noise = "b_nn_layer" in k or ("b_nn_output" in k and "fcn_alpha" not in k)
            np.testing.assert_allclose(a[k], c[k], rtol=5e-4, atol=(2.0 if noise else 0.1) * 3 * float(hp.learning_rate),
                                       err_msg="fused heads: " + k)


def _abort_worker(out):
    """ONE process that plays rank 0 of a two-rank communicator whose peer never pushes: the all-reduce must give up after
    CLSR_P2P_TIMEOUT_S, raise the sticky error word AND the abort flag, return NaN instead of a partial sum, and every
    optimiser kernel must then leave parameters, moments and the Adam clock alone while the flag is up."""
    import ctypes

    os.environ["CLSR_P2P_TIMEOUT_S"] = "0.05"        # (read once per process: set before the first call)
    from clsr_amd import _lib, ops

    torch.cuda.set_device(0)
    lib = _lib.load()
    bufs = (ctypes.c_void_p * 2)()
    for r in range(2):
        b = ctypes.c_void_p()
        _lib.check(lib.clsr_comm_alloc(ctypes.byref(b)), "clsr_comm_alloc")
        bufs[r] = b
    comm = ctypes.c_void_p()
    _lib.check(lib.clsr_comm_create(0, 2, bufs, ctypes.byref(comm)), "clsr_comm_create")
    st = torch.tensor([0.0, 1.0, 1.0, 0.0, 0.0], dtype=torch.float64, device="cuda")      # Adam state: [4] = abort flag
    _lib.check(lib.clsr_comm_set_abort(comm, ctypes.c_void_p(st[4:].data_ptr())), "clsr_comm_set_abort")
    x = torch.arange(1, 9, dtype=torch.float64, device="cuda")
    ops.call("clsr_allreduce_small", comm.value, x, 8)
    torch.cuda.synchronize()
    out["nan"] = bool(torch.isnan(x).all())
    out["err"] = int(lib.clsr_comm_error(comm))
    out["flag"] = float(st[4])
    # an optimiser kernel under the raised flag: nothing moves
    n = 1024
    p, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    seg = torch.zeros(n, dtype=torch.int32, device="cuda")
    sumsq = torch.ones(1, dtype=torch.float64, device="cuda")
    ops.call("clsr_adam_tick", st, 1e-3, 0.9, 0.999)
    p0 = p.clone()
    ops.call("clsr_dense_adam", p, g, m, v, seg, sumsq, 0.0, st, 0.9, 0.999, 1e-8, n)
    torch.cuda.synchronize()
    out["frozen"] = bool(torch.equal(p, p0) and float(m.abs().max()) == 0.0 and float(v.abs().max()) == 0.0
                         and float(st[0]) == 0.0 and float(st[1]) == 1.0)      # (the clock too)
    st[4] = 0.0                                       # flag down (CLSRNet: load_state_dict of a checkpoint): updates resume
    ops.call("clsr_adam_tick", st, 1e-3, 0.9, 0.999)
    ops.call("clsr_dense_adam", p, g, m, v, seg, sumsq, 0.0, st, 0.9, 0.999, 1e-8, n)
    torch.cuda.synchronize()
    out["resumed"] = bool(not torch.equal(p, p0) and float(m.abs().max()) > 0.0)
    lib.clsr_comm_destroy(comm)
    for r in range(2):
        lib.clsr_comm_free(bufs[r])


def test_all_reduce_that_gives_up_aborts_the_step():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as man:
        out = man.dict()
        p = ctx.Process(target=_abort_worker, args=(out,))
        p.start()
        p.join(120)
        assert p.exitcode == 0
        assert out["nan"], "a timed-out all-reduce must not return a partial sum"
        assert out["err"] == 1 and out["flag"] != 0.0
        assert out["frozen"] and out["resumed"]

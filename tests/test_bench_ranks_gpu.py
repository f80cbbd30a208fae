"""bench.py's per-rank program with WORLD_SIZE = 2 -- the code the driver's N > 1 scaling runs execute: process-group
set-up, the all-reduce-of-ones proof, DataParallel with synchronised batch-norm statistics (peer-to-peer all-reduces +
the fused heads pushing their statistics across ranks), barriers around the timed region, per-rank times gathered, ONE
JSON line from rank 0.  The box has one GPU: both ranks share it and the process-group collectives are staged through the
host (CLSR_BENCH_TRANSPORT=staged, clsr_amd.dp.HostStagedDist); the peer-to-peer communicators run as they do between
GPUs (hipIpc-mapped exchange buffers, system-scope atomics).  Small batch (the "plumbing" configuration)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("local_bn", [False, True])
def test_two_rank_bench_program_prints_one_line(local_bn):
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CLSR_BENCH_TRANSPORT="staged", CLSR_BENCH_DEVICE="0", CLSR_HEADS_COMM_SHARED="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", CLSR_P2P_TIMEOUT_S="20")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "3",
               "--config", "plumbing", "--no-extra", "--no-cpu-baseline", "--no-catalogue"] + (["--local-bn"] if local_bn else [])
        procs.append(subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, se[-3000:])
    lines = [[ln for ln in so.splitlines() if ln.startswith('{"metric"')] for so, _ in outs]
    assert len(lines[0]) == 1 and not lines[1], "exactly ONE JSON line, from rank 0"
    d = json.loads(lines[0][0])
    assert d["n_gpus"] == world and d["steps"] == 4 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["config"]["rccl_ranks"] == world
    per_rank = d["config"]["ms_per_step_by_rank"]
    assert len(per_rank) == world and all(t > 0 for t in per_rank)
    assert abs(d["ms_per_step"] - max(per_rank)) < 1e-3            # MAX over the ranks
    assert abs(d["value"] - world * 64 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]   # whole-job interactions/s
    assert d["config"]["batch_norm"] == ("per-rank" if local_bn else "sync")
    if not local_bn:
        assert "p2p" in d["config"]["sync_bn_statistics_through"], d["config"]["sync_bn_statistics_through"]
    assert d["loss"] == d["loss"] and d["loss"] > 0                 # finite: no aborted step (CLSRNet.check_abort)

"""The N>1 path on CPU: world_size-2 gloo.  Envs shard by GLOBAL env id with no data-path
collective; the only collective is the SUM all-reduce of the 4-double metric vector."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H
from sorrel_amd import distributed as D
from sorrel_amd.spec import treasurehunt_spec


def test_shard_range_partitions_exactly():
    for n, w in ((65536, 8), (524288, 8), (10, 3), (7, 8), (4096, 2)):
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, E, T, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    first, count = D.shard_range(E, rank, world)
    ws = treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.02, seed=4)
    co = H.COracle(ws, count, first_env_id=first, threads=1)      # stands in for the rank's GPU engine
    co.reset(0)
    for t in range(1, T + 1):
        co.step(0, t, random_actions=True)
    m = torch.from_numpy(co.metrics())
    D.all_reduce_metrics(m)
    if rank == 0:
        out.put((m.numpy().copy(), co.total.copy(), co.grid.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process():
    E, T = 64, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, E, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    metrics, total0, grid0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ws = treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.02, seed=4)
    whole = H.COracle(ws, E, first_env_id=0, threads=1)
    whole.reset(0)
    for t in range(1, T + 1):
        whole.step(0, t, random_actions=True)
    ref = whole.metrics()
    assert np.array_equal(metrics, ref)                         # integer rewards: exact in any order
    assert np.array_equal(total0, whole.total[: E // 2])        # rank 0's shard == first half of the batch
    assert np.array_equal(grid0, whole.grid[: E // 2])


def test_bench_starts_its_own_ranks_before_touching_a_gpu():
    """``python bench.py --gpus 2`` without a launcher's WORLD_SIZE (round 6): the parent starts ``torch.distributed.run`` with two ranks as a CHILD process and relays
    its exit code -- here, without a GPU, every rank stops at "no HIP device" (rc 2), the launcher reports the failure and the parent hands it on.  The parent itself never
    imports torch (nothing in it can touch a GPU): checked on the module."""
    import subprocess
    import sys

    if torch.cuda.is_available():           # (on a GPU box the two-rank run is the GPU tests' business: tests/test_gpu_bench_line.py)
        import pytest

        pytest.skip("a GPU box: covered by the -m gpu tests")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode != 0
    assert out.stderr.count("no HIP device") >= 2, out.stderr[-1500:]          # both ranks were started, and each said why it stopped
    src = open(os.path.join(root, "bench.py")).read()
    head = src[src.index("def launch_ranks"):src.index("def series_stats")]
    import re
    assert not re.search(r"^\s*(import|from)\s+torch", head, re.M) and "os.exec" not in head and "execv" not in head

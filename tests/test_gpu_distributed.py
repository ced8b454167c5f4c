"""RCCL on the hardware there is: a one-GPU box can still load RCCL, build a communicator and run the product's ONE collective -- the
SUM all-reduce of the metric vector on the device -- and the device-side barrier bench.py brackets its timed region with.
Each case runs in a child process (a communicator that hangs must not take the suite with it); the child is ``python tests/_rccl_child.py``.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(mode, timeout=420):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SGW_BENCH_REHEARSAL"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_child.py"), mode], capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_metric_all_reduce_and_barrier_over_rccl_world_size_1(built):
    """backend "nccl" (= RCCL), world_size 1, device_id cuda:0: ``distributed.rollout_metrics`` runs the real ``all_reduce`` of the
    float64[4] on the device, ``dist.barrier(device_ids=[0])`` the barrier of bench.py; the numbers equal the C oracle's
    ``sgo_reduce_metrics`` over the same rollout.  The first time RCCL itself is loaded by this code."""
    r = _child("metrics")
    assert r["backend"] == "nccl" and r["world_size"] == 1 and r["rccl_loaded"] is True, r
    assert r["all_reduced_on"] == "cuda:0"
    assert r["metrics"] == r["oracle_metrics"], r
    assert r["reduced_twice"] == r["metrics"]                 # SUM over one rank is the identity: a second all-reduce changes nothing
    assert r["sum_total_reward"] == r["oracle_metrics"][0] and r["envs"] == 4096.0
    assert r["barrier_ok"] is True and r["destroyed"] is True


@pytest.mark.gpu
def test_bench_line_over_rccl_world_size_1(built):
    """bench.py's own N > 1 control path (init_process_group("nccl", device_id=...), barrier(device_ids=...), the three reductions) under
    ``torch.distributed.run --nproc-per-node 1``: WORLD_SIZE is set, so the script takes the distributed branch with RCCL -- on one rank."""
    r = _child("bench")
    assert r["rc"] == 0, r
    line = r["line"]
    assert line["n_gpus"] == 1 and line["config"]["process_group_world_size"] == 1 and line["rollout"]["status"] == 0
    assert line["config"]["collectives_backend"] == "nccl"
    assert line["rollout"]["sum_total_reward"] == r["plain_sum_total_reward"]     # the same rollout without a process group

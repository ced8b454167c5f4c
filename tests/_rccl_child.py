"""Child process of tests/test_gpu_distributed.py (not collected by pytest): one rank, backend "nccl" = RCCL, on cuda:0."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def metrics():
    import numpy as np
    import torch
    import torch.distributed as dist

    from sorrel_amd import distributed as D
    from sorrel_amd.engine import GridEngine
    from sorrel_amd.spec import treasurehunt_spec
    from tests import helpers as H

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    E, T = 4096, 12
    ws = treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.02, seed=4)
    eng = GridEngine(ws, E, device=dev, first_env_id=0)
    eng.reset(epoch=0)
    for _ in range(T):
        eng.step(random_actions=True)
    m = D.rollout_metrics(eng)                                   # engine reduction + dist.all_reduce(SUM) of the f64[4] ON THE DEVICE
    raw = eng.reduce_metrics().clone()
    where = str(raw.device)
    D.all_reduce_metrics(raw)
    once = [float(x) for x in raw.tolist()]
    D.all_reduce_metrics(raw)
    twice = [float(x) for x in raw.tolist()]
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize(dev)
    co = H.COracle(ws, E, first_env_id=0, threads=0)
    co.reset(0)
    for t in range(1, T + 1):
        co.step(0, t, random_actions=True)
    assert np.array_equal(eng.total_reward.cpu().numpy(), co.total)
    with open("/proc/self/maps") as fh:
        rccl = any("librccl" in ln for ln in fh)
    backend, world = dist.get_backend(), dist.get_world_size()
    dist.destroy_process_group()
    print(json.dumps({"backend": backend, "world_size": world, "rccl_loaded": rccl, "all_reduced_on": where, "metrics": once,
                      "reduced_twice": twice, "oracle_metrics": [float(x) for x in co.metrics()], "sum_total_reward": m["sum_total_reward"],
                      "envs": m["envs"], "barrier_ok": True, "destroyed": not dist.is_initialized()}), flush=True)


def bench():
    """No GPU call in THIS process: it only starts bench.py twice (under the launcher with one rank, and plain)."""
    common = ["--gpus", "1", "--envs", "4096", "--steps", "8", "--warmup", "2", "--prewarm-steps", "0", "--no-cpu-baseline", "--no-series",
              "--no-side-configs", "--no-self-check"]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + common
    a = subprocess.run(launcher, capture_output=True, text=True, timeout=300, cwd=ROOT)
    if a.returncode != 0:
        print(json.dumps({"rc": a.returncode, "stderr": a.stderr[-3000:]}), flush=True)
        return
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert b.returncode == 0, b.stderr[-2000:]
    line = json.loads([ln for ln in a.stdout.splitlines() if ln.startswith("{")][-1])
    plain = json.loads([ln for ln in b.stdout.splitlines() if ln.startswith("{")][-1])
    print(json.dumps({"rc": 0, "line": line, "plain_sum_total_reward": plain["rollout"]["sum_total_reward"]}), flush=True)


if __name__ == "__main__":
    {"metrics": metrics, "bench": bench}[sys.argv[1]]()

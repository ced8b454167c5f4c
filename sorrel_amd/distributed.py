"""Multi-GPU: envs shard embarrassingly, one process per GPU; the only collective is one
all-reduce (RCCL over xGMI = backend "nccl" on ROCm) of a 4-double metric vector at the
end of a rollout.  RNG is keyed by the GLOBAL env id, so any sharding of
the same global batch is bit-identical."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(global_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of global env ids owned by ``rank``: (first_env_id, count)."""
    base, rem = divmod(global_envs, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def init_from_env(backend: str = None):
    """Join the process group torchrun describes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            # one process per GPU: bind this rank to ITS device before the group exists, or every rank's tensors
            # (and RCCL's communicator) land on cuda:0
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def all_reduce_metrics(metrics: torch.Tensor) -> torch.Tensor:
    """SUM-all-reduce the metric vector ``[sum(total_reward), sum(total_reward^2), envs, 0]``.
    Rewards of the canonical worlds are integers, so the float64 sums are exact in any order."""
    if dist.is_available() and dist.is_initialized():      # (a group of ONE rank still runs the collective: RCCL on a one-GPU box)
        if dist.get_backend() == "gloo" and metrics.is_cuda:   # gloo reduces host tensors (CPU rehearsals of the N > 1 path)
            host = metrics.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            metrics.copy_(host)
        else:
            dist.all_reduce(metrics, op=dist.ReduceOp.SUM)
    return metrics


def rollout_metrics(engine, all_reduce: bool = True) -> dict:
    m = engine.reduce_metrics().clone()
    if all_reduce:
        m = all_reduce_metrics(m)
    s, s2, n = (float(v) for v in m[:3].tolist())
    mean = s / n if n else 0.0
    return {"sum_total_reward": s, "sum_sq_total_reward": s2, "envs": n, "mean_total_reward": mean,
            "var_total_reward": max(s2 / n - mean * mean, 0.0) if n else 0.0}

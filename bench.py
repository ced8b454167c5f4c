#!/usr/bin/env python3
"""bench.py -- agent-steps/s of the batched take_turn hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W        (N=1: run directly)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one ``Environment.take_turn`` over the whole env batch (entity sweep,
A sequential agent transitions with egocentric observations, rewards, moves) with
random actions drawn on device.  Workload at N=1 = BASELINE.json configs[2]
(headline): 32x32 grid, 8 agents, 7x7 window, 65 536 envs; for N>1 each rank owns
65 536 envs of a global batch of N*65 536 (weak scaling; N=8 is configs[3]).  Envs
shard with no data-path collective; the only collective is one all-reduce (RCCL)
of the 4-double metric vector at the end of the rollout.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_COPY_GBS = 6290.0   # measured float4-copy ceiling (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)

CONFIGS = {
    # name: (H, W, agents, radius, envs per GPU, spawn_prob, dense_prob)
    "c2": (16, 16, 4, 2, 4096, 0.005, 0.0),
    "c3": (32, 32, 8, 3, 65536, 0.005, 0.0),
    "c5": (128, 128, 64, 5, 2048, 0.05, 0.25),
}


def cpu_baseline(spec, seconds_target: float = 12.0):
    """The C oracle ("port") timed on this host's cores on a bounded sample of the
    same workload.  Only the checker is used here, never as the thing measured above."""
    import ctypes as C

    import numpy as np

    import __graft_entry__ as g

    lib = C.CDLL(g.build_oracle())
    lib.sgo_threads.argtypes = [C.c_int]
    cores = int(lib.sgo_threads(0))
    A = spec.num_agents
    E = 4096 if spec.grid_bytes_per_env() <= 4096 else 256
    cfg = spec.to_config(E, 0)
    grid = np.zeros((E, spec.layers, spec.height, spec.width), np.uint8)
    pos = np.zeros((E, A, 2), np.uint8)
    act = np.zeros((E, A), np.uint8)
    obs = np.zeros((E,) + spec.obs_shape, np.float32)
    rew = np.zeros((E, A), np.float32)
    tot = np.zeros((E,), np.float64)

    def p(a):
        return a.ctypes.data_as(C.c_void_p)

    lib.sgo_reset(C.byref(cfg), p(grid), p(pos), p(tot), C.c_uint32(0), C.c_int(0), C.c_void_p(0))

    def run(t0, n, threads=0):
        for t in range(t0, t0 + n):
            lib.sgo_step(C.byref(cfg), p(grid), p(pos), p(act), p(obs), p(rew), p(tot), C.c_uint32(0), C.c_uint32(t),
                         C.c_int32(0), C.c_int32(A), C.c_uint32(1 | 2), C.c_int(threads), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0))

    run(1, 2)                                   # warm-up + page-in
    t = time.perf_counter()
    run(3, 3)
    per_turn = (time.perf_counter() - t) / 3
    turns = max(3, min(2000, int(seconds_target / max(per_turn, 1e-6))))
    t = time.perf_counter()
    run(6, turns)
    dt = time.perf_counter() - t
    # the same code on ONE thread (total, per core, and the single-core figure), a ~3 s sample
    t1 = time.perf_counter()
    run(6 + turns, 1, threads=1)
    per_turn1 = time.perf_counter() - t1
    turns1 = max(1, min(50, int(3.0 / max(per_turn1, 1e-6))))
    t1 = time.perf_counter()
    run(7 + turns, turns1, threads=1)
    dt1 = time.perf_counter() - t1
    value = E * A * turns / dt
    return {
        "value": value, "unit": "agent-steps/s", "cores": cores, "kind": "port",
        "per_core": value / cores, "single_core_value": E * A * turns1 / dt1,
        "sample": f"oracle/gridstep_oracle.c (OpenMP, {cores} threads), {E} envs x {turns} turns of the same "
                  f"{spec.height}x{spec.width}x{A}-agent workload, {dt:.1f} s; single thread: {turns1} turns, {dt1:.1f} s",
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-obs", action="store_true", help="diagnostic: skip observation stores (NOT a valid bench line)")
    ap.add_argument("--diag-agents", type=int, default=-1, help="diagnostic: step only the first N agents (NOT valid)")
    ap.add_argument("--obs-dtype", default="f32", choices=["f32", "u8"],
                    help="u8 = compact one-hot counts (an extra, reported separately; the contract format is f32)")
    ap.add_argument("--no-sweep", action="store_true", help="diagnostic: skip the entity sweep (NOT a valid bench line)")
    args = ap.parse_args()

    # the C-ABI library travels with the tree; if it is missing or older than its source (a fresh checkout on a box
    # with hipcc) build it once -- ranks of one node take turns on a lock file, the first one builds
    import fcntl

    import __graft_entry__ as graft

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            graft.build_hip()
        except Exception as exc:   # no compiler on this box: an existing library is still the product under test
            if not os.path.isfile(graft.HIP_LIB):
                raise
            print(f"bench.py: could not rebuild libsgw.so ({exc}); using the existing library", file=sys.stderr)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)

    import torch
    import torch.distributed as dist

    from sorrel_amd.engine import GridEngine
    from sorrel_amd.spec import treasurehunt_spec

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
            return 2
    if not torch.cuda.is_available():
        print("bench.py: no HIP device; the hot path has no CPU fallback", file=sys.stderr)
        return 2
    # SGW_BENCH_REHEARSAL=1 (1-GPU box only): every rank shares cuda:0 and the collectives run over gloo,
    # to rehearse the N>1 control path where a second GPU is not available.  Never set by the driver.
    rehearsal = os.environ.get("SGW_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    H, W, A, r, E_cfg, p_spawn, p_dense = CONFIGS[args.config]
    E = args.envs or E_cfg
    spec = treasurehunt_spec(H, W, A, r, spawn_prob=p_spawn, seed=0, dense_prob=p_dense)
    obs_dtype = torch.float32 if args.obs_dtype == "f32" else torch.uint8
    eng = GridEngine(spec, E, device=dev, first_env_id=rank * E, obs_dtype=obs_dtype)   # global env ids: re-sharding is bit-exact
    eng.reset(epoch=0)

    def barrier():
        if world > 1:
            if rehearsal:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev_index])
        torch.cuda.synchronize(dev)

    write_obs = not args.no_obs
    sweep = not args.no_sweep
    if args.diag_agents >= 0:
        _orig_step = eng.step
        eng.step = lambda *a, **k: _orig_step(*a, agent_end=args.diag_agents, **k)
    for _ in range(args.warmup):
        eng.step(random_actions=True, write_obs=write_obs, sweep=sweep)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                   # on the stream sgw_step launches on
    for _ in range(args.steps):
        eng.step(random_actions=True, write_obs=write_obs, sweep=sweep)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events over the timed region, per launch

    # end-of-rollout metrics: on-device reduction + the one collective
    metrics = eng.reduce_metrics().clone()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        if rehearsal:     # gloo reduces host tensors
            metrics, tmax = metrics.cpu(), tmax.cpu()
        dist.all_reduce(metrics, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    status = eng.status()

    if rank == 0:
        total_envs = E * world
        value = total_envs * A * args.steps / dt
        alg_bytes = spec.algorithmic_bytes_per_env_step() * E          # per launch (one rank's kernel)
        if not write_obs:
            alg_bytes -= E * A * spec.num_channels * spec.window ** 2 * 4
        elif args.obs_dtype == "u8":
            alg_bytes -= E * A * spec.num_channels * spec.window ** 2 * 3      # C*V*V*1 instead of *4
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), collected by
        # tools/profile_gpu.sh on this same command and committed under profiles/; null when no matching pass exists
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")) as fh:
                tj = json.load(fh)
            if tj.get("envs") == E and write_obs and sweep and args.obs_dtype == "f32":
                traffic = tj["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "agent-steps/sec" if args.obs_dtype == "f32" else "agent-steps/sec (compact uint8 observations; NOT the contract metric)",
            "value": value, "unit": "agent-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {H}x{W} grid x {spec.layers} layers, {A} agents, {spec.window}x{spec.window} window, "
                            f"{E} envs/GPU x {world} GPU = {total_envs} envs, treasurehunt rules, random actions, {args.obs_dtype} one-hot obs",
                "envs_per_gpu": E, "global_envs": total_envs, "agents": A, "grid": [H, W, spec.layers],
                "window": spec.window, "channels": spec.num_channels, "spawn_prob": p_spawn, "dense_prob": p_dense,
                "sharding": f"env-batch x{world}, no data-path collective; one 32-byte all-reduce at end of rollout",
                "obs_written": write_obs, "sweep": sweep,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "frac_of_copy_ceiling": achieved / HBM_COPY_GBS,   # both denominators
                "kernel": "step_fast<...> (sgw_step)" if spec.grid_bytes_per_env() <= 4096 else "step_big<...> (sgw_step)", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_per_agent_step": spec.algorithmic_bytes_per_env_step() / A,
            },
            "rollout": {"sum_total_reward": float(metrics[0].item()), "envs": float(metrics[2].item()), "status": status},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

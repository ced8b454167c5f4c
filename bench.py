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

Order of a run: build check (before anything touches the GPU) -> reset -> P untimed
pre-warm steps (``--prewarm-steps``, reported, and each of them timed with its own pair of
HIP events: ``roofline.prewarm_series`` shows the cold start -- a fresh process runs
launches ~10-100 slower than the settled kernel and the driver's default ``--steps 20
--warmup 5`` is 3 ms of work) -> R more untimed steps (``--rewarm-steps``, reported: reading the
pre-warm pass's 1 500 timers leaves the chip idle for a few ms, and an idle of 3 / 10 ms costs the
next 20 launches 5 / 15-20 %, tools/idle_probe.py) flowing without a host-side gap into the
W untimed warm-up steps -> barrier + synchronize ->
EXACTLY K timed steps (``value`` / ``ms_per_step`` / ``roofline.kernel_ms`` come from here;
nothing but the K launches and two HIP events is in the region) -> barrier + synchronize ->
the same K steps once more with a pair of HIP events around EVERY launch
(``roofline.series``) -> the same workload through ``sgw_rollout`` (``fused_rollout``) ->
end-of-rollout metric all-reduce -> (rank 0, N=1) the other 1-GPU BASELINE shapes, briefly
(``configs``: config 2, config 5's per-GPU share, config 3 at 524 288 envs) -> the CPU
baseline -> the envs the CPU baseline played, replayed on the GPU and compared
(``rollout.checked_vs_oracle``).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_COPY_GBS = 6290.0   # measured float4-copy ceiling (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
CACHE_RESIDENT_GRID_BYTES = 288 << 20   # = kCacheResidentGrid in csrc/step_fast.h: grids of a batch up to this size are re-read
                                        # from the 256 MiB Infinity Cache + 32 MiB of L2 on the next turn, not from HBM

CONFIGS = {
    # name: (H, W, agents, radius, envs per GPU, spawn_prob, dense_prob)
    "c2": (16, 16, 4, 2, 4096, 0.005, 0.0),
    "c3": (32, 32, 8, 3, 65536, 0.005, 0.0),
    "c5": (128, 128, 64, 5, 2048, 0.05, 0.25),
    # diagnostic shapes (not BASELINE configs): the reference's Treasurehunt example default and a ragged small world
    "th21": (21, 21, 2, 2, 65536, 0.005, 0.0),
    "th10": (10, 10, 2, 2, 65536, 0.005, 0.0),
}

# The true reference (Python, one core, Xeon 2.1 GHz, build container) on one env of each shape: BASELINE.md section 2,
# code timed = sorrel/environment.py:81-93 through the oracle loader.  It cannot travel to the GPU box.
REFERENCE_PYTHON_AGENT_STEPS_PER_S = {"c2": 1701.0, "c3": 641.0, "c5": 44.0}


def host_cores() -> int:
    """CPU cores this process may really use: the scheduler affinity, cut down to the cgroup's CPU quota (a GPU box
    shows 128 logical CPUs but grants a 1-GPU job a share of them; 128 OpenMP threads on a 16-CPU quota is what made
    last round's 'parallel' baseline 4 % efficient)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    period = float(fh.read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(math.ceil(float(quota) / period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(spec, config_name: str, seconds_target: float = 12.0):
    """The C oracle ("port": oracle/gridstep_oracle.c, the reference's step loop restated, parity-pinned) timed on this
    host's cores on a bounded sample of the same workload: envs in parallel, each thread playing its own block of envs
    through all the turns (persistent threads, one fork/join per rollout).  Only the checker is used here, never as
    the thing measured above.  Loads the library built at the top of main(); compiles nothing."""
    import ctypes as C

    import numpy as np

    import __graft_entry__ as g

    lib = C.CDLL(g.ORACLE_LIB)
    cores = host_cores()
    A = spec.num_agents
    small = spec.grid_bytes_per_env() <= 4096

    def make(E):
        cfg = spec.to_config(E, 0)
        arr = dict(grid=np.zeros((E, spec.layers, spec.height, spec.width), np.uint8), pos=np.zeros((E, A, 2), np.uint8),
                   act=np.zeros((E, A), np.uint8), obs=np.zeros((E,) + spec.obs_shape, np.float32),
                   rew=np.zeros((E, A), np.float32), tot=np.zeros((E,), np.float64))
        return cfg, arr

    def p(a):
        return a.ctypes.data_as(C.c_void_p)

    def rollout(cfg, arr, t0, turns, threads):
        t = time.perf_counter()
        lib.sgo_rollout(C.byref(cfg), p(arr["grid"]), p(arr["pos"]), p(arr["act"]), p(arr["obs"]), p(arr["rew"]), p(arr["tot"]),
                        C.c_uint32(0), C.c_uint32(t0), C.c_uint32(turns), C.c_int(threads), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0))
        return time.perf_counter() - t

    def measure(E, threads, seconds):
        cfg, arr = make(E)
        lib.sgo_reset(C.byref(cfg), p(arr["grid"]), p(arr["pos"]), p(arr["tot"]), C.c_uint32(0), C.c_int(threads), C.c_void_p(0))
        rollout(cfg, arr, 1, 1, threads)                       # page-in + thread pool start
        per_turn = rollout(cfg, arr, 2, 2, threads) / 2
        turns = max(2, min(5000, int(seconds / max(per_turn, 1e-7))))
        dt = rollout(cfg, arr, 4, turns, threads)
        arr["turns_played"] = 3 + turns                        # epoch 0, turns 1 .. 3 + turns, from the reset state
        return E * A * turns / dt, turns, dt, arr

    E = 32768 if small else max(256, 16 * cores)               # >= 32 768 envs of the config-3 shape
    value, turns, dt, played = measure(E, cores, seconds_target)
    E1 = 512 if small else 16
    single, turns1, dt1, _ = measure(E1, 1, 3.0)
    out = {
        "value": value, "unit": "agent-steps/s", "cores": cores, "kind": "port",
        "per_core": value / cores, "single_core_value": single,
        "sample": f"the C port of the reference's step loop -- NOT the one-env-at-a-time numpy restatement SURVEY 8(d)(ii) named, which "
                  f"is ~10^4 x slower and would not finish a bounded sample: oracle/gridstep_oracle.c sgo_rollout (envs outer / turns inner, OpenMP static blocks, {cores} threads = this "
                  f"process's CPU share), {E} envs x {turns} turns of the same {spec.height}x{spec.width}x{A}-agent workload, "
                  f"{dt:.1f} s; one thread: {E1} envs x {turns1} turns, {dt1:.1f} s",
    }
    ref = REFERENCE_PYTHON_AGENT_STEPS_PER_S.get(config_name)
    if ref is not None:
        out["reference_python_single_core"] = ref
        out["reference_python_source"] = ("BASELINE.md section 2: the reference's own Environment.take_turn (sorrel/environment.py:81-93), "
                                          "one env of this shape, one Xeon 2.1 GHz core of the build container (the Python reference "
                                          "cannot travel to the GPU box)")
    return out, played


def check_against_oracle(spec, played, dev, obs_dtype):
    """The self-check of the line: the envs the CPU baseline just played (the C oracle: E envs from reset through
    `turns_played` turns of epoch 0, global env ids 0 .. E-1) are played again by the GPU engine -- a second, small
    engine, outside every timed region -- and the final grid, positions, float64 totals and the last turn's
    observations / rewards / actions are compared element by element.  Long horizon at the benchmark's own shape."""
    import numpy as np
    import torch

    from sorrel_amd.engine import GridEngine

    E, T = played["grid"].shape[0], int(played["turns_played"])
    eng = GridEngine(spec, E, device=dev, first_env_id=0, obs_dtype=obs_dtype)
    eng.reset(epoch=0)
    eng.rollout(T)
    torch.cuda.synchronize(dev)
    pairs = dict(grid=(eng.grid, played["grid"]), agent_pos=(eng.agent_pos, played["pos"]), total_reward=(eng.total_reward, played["tot"]),
                 actions=(eng.actions, played["act"]), rewards=(eng.rewards, played["rew"]), obs=(eng.obs, played["obs"]))
    differ = [k for k, (mine, ref) in pairs.items() if not np.array_equal(mine.cpu().numpy().astype(ref.dtype), ref)]
    status = eng.status()
    sum_total = float(eng.total_reward.sum().item())
    eng.close()
    return {"envs": E, "turns": T, "equal": not differ and status == 0, "tensors_that_differ": differ, "status": status,
            "sum_total_reward": sum_total, "oracle_sum_total_reward": float(played["tot"].sum()),
            "what": "the envs the cpu_baseline leg played with oracle/gridstep_oracle.c, replayed by a second GridEngine (sgw_rollout) "
                    "after the timed regions; grid, positions, float64 totals and the last turn's observations / rewards / actions "
                    "compared with np.array_equal"}


def rollout_bytes_per_turn(spec, E, T, obs_bytes):
    """HBM-side bytes one turn of ``sgw_rollout(T)`` moves: every turn's windows, actions and rewards; grid, positions and totals once per launch."""
    A = spec.num_agents
    return E * (A * (spec.num_channels * spec.window ** 2 * obs_bytes + 1 + 4)) + (E * (2 * spec.grid_bytes_per_env() + A * 4 + 16)) / T


def continue_against_oracle(eng, spec, turns, rollout=False):
    """From the state THIS engine is in (the launches that were just timed ended there): copy it to the host, step the engine `turns` more
    turns with the very call that was timed (``sgw_step`` with device-drawn actions; `rollout`: one ``sgw_rollout`` of that many turns) and
    oracle/gridstep_oracle.c from the copy, compare every tensor with np.array_equal (each turn; the rollout: after its last).  The checker
    only: nothing here is timed."""
    import ctypes as C

    import numpy as np
    import torch

    import __graft_entry__ as g

    lib = C.CDLL(g.ORACLE_LIB)
    E, A = eng.num_envs, spec.num_agents
    cfg = spec.to_config(E, eng.first_env_id)
    torch.cuda.synchronize(eng.device)
    arr = dict(grid=np.ascontiguousarray(eng.grid.cpu().numpy()), pos=eng.agent_pos.cpu().numpy().copy(), act=np.zeros((E, A), np.uint8),
               obs=np.zeros((E,) + spec.obs_shape, np.float32), rew=np.zeros((E, A), np.float32), tot=eng.total_reward.cpu().numpy().copy())
    t_at = eng.turn
    differ = []
    if rollout:
        eng.rollout(turns)
    for k in range(1, turns + 1):
        if not rollout:
            eng.step(random_actions=True)
        lib.sgo_step(C.byref(cfg), *(a.ctypes.data_as(C.c_void_p) for a in (arr["grid"], arr["pos"], arr["act"], arr["obs"], arr["rew"], arr["tot"])),
                     C.c_uint32(eng.epoch), C.c_uint32(t_at + k), C.c_int32(0), C.c_int32(A), C.c_uint32(1 | 2), C.c_int(host_cores()),
                     C.c_void_p(0), C.c_void_p(0), C.c_void_p(0))
        if rollout and k < turns:
            continue
        torch.cuda.synchronize(eng.device)
        for key, mine in (("grid", eng.grid), ("pos", eng.agent_pos), ("act", eng.actions), ("obs", eng.obs), ("rew", eng.rewards), ("tot", eng.total_reward)):
            if key not in differ and not np.array_equal(mine.cpu().numpy(), arr[key]):
                differ.append(key)
    return {"envs": E, "from_turn": t_at, "turns": turns, "equal": not differ and eng.status() == 0, "tensors_that_differ": differ,
            "kernel": eng.launch_info().split(" group")[0],
            "what": "from the state the timed launches ended in: the TIMED engine and oracle/gridstep_oracle.c (started from a copy of that state) "
                    "step on, every tensor compared with np.array_equal" + (" after the rollout's last turn" if rollout else " each turn")}


def side_config(name, dev, steps, prewarm, envs=0, check_turns=0, tune_placement=False, rollout_turns=0):
    """One more BASELINE shape in the same line: a short pre-warm, then `steps` launches between two
    HIP events on the launch stream.  Random actions, sweep on, float32 observations written, no reset in the region."""
    import torch

    from sorrel_amd.engine import GridEngine
    from sorrel_amd.spec import treasurehunt_spec

    H, W, A, r, E_cfg, p_spawn, p_dense = CONFIGS[name]
    E = envs or E_cfg
    spec = treasurehunt_spec(H, W, A, r, spawn_prob=p_spawn, seed=0, dense_prob=p_dense)
    eng = GridEngine(spec, E, device=dev, first_env_id=0)
    eng.reset(epoch=0)
    placement = None
    if tune_placement:       # where the observation tensor lies moves this kernel by up to 9 % (profiles/r05_c5_placement.txt): the best of four allocations
        placement = eng.pick_obs_placement(4, 30)
    for _ in range(prewarm):
        eng.step(random_actions=True)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        eng.step(random_actions=True)
    e1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / steps
    alg = spec.algorithmic_bytes_per_env_step() * E
    achieved = alg / (kernel_ms * 1e-3) / 1e9
    checked = continue_against_oracle(eng, spec, check_turns) if check_turns > 0 else None
    rollout_leg = None
    if rollout_turns > 0:   # the same shape through sgw_rollout: T whole turns per launch (how a user with a launch-bound batch runs it)
        calls = max(2, steps // 4)
        eng.rollout(rollout_turns)
        for _ in range(calls):
            eng.rollout(rollout_turns)
        torch.cuda.synchronize(dev)
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for _ in range(calls):
            eng.rollout(rollout_turns)
        r1.record()
        torch.cuda.synchronize(dev)
        rms = r0.elapsed_time(r1) / (calls * rollout_turns)
        moved = rollout_bytes_per_turn(spec, E, rollout_turns, 4)
        rollout_leg = {"turns_per_launch": rollout_turns, "calls": calls, "ms_per_turn": rms, "bytes_moved_per_turn": moved,
                       "algorithmic_frac": alg / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS, "hbm_side_frac": moved / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "value": E * A / (rms * 1e-3),
                       "what": "sgw_rollout, T turns per launch: algorithmic_frac prices a turn at SURVEY 8d's per-turn bytes (same numerator as `frac`), "
                               "hbm_side_frac at what a turn of the rollout moves (windows / actions / rewards of every turn + grid, positions, totals once per launch)"}
        if check_turns > 0:
            rollout_leg["checked_vs_oracle_equal"] = continue_against_oracle(eng, spec, 2, rollout=True)["equal"]
    out = {"workload": f"{H}x{W} grid x {spec.layers} layers, {A} agents, {spec.window}x{spec.window} window, {E} envs on one GPU",
           "envs": E, "steps": steps, "prewarm_steps": prewarm, "kernel_ms": kernel_ms, "wall_ms_per_step": wall / steps * 1e3,
           "value": E * A / (kernel_ms * 1e-3), "unit": "agent-steps/s",
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "frac_of_copy_ceiling": achieved / HBM_COPY_GBS, "algorithmic_bytes_per_launch": alg},
           "kernel": eng.launch_info(), "status": eng.status()}
    if checked is not None:
        out["checked_vs_oracle"] = checked
    if rollout_leg is not None:
        out["rollout"] = rollout_leg
    if placement is not None:
        out["obs_placement"] = dict(placement, what="GridEngine.pick_obs_placement: us per launch with the observation tensor in each of four allocations; the fastest kept "
                                                     "(the walking workgroups' dword stores are sensitive to where the tensor lies, stable per allocation)")
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return out


def policy_turn_bench(eng, iters: int = 40):
    """Engine time of a POLICY-DRIVEN turn (SURVEY a6 as a trained model runs it: pov -> get_action -> act, agent after agent)
    on the engine the headline was measured on, outside its timed region: the entity sweep + every agent's window once
    (SGW_STEP_NO_MOVE), then one sgw_act per agent (the act + the repair of the later agents' windows).  Actions are
    precomputed (a real policy's forward pass comes on top); HIP events around the loop, per turn."""
    import torch

    A = eng.spec.num_agents
    eng.random_actions()                       # some valid actions in eng.actions
    rows = eng.window_rows(None)
    dests = [torch.empty((eng.num_envs, eng.obs[0, 0].numel()), dtype=eng.obs_dtype, device=eng.device) for _ in range(A)]
    rows_own = eng.window_rows(dests)

    def timed(fn, warm):
        for _ in range(warm):                      # ~60 ms of uninterrupted launches: what ran before left the chip idle for a while
            fn()
        torch.cuda.synchronize(eng.device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize(eng.device)
        return a.elapsed_time(b) / iters

    def fused():
        eng.step(eng.actions)

    def tensor_windows():
        eng.turn += 1
        eng.step(eng.actions, sweep=True, no_move=True, turn=eng.turn)
        for a in range(A):
            eng.act(a, rows)

    def replay_rows():
        eng.turn += 1
        eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=eng.turn)
        eng.observe_rows(rows_own)
        for a in range(A):
            eng.act(a, rows_own)

    def replay_rows_fused():
        eng.turn += 1
        eng.sweep_observe_rows(rows_own, sweep=True, turn=eng.turn)
        for a in range(A):
            eng.act(a, rows_own)

    out = {"fused_turn_ms": timed(fused, 500), "policy_turn_ms": timed(tensor_windows, 400), "launches": 1 + A,
           "what": "policy_turn_ms: sgw_step(SGW_STEP_NO_MOVE) = sweep + every agent's window (into the observation tensor), then "
                   "sgw_act per agent; fused_turn_ms: the same engine's one-launch turn with given actions; the policy's own forward "
                   "pass is not in either"}
    from sorrel_amd import _native as N
    if eng.capabilities() & N.CAP_OBSERVE_ROWS:
        out["policy_turn_replay_rows_ms"] = timed(replay_rows, 300)
        out["replay_rows_what"] = "windows rendered straight into per-agent [E][C*V*V] rows (what Environment.take_turn does when every agent has a replay Buffer): sweep, sgw_observe_rows, sgw_act per agent"
    if eng.capabilities() & N.CAP_SWEEP_ROWS:
        out["policy_turn_replay_rows_one_launch_ms"] = timed(replay_rows_fused, 300)
        out["replay_rows_one_launch_what"] = "the same with sweep + every window in ONE launch (sgw_sweep_observe_rows, round 5): what Environment.take_turn does now"
    out["status"] = eng.status()
    return out


def write_only_probe(dev, nbytes: int, iters: int = 60):
    """What a linear write-only kernel reaches on THIS card right now: torch's ``fill_`` over a buffer of the observation tensor's size.
    Context for ``roofline.frac`` (priced against the 8 TB/s peak): the step kernels are write-dominated."""
    import torch

    x = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    for _ in range(20):
        x.fill_(1.0)
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        x.fill_(2.0)
    b.record()
    torch.cuda.synchronize(dev)
    us = a.elapsed_time(b) * 1e3 / iters
    del x
    torch.cuda.empty_cache()
    return {"what": "torch fill_ over a buffer of the observation tensor's size: a linear write-only stream, nothing else",
            "bytes": nbytes, "us": us, "tb_per_s": nbytes / us / 1e6, "of_peak": nbytes / us / 1e6 / 8.0}


def recorded_turn_bench(dev, envs: int = 1024, turns: int = 400):
    """Wall time per ``Environment.take_turn()`` through the Python API for the headline's world shape (32x32, 8 agents, 7x7) at a
    batch where the eager agent loop is host-bound: a one-layer torch policy per agent with replay memories, (a) the eager loop (sweep +
    windows + A x (policy, sgw_act) from Python), (b) the same turn RECORDED once (``Environment.capture_turn``: the turn number and
    the replay rows are counted on the device, ``sgw_turn_*``) and replayed, (c) recorded, with the policy handing its action VALUES
    to the act launch (``SGW_ACT_QF32``: argmax + epsilon exploration in-kernel).  Not part of `value`."""
    import time

    import torch
    from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
    from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
    from sorrel_amd.examples.treasurehunt.main import make_config
    from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld
    from sorrel_amd.models import BaseModel

    def factory(values):
        class Linear(BaseModel):
            def __init__(self, input_size, action_space):
                super().__init__(input_size, action_space, memory_size=64, num_envs=envs, device=dev)
                self.w = torch.randn(int(input_size[0]), action_space, generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)
                self.epsilon = 0.05 if values else 0.0

            def take_action(self, state):
                q = state.reshape(state.shape[0], -1) @ self.w
                return q if values else q.argmax(dim=1)

        return Linear

    def run(values, capture, generic=False):
        cfg = make_config(32, 32, 8, 3, spawn_prob=0.005)
        env = TreasurehuntEnv(TreasurehuntWorld(cfg, EmptyEntity(), num_envs=envs, device=dev, seed=0), cfg, model_factory=factory(values))
        env.fast_policy_loop = not generic
        if capture and env.capture_turn() is None:
            raise RuntimeError(f"not recordable: {getattr(env, 'capture_error', None)!r}")
        for _ in range(50 - (2 if capture else 0)):      # (capture_turn played two real turns before it recorded: the same count either way)
            env.take_turn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(turns):
            env.take_turn()
        torch.cuda.synchronize(dev)
        us = (time.perf_counter() - t0) / turns * 1e6
        env.raise_on_status()
        import hashlib

        # what the run ended in: world state, the step's outputs and every agent's replay ring -- a recorded turn must leave exactly
        # what the eager loop leaves (same policies, same seeds, same number of turns)
        h = hashlib.sha256()
        for t in [env.world.grid, env.world.agent_pos, env.world.total_reward, env.rewards, env.actions] + \
                [x for a in env.agents for x in (a.model.memory.states, a.model.memory.actions, a.model.memory.rewards)]:
            h.update(t.cpu().numpy().tobytes())
        return us, h.hexdigest()[:16]

    out = {"workload": f"32x32 grid x 2 layers, 8 agents, 7x7 window, {envs} envs, one linear policy per agent, 64-row replay memories",
           "unit": "us per take_turn (wall)", "turns": turns}
    try:
        out["eager_loop"], eager_state = run(False, False)
        out["eager_generic_loop"], generic_state = run(False, False, generic=True)
        out["recorded"], recorded_state = run(False, True)
        out["recorded_action_values"], _ = run(True, True)
        out["recorded_equals_eager"] = recorded_state == eager_state      # (digest of grid, positions, totals, step outputs and all replay rings)
        out["generic_equals_eager"] = generic_state == eager_state
        out["what"] = ("eager_loop: Python drives sweep, windows and per agent policy + sgw_act (Environment.fast_policy_loop: agents with the "
                       "standard hooks); eager_generic_loop: the same through Agent.transition's hooks (round 4's loop); recorded: the same turn as "
                       "ONE graph replay (27 dependent launches); recorded_action_values: the act launch takes argmax / explores itself (19 launches)")
    except Exception as exc:      # (reported, never fatal for the line)
        out["error"] = repr(exc)[:300]
    return out


def many_agents_turn_bench(dev, envs: int = 2048, turns: int = 60):
    """Config 5's shape (128x128 grid x 2 layers, 64 agents, 11x11 windows: BASELINE.json's configs[4]) through the Python API with
    ONE linear policy and ONE replay ring shared by every agent: the eager agent-after-agent loop against ``Environment.speculate_turns``
    (every policy evaluated in one batch on the pre-move windows, ``sgw_turn_resolve`` finds whose window an earlier mover changed, those
    are evaluated again until nothing changes: the sequential turn, bit for bit).  Wall us per take_turn; not part of `value`."""
    import hashlib
    import time

    import torch
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
    from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
    from sorrel_amd.examples.treasurehunt.main import make_config
    from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld
    from sorrel_amd.models import BaseModel

    h, w, a, r, _envs, p_spawn, _dense = CONFIGS["c5"]

    def run(speculate):
        one = []

        class Shared(BaseModel):
            def __init__(self, input_size, action_space):
                super().__init__(input_size, action_space, memory_size=0, num_envs=envs, device=dev)
                self.memory = Buffer(capacity=4 * a, obs_shape=tuple(input_size), num_envs=envs, device=dev)
                self.w = torch.randn(int(input_size[0]), action_space, generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)

            def take_action(self, state):
                return (state.reshape(state.shape[0], -1) @ self.w).argmax(dim=1)

        def factory(input_size, action_space):
            if not one:
                one.append(Shared(input_size, action_space))
            return one[0]

        cfg = make_config(h, w, a, r, spawn_prob=p_spawn)
        env = TreasurehuntEnv(TreasurehuntWorld(cfg, EmptyEntity(), num_envs=envs, device=dev, seed=0), cfg, model_factory=factory)
        env.speculate_turns = speculate                # (True: the cost model agrees for 64 agents on one model)
        for _ in range(12):
            env.take_turn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(turns):
            env.take_turn()
        torch.cuda.synchronize(dev)
        us = (time.perf_counter() - t0) / turns * 1e6
        env.raise_on_status()
        d = hashlib.sha256()
        mem = one[0].memory
        for t in (env.world.grid, env.world.agent_pos, env.world.total_reward, env.rewards, env.actions, mem.actions, mem.rewards):
            d.update(t.cpu().numpy().tobytes())
        passes = getattr(env, "speculation_passes", None)
        del env, one
        torch.cuda.empty_cache()
        return us, d.hexdigest()[:16], passes

    out = {"workload": f"{h}x{w} grid, {a} agents, {2 * r + 1}x{2 * r + 1} window, {envs} envs, ONE linear policy + one replay ring shared by all agents",
           "unit": "us per take_turn (wall)", "turns": turns}
    try:
        out["eager_loop"], eager_state, _ = run(False)
        out["speculative"], spec_state, out["passes_of_the_last_turn"] = run(True)
        out["speculative_equals_eager"] = spec_state == eager_state       # (grid, positions, totals, step outputs, the ring's actions and rewards)
    except Exception as exc:
        out["error"] = repr(exc)[:300]
    return out


def ensure_built() -> None:
    """Both native libraries, checked (and, outside a profiler, rebuilt if stale) BEFORE torch or anything else touches
    the GPU: no compiler is ever started from a GPU-initialised process.  Ranks of one node take turns on a lock."""
    import fcntl

    import __graft_entry__ as graft

    profiled = any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if graft.libs_up_to_date():
                return
            if profiled:
                raise SystemExit("bench.py: libsgw.so / libgridstep_oracle.so are stale or missing and this process runs under a "
                                 "profiler (its preload has already initialised the GPU): run "
                                 "`python3 -c 'import __graft_entry__ as g; g.build()'` first")
            try:
                graft.build_hip()
                graft.build_oracle()
            except Exception as exc:   # no compiler on this box: an existing library is still the product under test
                if not (os.path.isfile(graft.HIP_LIB) and os.path.isfile(graft.ORACLE_LIB)):
                    raise
                print(f"bench.py: could not rebuild the native libraries ({exc}); using the existing ones", file=sys.stderr)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks as a FRESH child (``python -m torch.distributed.run``, one
    process per GPU) with this very command line, relay its output and exit code.  This process has made no GPU call (nothing here
    imports torch; the libraries were checked just above, so the ranks find them built) and never replaces itself: no exec."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, env=env, cwd=ROOT)
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        return child.wait()


def series_stats(ms):
    if not ms:
        return None
    s = sorted(ms)

    def q(f):
        return s[min(len(s) - 1, int(f * len(s)))]

    return {"n": len(ms), "median_ms": q(0.5), "p10_ms": q(0.1), "p90_ms": q(0.9), "min_ms": s[0], "max_ms": s[-1],
            "mean_ms": sum(ms) / len(ms), "first5_mean_ms": sum(ms[:5]) / len(ms[:5])}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the config's)")
    ap.add_argument("--rewarm-steps", type=int, default=600,
                    help="untimed launches between the drain of the pre-warm pass's per-launch timers (a few ms of GPU idle) and the "
                         "warm-up launches: the timed region then starts on a chip that has been busy for ~70 ms without a gap")
    ap.add_argument("--prewarm-steps", type=int, default=1500,
                    help="untimed clock-ramp steps before the W warm-up steps (0 = none); reported in the JSON line")
    ap.add_argument("--max-turns", type=int, default=0,
                    help="epoch length: every env is auto-reset (K3, inside sgw_step) after this many turns, inside the "
                         "timed loop too; 0 = one endless epoch (SURVEY 8d: no reset in the timed region)")
    ap.add_argument("--turns-per-launch", type=int, default=50,
                    help="after the timed region: the same workload through sgw_rollout, this many turns per call (the env's "
                         "grid stays in LDS from turn to turn); reported beside the per-turn numbers as fused_rollout; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-series", action="store_true", help="skip the per-launch timing pass after the timed region")
    ap.add_argument("--graph", action="store_true",
                    help="replay ONE hipGraph of the K timed launches instead of calling sgw_step K times (same launches, same turn numbers, same "
                         "results; measured on the headline: 0.7 us more per launch between the events, 0.8 us less on the wall clock)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-obs", action="store_true", help="diagnostic: skip observation stores (NOT a valid bench line)")
    ap.add_argument("--diag-agents", type=int, default=-1, help="diagnostic: step only the first N agents (NOT valid)")
    ap.add_argument("--obs-dtype", default="f32", choices=["f32", "u8"],
                    help="u8 = compact one-hot counts (an extra, reported separately; the contract format is f32)")
    ap.add_argument("--no-sweep", action="store_true", help="diagnostic: skip the entity sweep (NOT a valid bench line)")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the short runs of the other 1-GPU BASELINE shapes (config 2, config 5's per-GPU share, config 3 at "
                         "524 288 envs) that the headline run adds to the line as `configs`")
    ap.add_argument("--side-steps", type=int, default=100, help="timed launches of each side config")
    ap.add_argument("--no-self-check", action="store_true",
                    help="skip replaying the CPU baseline's envs on the GPU and comparing the results (rollout.checked_vs_oracle)")
    args = ap.parse_args()

    ensure_built()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)

    import torch
    import torch.distributed as dist

    import __graft_entry__ as graft
    from sorrel_amd.engine import GridEngine
    from sorrel_amd.spec import treasurehunt_spec

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
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
    # under a launcher (torch.distributed.run exports RANK / WORLD_SIZE / MASTER_*) the process group exists even for ONE rank: a 1-GPU box then
    # runs the very RCCL calls of the N > 1 line (tests/test_gpu_distributed.py); a plain `python bench.py` (the driver's N = 1 command) has none
    grouped = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    backend = None
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if rehearsal else "nccl"
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
    if args.max_turns > 0:
        eng.set_auto_reset(args.max_turns)

    def barrier():
        if grouped:
            if rehearsal:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev_index])
        # spin on an event before the blocking synchronize: a sleeping host thread wakes up tens of microseconds after
        # the GPU is done, which is 1-2 % of the driver's 2.4 ms timed region (20 steps)
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize(dev)

    write_obs = not args.no_obs
    sweep = not args.no_sweep
    if args.diag_agents >= 0:
        _orig_step = eng.step
        eng.step = lambda *a, **k: _orig_step(*a, agent_end=args.diag_agents, **k)

    def step():
        eng.step(random_actions=True, write_obs=write_obs, sweep=sweep)

    # --graph: the K timed launches as ONE hipGraph (captured here, before anything is warm: a capture is host work and would leave
    # the chip idle for a millisecond if it sat between the warm launches and the timed region).  Every node is an ordinary sgw_step
    # launch with the turn number it will carry when the region runs; the region then costs the host one replay call instead of K
    # sgw_step calls, so a host thread descheduled for a few hundred microseconds inside the driver's 2.4 ms region (seen once in
    # eight runs: wall 0.134 ms per step against 0.122 between the events) cannot starve the GPU.  Not the default: same box, three
    # runs each, graph / plain loop: kernel_ms 0.1187 / 0.1180 (graph nodes start ~0.7 us apart), ms_per_step 0.1216 / 0.1224.
    # The results are identical either way: same launches, same turn numbers -- a GPU test compares them.
    timed_graph = None
    lead = (max(0, args.prewarm_steps) + max(0, args.rewarm_steps) if args.prewarm_steps > 0 else 0) + args.warmup   # launches before the region
    if args.graph and args.max_turns == 0 and args.diag_agents < 0:
        saved_turn = eng.turn
        try:
            eng.turn = saved_turn + lead
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(args.steps):
                    step()
            timed_graph = g
        except Exception as exc:   # no capture on this stack: the plain loop
            print(f"bench.py: hipGraph capture of the timed launches failed ({exc}); submitting them one by one", file=sys.stderr)
            timed_graph = None
        finally:
            eng.turn = saved_turn
        torch.cuda.synchronize(dev)

    # the pre-warm pass: untimed for `value`, but every launch of it sits between its own pair of HIP events, so the line
    # shows the cold start next to the settled kernel
    prewarm_series = None
    if args.prewarm_steps > 0:
        eng.set_timing(True)
        for _ in range(args.prewarm_steps):
            step()
        torch.cuda.synchronize(dev)
        pw = eng.step_times_ms()
        eng.set_timing(False)

        def mean(v):
            return sum(v) / len(v) if v else None

        prewarm_series = {"n": len(pw), "launches_0_10_mean_ms": mean(pw[:10]), "launches_10_100_mean_ms": mean(pw[10:100]),
                          "launches_100_400_mean_ms": mean(pw[100:400]), "last_100_mean_ms": mean(pw[-100:]) if len(pw) >= 200 else None,
                          "what": "the untimed pre-warm launches, first launch of the process onwards, each between its own pair of HIP "
                                  "events: a fresh process runs launches ~10-100 slower than the settled kernel (profiles/r03_c3_launch_ramp.txt "
                                  "has the clock / power samples taken alongside)"}
    # Draining the 1 500 event pairs above leaves the GPU idle for a few milliseconds, and this chip answers an idle of that length
    # with its power ramp again (tools/idle_probe.py, config 3: after 0 / 1 / 3 / 10 / 30 ms of idle the next 20 launches take
    # 118 / 119 / 123-126 / 133-143 / 141-145 us, and the 200 after a 10 ms idle still 129) -- so the launches flow on without a
    # host-side gap from here to the timed region: `rewarm_steps` untimed launches, the W warm-up launches, barrier, K timed.
    for _ in range(max(0, args.rewarm_steps) if args.prewarm_steps > 0 else 0):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                                   # on the stream sgw_step launches on
    if timed_graph is not None:
        timed_graph.replay()                       # the K launches, turn numbers eng.turn + 1 .. eng.turn + K
        eng.turn += args.steps
    else:
        for _ in range(args.steps):
            step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events over the timed region, per launch

    # the same K steps again, every launch between its own pair of HIP events (outside the timed region: two event
    # packets per launch would sit in the measured stream otherwise)
    series = None
    if not args.no_series:
        eng.set_timing(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        series_wall = time.perf_counter() - t1
        per_launch = eng.step_times_ms()
        eng.set_timing(False)
        series = series_stats(per_launch)
        if series is not None:
            series["pass_wall_ms_per_step"] = series_wall / args.steps * 1e3
            series["what"] = ("a second pass of the same K steps right after the timed region, one pair of HIP events per launch on the "
                              "launch stream (sgw_set_timing); first5_mean = the first five launches of that pass")

    # the same workload through sgw_rollout: T turns per call, the grid resident in LDS between turns (reported BESIDE
    # the per-turn numbers: a launch no longer moves the grid of every turn, so the algorithmic-bytes accounting of the
    # per-turn launch does not describe it; its own HBM-side bytes are the observation / action / reward / position
    # writes of every turn plus one grid read + write-back per call)
    fused = None
    if args.turns_per_launch > 0 and write_obs and sweep and args.diag_agents < 0 and args.max_turns == 0:
        T = max(1, min(args.turns_per_launch, args.steps))
        calls = max(1, args.steps // T)
        eng.rollout(T)                                  # warm
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tw = time.perf_counter()
        e0.record()
        for _ in range(calls):
            eng.rollout(T)
        e1.record()
        barrier()
        fwall = time.perf_counter() - tw
        fms = e0.elapsed_time(e1) / (calls * T)
        per_turn_moved = rollout_bytes_per_turn(spec, E, T, 4 if args.obs_dtype == "f32" else 1)
        fused = {"turns_per_launch": T, "calls": calls, "ms_per_step": fms, "wall_ms_per_step": fwall / (calls * T) * 1e3,
                 "value_one_rank": E * A / (fms * 1e-3), "unit": "agent-steps/s",
                 "bytes_moved_per_step": per_turn_moved, "hbm_side_achieved": per_turn_moved / (fms * 1e-3) / 1e9,
                 "hbm_side_frac": per_turn_moved / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "what": "sgw_rollout: T whole turns per call; every turn's observations, actions and rewards are written, the grid "
                         "is read once and written back once per call (it stays in LDS in between), so a step moves fewer bytes than "
                         "SURVEY 8d's per-turn-launch formula -- hence its own bytes_moved_per_step instead of roofline.frac"}

    # what one barrier + synchronize costs on this job (it closes the timed region, so it is inside `value`): stated, not hidden
    tb = time.perf_counter()
    barrier()
    barrier_ms = (time.perf_counter() - tb) * 1e3

    # end-of-rollout metrics: on-device reduction + the one collective
    metrics = eng.reduce_metrics().clone()
    kernel_ms_rank = kernel_ms
    tmax = torch.tensor([dt, kernel_ms, barrier_ms], dtype=torch.float64, device=dev)
    per_rank_kernel_ms, group_size = [kernel_ms], 1
    if grouped:
        group_size = dist.get_world_size()       # what the process group itself says (the line's n_gpus is checked against it)
        mine = torch.zeros((world,), dtype=torch.float64, device=dev)
        mine[rank] = kernel_ms
        from sorrel_amd.distributed import all_reduce_metrics

        metrics = all_reduce_metrics(metrics)                # the product's ONE collective: SUM of the float64[4] metric vector (RCCL; gloo in a rehearsal)
        if rehearsal:     # gloo reduces host tensors
            tmax, mine = tmax.cpu(), mine.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)      # (every rank wrote its own slot: the sum is the list)
        per_rank_kernel_ms = [float(x) for x in mine.tolist()]
    dt, kernel_ms, barrier_ms = float(tmax[0].item()), float(tmax[1].item()), float(tmax[2].item())   # MAX over ranks: the slowest rank's kernel prices the roofline
    status = eng.status()

    if rank == 0:
        total_envs = E * world
        value = total_envs * A * args.steps / dt
        grid_bytes = spec.grid_bytes_per_env() * E
        alg_bytes = spec.algorithmic_bytes_per_env_step() * E          # per launch (one rank's kernel)
        if not write_obs:
            alg_bytes -= E * A * spec.num_channels * spec.window ** 2 * 4
        elif args.obs_dtype == "u8":
            alg_bytes -= E * A * spec.num_channels * spec.window ** 2 * 3      # C*V*V*1 instead of *4
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # What of that the design says never reaches HBM: while the batch's grids fit the Infinity Cache + L2 (the
        # observation bursts are streaming stores and leave the caches to the grids), the grid READ of a turn is
        # served on-die.  The write-back is still counted as HBM traffic (a memory-side cache may keep it too).
        cache_served = grid_bytes if grid_bytes <= CACHE_RESIDENT_GRID_BYTES else 0
        hbm_side = (alg_bytes - cache_served) / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), collected by
        # tools/profile_gpu.sh on this same command and committed under profiles/; null unless that file was measured
        # on exactly this kernel source (hash) and workload
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
        try:
            with open(tpath) as fh:
                tj = json.load(fh)
            src_hash = graft.source_digest()
            if tj.get("envs") == E and write_obs and sweep and args.obs_dtype == "f32" and args.max_turns == 0:
                if tj.get("sgw_source_sha256") == src_hash:
                    traffic = tj["hbm_bytes_per_launch"]
                    traffic_source = f"profiles/traffic_{args.config}.json (rocprofv3 --pmc passes on kernel source {src_hash[:12]}; not measured in this run)"
                else:
                    traffic_source = (f"profiles/traffic_{args.config}.json was measured on kernel source {str(tj.get('sgw_source_sha256'))[:12]}, "
                                      f"this run is {src_hash[:12]}: stale, not reported")
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "agent-steps/sec" if args.obs_dtype == "f32" else "agent-steps/sec (compact uint8 observations; NOT the contract metric)",
            "value": value, "unit": "agent-steps/s", "n_gpus": group_size if grouped else 1,
            "steps": args.steps, "warmup": args.warmup, "prewarm_steps": max(0, args.prewarm_steps),
            "rewarm_steps": max(0, args.rewarm_steps) if args.prewarm_steps > 0 else 0,
            "timed_region_submission": "one hipGraph replay of the K sgw_step launches (captured before the pre-warm pass, each node with the turn number it carries)" if timed_graph is not None else "K sgw_step calls",
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 grid, " + args.obs_dtype + " obs",   # entity type ids are uint8 (the path's arithmetic); observations leave as float32
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {H}x{W} grid x {spec.layers} layers, {A} agents, {spec.window}x{spec.window} window, "
                            f"{E} envs/GPU x {world} GPU = {total_envs} envs, treasurehunt rules, random actions, {args.obs_dtype} one-hot obs",
                "envs_per_gpu": E, "global_envs": total_envs, "agents": A, "grid": [H, W, spec.layers],
                "window": spec.window, "channels": spec.num_channels, "spawn_prob": p_spawn, "dense_prob": p_dense,
                "sharding": f"env-batch x{world} by global env id, no data-path collective; after the rollout ONE 32-byte SUM all-reduce of the metric "
                            f"vector (the product's only collective), plus this script's own timing reductions (one MAX over [wall, kernel_ms, barrier_ms], "
                            f"one SUM that gathers kernel_ms per rank) and the barriers that bracket the timed region",
                "process_group_world_size": group_size, "collectives_backend": backend,
                "obs_written": write_obs, "sweep": sweep, "max_turns": args.max_turns,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "what": "ALGORITHMIC bytes per launch (SURVEY 8d) / average launch duration; it can exceed the copy ceiling because "
                        "cache-served bytes are in the numerator -- see hbm_side_*",
                "frac_of_copy_ceiling": achieved / HBM_COPY_GBS,   # both denominators
                "hbm_side_achieved": hbm_side, "hbm_side_frac": hbm_side / HBM_PEAK_GBS,
                "hbm_side_frac_of_copy_ceiling": hbm_side / HBM_COPY_GBS, "cache_served_bytes_per_launch": cache_served,
                "hbm_side_what": "MODELLED, not measured: algorithmic bytes minus the grid READ of a turn, which the Infinity Cache serves while the batch's grids "
                                 f"({grid_bytes / 1e6:.0f} MB here) stay resident (<= {CACHE_RESIDENT_GRID_BYTES >> 20} MiB); 0 subtracted otherwise",
                "kernel": eng.launch_info(), "kernel_ms": kernel_ms, "kernel_ms_what": "HIP events over the timed region / steps; MAX over ranks for N > 1",
                "kernel_ms_rank0": kernel_ms_rank, "kernel_ms_per_rank": per_rank_kernel_ms, "series": series, "prewarm_series": prewarm_series,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_per_agent_step": spec.algorithmic_bytes_per_env_step() / A,
            },
            "fused_rollout": fused,
            "timing": {"region": "barrier + synchronize | K launches | barrier + synchronize, MAX over ranks",
                       "barrier_plus_synchronize_ms": barrier_ms,
                       "barrier_what": "one more barrier + synchronize timed right after the region (MAX over ranks): the closing one is inside "
                                       "`value`; kernel_ms is not affected by it"},
            "rollout": {"sum_total_reward": float(metrics[0].item()), "envs": float(metrics[2].item()), "status": status,
                        "first_env_id_rank0": 0, "first_env_id_last_rank": (world - 1) * E},
        }
        valid_line = write_obs and sweep and args.diag_agents < 0 and args.obs_dtype == "f32"
        rf = out["roofline"]
        if prewarm_series is not None:
            rf["prewarm_10_100_ms"] = prewarm_series["launches_10_100_mean_ms"]    # what a short-lived process gets (the driver's record keeps scalars of `roofline`)
            rf["prewarm_last_100_ms"] = prewarm_series["last_100_mean_ms"]
        if fused is not None:
            rf["c3_rollout_ms_per_turn"] = fused["ms_per_step"]
            rf["c3_rollout_algorithmic_frac"] = alg_bytes / (fused["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if world == 1 and valid_line and not args.no_self_check and args.max_turns == 0:
            # the self-check of the headline on the engine (and code object) that was just timed: from the state its launches ended in
            chk = continue_against_oracle(eng, spec, 3)
            out["rollout"]["timed_engine_checked_vs_oracle"] = chk
            rf["timed_engine_checked_equal"] = chk["equal"]
            rf["timed_engine_checked_what"] = f"{chk['envs']} envs x {chk['turns']} turns from turn {chk['from_turn']} on {chk['kernel']} vs the C oracle, all tensors"
        if world == 1 and valid_line and not args.no_side_configs and args.max_turns == 0:
            out["policy_turn"] = policy_turn_bench(eng)          # (after every timed region of the headline; same engine)
            pt = out["policy_turn"]
            for key in ("fused_turn_ms", "policy_turn_ms", "policy_turn_replay_rows_ms", "policy_turn_replay_rows_one_launch_ms"):
                if key in pt:
                    rf[key] = pt[key]
            out["roofline"]["write_only_probe"] = write_only_probe(dev, eng.obs.numel() * eng.obs.element_size())
            rf["write_only_tb_per_s"] = rf["write_only_probe"]["tb_per_s"]
        if world == 1 and args.config == "c3" and valid_line and not args.no_side_configs:
            # the other 1-GPU BASELINE shapes, briefly, AFTER the headline run (its numbers are not touched by them)
            eng_obs, eng.obs = eng.obs, None                       # give the headline's 617 MB observation tensor back first
            del eng_obs
            torch.cuda.empty_cache()
            out["configs"] = {
                # (pre-warm counts: ~60 ms of uninterrupted launches each -- the engine's creation leaves the chip idle)
                "c2": side_config("c2", dev, args.side_steps, 5000, check_turns=0 if args.no_self_check else 3, rollout_turns=args.turns_per_launch),
                "c5": side_config("c5", dev, args.side_steps, 700, check_turns=0 if args.no_self_check else 3, tune_placement=True),
                "c3_524288": side_config("c3", dev, max(10, args.side_steps // 2), 60, envs=524288),
            }
            torch.cuda.empty_cache()
            # the same three, in brief, INSIDE `roofline` (the driver's record keeps `roofline` and `cpu_baseline` of the line): launch time,
            # fraction of the 8 TB/s peak on the same algorithmic-bytes definition as the headline, and the oracle self-check
            out["roofline"]["side_configs"] = {
                k: {"kernel_ms": v["kernel_ms"], "frac": v["roofline"]["frac"], "envs": v["envs"], "obs_placement_us": (v.get("obs_placement") or {}).get("candidates_us"),
                    "checked_vs_oracle_equal": (v.get("checked_vs_oracle") or {}).get("equal"), "kernel": v["kernel"].split(" group")[0]}
                for k, v in out["configs"].items()}
            # ... and FLAT, as scalars of `roofline` itself: the driver's record keeps only those
            for k, v in out["configs"].items():
                rf[f"{k}_kernel_ms"], rf[f"{k}_frac"] = v["kernel_ms"], v["roofline"]["frac"]
                if v.get("checked_vs_oracle") is not None:
                    rf[f"{k}_checked_vs_oracle_equal"] = v["checked_vs_oracle"]["equal"]
                if v.get("rollout") is not None:
                    rf[f"{k}_rollout_ms_per_turn"], rf[f"{k}_rollout_frac"] = v["rollout"]["ms_per_turn"], v["rollout"]["algorithmic_frac"]
                    rf[f"{k}_rollout_hbm_side_frac"] = v["rollout"]["hbm_side_frac"]
                    if "checked_vs_oracle_equal" in v["rollout"]:
                        rf[f"{k}_rollout_checked_vs_oracle_equal"] = v["rollout"]["checked_vs_oracle_equal"]
                pl = v.get("obs_placement")
                if pl and pl.get("candidates_us"):     # the picked placement is the best of four: the first and the median beside it
                    c = sorted(pl["candidates_us"])
                    to_frac = v["roofline"]["algorithmic_bytes_per_launch"] / 1e3 / HBM_PEAK_GBS
                    rf[f"{k}_first_placement_frac"] = to_frac / pl["candidates_us"][0]
                    rf[f"{k}_median_placement_frac"] = to_frac / (0.5 * (c[(len(c) - 1) // 2] + c[len(c) // 2]))
            out["recorded_turn"] = recorded_turn_bench(dev)      # (the Python API at a host-bound batch: eager loop vs one graph replay per turn)
            out["many_agents_turn"] = many_agents_turn_bench(dev)   # (config 5's shape: the eager loop vs the speculative turn)
            # ... and, in brief, inside `roofline` like the side configs (wall us per Environment.take_turn through the Python API)
            rt, ma = out["recorded_turn"], out["many_agents_turn"]
            out["roofline"]["side_configs"]["policy_turns_us"] = {
                "c3_shape_1024_envs": {k: rt.get(k) for k in ("eager_loop", "eager_generic_loop", "recorded", "recorded_action_values",
                                                               "recorded_equals_eager", "generic_equals_eager", "error") if k in rt},
                "c5_shape_2048_envs": {k: ma.get(k) for k in ("eager_loop", "speculative", "passes_of_the_last_turn", "speculative_equals_eager", "error") if k in ma}}
            for key in ("eager_loop", "recorded", "recorded_equals_eager"):
                if key in rt:
                    rf[f"take_turn_1024_envs_{key}" + ("" if key.endswith("eager") else "_us")] = rt[key]
            for key in ("eager_loop", "speculative", "speculative_equals_eager"):
                if key in ma:
                    rf[f"take_turn_c5_{key}" + ("" if key.endswith("eager") else "_us")] = ma[key]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], played = cpu_baseline(spec, args.config, args.cpu_seconds)
            if valid_line and not args.no_self_check:
                out["rollout"]["checked_vs_oracle"] = check_against_oracle(spec, played, dev, obs_dtype)
        print(json.dumps(out), flush=True)
    if grouped:
        barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""BASELINE config 5 (128x128x2, 64 agents, 11x11 windows): workgroups per CU of step_big, A/B on ONE card in ONE process.

Variants (engines created under different options, timed round-robin so that every variant sees the same card, clocks and
temperature): the walking variant (resident workgroups walk the batch), one env per workgroup with direct stores, one env per
workgroup with staged windows -- each of the last two at the occupancy the code object admits (SGW_BIG_WAVES = 8 waves per SIMD:
four 512-thread workgroups per CU) and capped to three per CU through the LDS request (what rounds 2-4 ran).  us per launch, the
tensors of every variant digested (equal across variants or the line says so).  GPU only.
usage: tools/c5_occupancy_ab.py [envs ...]      (default 1024 2048 4096 8192)"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from _warm import timed_us
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

VARIANTS = [
    ("default dispatch", {}),
    ("walking", dict(big_walk_blocks=-1)),                       # (filled in below: whatever the batch)
    ("plain direct, 4/CU", dict(big_walk=0, big_stage=0)),
    ("plain direct, 3/CU", dict(big_walk=0, big_stage=0, big_wg_per_cu=3)),
    ("plain staged, 4/CU", dict(big_walk=0, big_stage=1)),
    ("plain staged, 3/CU", dict(big_walk=0, big_stage=1, big_wg_per_cu=3)),
]


def make(E, opts, shared):
    """An engine under `opts` over the SAME tensors as every other variant (r05: where a tensor lies in memory moves config 5's launch
    by up to 9 % -- tools/placement_probe.py -- so variants must not differ in that)."""
    spec = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=3, dense_prob=0.25)
    opts = dict(opts)
    if opts.get("big_walk_blocks") == -1:
        opts["big_walk_blocks"] = 768
    with N.options(**opts):
        eng = GridEngine(spec, E, device="cuda:0", tensors=shared.get("t"), allocate_obs="obs" not in shared)
    if "obs" not in shared:
        if HUGE:     # the observation tensor inside one 4 GiB allocation: physically contiguous memory, the reproducible (slow) placement
            shared["huge"] = torch.empty((4 << 30) // 4, dtype=torch.float32, device="cuda:0")
            eng.obs = shared["huge"][:eng.obs.numel()].view(eng.obs.shape)
        shared["obs"] = eng.obs
        shared["t"] = dict(grid=eng.grid, agent_pos=eng.agent_pos, actions=eng.actions, rewards=eng.rewards, total_reward=eng.total_reward)
    eng.obs = shared["obs"]
    return eng


HUGE = "--huge" in sys.argv
if "--thresholds" in sys.argv:      # the dispatcher's choice against each forced variant, for the batch sizes around its thresholds
    VARIANTS = [v for v in VARIANTS if v[0] in ("default dispatch", "walking", "plain direct, 4/CU", "plain staged, 4/CU")]


# (Round 6 ran two more A/Bs with this tool -- the XCD-contiguous env assignment and ordinary instead of streaming stores for the staged windows, through options
#  `big_remap` / `big_nt` that existed for the measurement only: profiles/r06_c5_remap_ab.txt.)


def main():
    sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1024, 2048, 4096, 8192]
    print(torch.cuda.get_device_name(0), flush=True)
    for E in sizes:
        shared = {}
        engines = [(name, make(E, o, shared)) for name, o in VARIANTS]
        digs = []
        for name, eng in engines:
            eng.reset(0)
            for _ in range(20):
                eng.step(random_actions=True)
            torch.cuda.synchronize()
            digs.append(hashlib.sha256(eng.obs.cpu().numpy().tobytes() + eng.grid.cpu().numpy().tobytes()
                                       + eng.total_reward.cpu().numpy().tobytes()).hexdigest()[:10])
        best = {name: [] for name, _ in engines}
        for rnd in range(3):
            for name, eng in engines:
                best[name].append(timed_us(lambda: eng.step(random_actions=True), 200, ms=60.0))
        alg = 251984 * E
        for (name, eng), d in zip(engines, digs):
            us = sorted(best[name])
            info = eng.launch_info()
            keep = " ".join(x for x in info.split() if x.startswith(("lds=", "grid=", "wg_per_cu=", "big_stage=")))
            print(f"E={E:6d} {name:22s} {us[0]:7.1f} {us[1]:7.1f} {us[2]:7.1f} us  frac(min)={alg / (us[0] * 1e-6) / 8e12:5.3f}  "
                  f"{'same' if d == digs[0] else 'DIFFERS'}  {info.split(' group')[0]} {keep}", flush=True)
        del engines
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

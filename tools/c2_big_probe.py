#!/usr/bin/env python3
"""BASELINE config 2 (16x16, 4 agents, 5x5, 4 096 envs) and its neighbours on the workgroup-per-env kernel (option force_big) against the
dispatcher's choice: a small batch of small worlds leaves most of the chip idle with a wave per env -- does spreading an env over
four / eight waves (moves by wave 0, windows by all waves) shorten the one wave's chain?  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hashlib
import torch
from _warm import timed_us
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

CASES = [(16, 16, 4, 2, 4096), (16, 16, 4, 2, 1024), (16, 16, 4, 2, 16384), (10, 10, 2, 2, 4096), (21, 21, 2, 2, 4096), (32, 32, 8, 3, 1024), (32, 32, 8, 3, 4096)]
VARIANTS = [("dispatcher", {}), ("wave per env", dict(group=64)), ("step_big 256 threads", dict(force_big=1, big_threads=256)),
            ("step_big 512 threads", dict(force_big=1, big_threads=512)), ("step_big 256, no walk/stage", dict(force_big=1, big_threads=256, big_walk=0, big_stage=0))]
for h, w, a, r, E in CASES:
    spec = treasurehunt_spec(h, w, a, r, spawn_prob=0.005, seed=0)
    ref = None
    for name, opts in VARIANTS:
        with N.options(**opts):
            eng = GridEngine(spec, E, device="cuda:0")
        eng.reset(0)
        for _ in range(20):
            eng.step(random_actions=True)
        torch.cuda.synchronize()
        dig = hashlib.sha256(eng.obs.cpu().numpy().tobytes() + eng.grid.cpu().numpy().tobytes() + eng.total_reward.cpu().numpy().tobytes()).hexdigest()[:10]
        ref = ref or dig
        us = sorted(timed_us(lambda: eng.step(random_actions=True), 2000, ms=60.0) for _ in range(3))
        alg = spec.algorithmic_bytes_per_env_step() * E
        print(f"{h}x{w} A{a} r{r} E={E:6d} {name:28s} {us[0]:6.2f} {us[1]:6.2f} {us[2]:6.2f} us  frac {alg / us[0] / 1e-6 / 8e12:.3f}  {'same' if dig == ref else 'DIFFERS'}  "
              f"{eng.launch_info().split(' threads')[0]}", flush=True)
        eng.close()

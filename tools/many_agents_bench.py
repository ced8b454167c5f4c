#!/usr/bin/env python3
"""Round 6: what a fused turn costs beyond 64 agents (the ticket-ordered generic kernel, step_kernel<256>) next to 64 agents on step_big.  128x128x2 map,
11x11 windows, 2 048 envs; us per turn (HIP events) and the fraction of the 8 TB/s peak on SURVEY 8d's algorithmic bytes.  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from _warm import timed_us
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

for E in (2048, 8192):
    for A in (32, 64, 65, 96, 128):
        ws = treasurehunt_spec(128, 128, A, 5, spawn_prob=0.05, seed=0, dense_prob=0.25)
        eng = GridEngine(ws, E, device="cuda:0")
        eng.reset(0)
        us = min(timed_us(lambda: eng.step(random_actions=True), 100) for _ in range(2))
        alg = ws.algorithmic_bytes_per_env_step() * E
        print(f"128x128x2, {A:3d} agents, {E} envs: {us:8.1f} us per turn = {alg / us / 1e6 / 8.0:5.3f} of 8 TB/s   {us / A * 1000 / E:6.2f} ns per agent-step   {eng.launch_info().split(' group')[0]}", flush=True)
        assert eng.status() == 0
        del eng
        torch.cuda.empty_cache()

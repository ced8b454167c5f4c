#!/usr/bin/env python3
"""Per-launch duration of the first N step launches after the process starts (config 3 by default): shows the
clock / cache ramp that separates a 25-launch run from a 1000-launch one.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
eng = GridEngine(treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=0), 65536, device="cuda:0")
eng.reset(0)
torch.cuda.synchronize()
eng.set_timing(True)
for _ in range(N):
    eng.step(random_actions=True)
ms = eng.step_times_ms()
print(f"{eng.launch_info()}")
print("launches      mean_us   min_us   max_us")
edges = [0, 5, 10, 25, 50, 100, 200, 400, 800, 1600, N]
for lo, hi in zip(edges, edges[1:]):
    seg = ms[lo:hi]
    if seg:
        print(f"{lo:5d}-{hi:5d}  {sum(seg) / len(seg) * 1e3:8.1f} {min(seg) * 1e3:8.1f} {max(seg) * 1e3:8.1f}")

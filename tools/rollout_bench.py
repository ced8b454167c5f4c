#!/usr/bin/env python3
"""sgw_rollout against turn-by-turn sgw_step: us per turn for T-turn rollouts (random actions, observations written every
turn into the engine's own tensor).  usage: tools/rollout_bench.py [H W A r E [T]]   (default: config 3)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

args = [int(v) for v in sys.argv[1:]]
h, w, A, r, E = args[:5] if len(args) >= 5 else (32, 32, 8, 3, 65536)
T = args[5] if len(args) >= 6 else 50
dense = 0.25 if h >= 128 else 0.0
spec = treasurehunt_spec(h, w, A, r, spawn_prob=0.05 if h >= 128 else 0.005, seed=0, dense_prob=dense)
eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
byt = spec.algorithmic_bytes_per_env_step() * E


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000


for _ in range(300): eng.step(random_actions=True)
step_us = timed(lambda: [eng.step(random_actions=True) for _ in range(T)], 6) / T
print(f"{h}x{w} A{A} r{r} E={E}  {eng.launch_info().split(' threads')[0]}")
print(f"  turn by turn (sgw_step)      {step_us:8.1f} us/turn  {E * A / step_us * 1e6:.3e} agent-steps/s  {byt / step_us / 1e3 / 8000:.3f} of 8 TB/s (algorithmic bytes of a turn)")
for t in (2, 5, T):
    us = timed(lambda: eng.rollout(t), max(2, 300 // t)) / t
    print(f"  sgw_rollout, {t:3d} turns/call   {us:8.1f} us/turn  {E * A / us * 1e6:.3e} agent-steps/s  {byt / us / 1e3 / 8000:.3f}")

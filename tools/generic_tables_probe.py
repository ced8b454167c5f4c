"""Worlds whose (layers, channels) have no compile-time tables: us per turn on the wave-per-env kernel, 65 536 envs.  GPU only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import WorldSpec, action_deltas, treasurehunt_spec
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _warm import timed_us


from generic_tables_probe_worlds import move_world

E = 65536
SMALL = (("10x10x2 C5 A2 r2", move_world(10, 10, 2, 5, 2, 2)), ("16x16x2 C8 A4 r2", move_world(16, 16, 2, 8, 4, 2)), ("21x21x2 C8 A2 r2", move_world(21, 21, 2, 8, 2, 2)),
         ("21x21x2 C5 A8 r2", move_world(21, 21, 2, 5, 8, 2)), ("16x16x1 C3 A4 r3", move_world(16, 16, 1, 3, 4, 3)), ("24x24x3 C7 A4 r2", move_world(24, 24, 3, 7, 4, 2)))
for name, spec in SMALL if os.environ.get("PROBE_SMALL") else (("treasurehunt tables 32x32x2 C6 A8 r3", treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005)),
                   ("32x32x2 C8 A8 r3", move_world(32, 32, 2, 8, 8, 3)), ("32x32x2 C5 A8 r3", move_world(32, 32, 2, 5, 8, 3)),
                   ("32x32x1 C4 A8 r3", move_world(32, 32, 1, 4, 8, 3)), ("32x32x3 C10 A8 r3", move_world(32, 32, 3, 10, 8, 3)),
                   ("24x24x2 C8 A6 r4", move_world(24, 24, 2, 8, 6, 4)), ("40x40x2 C12 A8 r2", move_world(40, 40, 2, 12, 8, 2))):
    eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
    for _ in range(100): eng.step(random_actions=True)
    us = timed_us(lambda: eng.step(random_actions=True), 100)
    by = spec.algorithmic_bytes_per_env_step() * E
    print(f"{name:40s} {us:7.1f} us  {by / us / 1e3 / 8000:.2f} of 8 TB/s  {eng.launch_info().split(' threads')[0]}", flush=True)
    del eng
    torch.cuda.empty_cache()

"""Run-time-shape worlds next to the compiled-in shapes: us per turn (sweep + moves + every window) of Treasurehunt-like
worlds whose shape has no static instance, with the bytes-per-turn roofline fraction.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _warm import timed_us

SHAPES = ((32, 32, 8, 3), (32, 33, 8, 3), (24, 24, 8, 3), (40, 40, 8, 3), (20, 20, 4, 4), (30, 30, 8, 4), (30, 26, 7, 5), (16, 16, 4, 2), (21, 21, 2, 2), (10, 10, 2, 2))
E = int(os.environ.get("E", 65536))
if os.environ.get("RT_SHAPES"):   # counter passes: one compile-time and one run-time shape
    SHAPES = ((32, 32, 8, 3), (32, 33, 8, 3)) if os.environ["RT_SHAPES"] == "1" else ((21, 21, 2, 2), (10, 10, 2, 2), (16, 16, 4, 2))
for (h, w, a, r) in SHAPES:
    spec = treasurehunt_spec(h, w, a, r, spawn_prob=0.005, seed=0)
    eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
    for _ in range(200): eng.step(random_actions=True)
    us = timed_us(lambda: eng.step(random_actions=True), 100)
    V = 2 * r + 1
    C = eng.obs.shape[-3] if eng.obs.dim() >= 4 else 0
    by = E * (2 * ((h * w * 2 + 15) // 16 * 16) + eng.obs[0].numel() * 4 + a * 9)
    print(f"{h}x{w} A={a} r={r}: {us:7.1f} us  {by / us / 1e3 / 8000:.2f} of 8 TB/s  {eng.launch_info().split(' threads')[0]}", flush=True)
    del eng
    torch.cuda.empty_cache()

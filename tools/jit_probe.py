"""Specialised instances against the prebuilt ones, and the whole-env burst against chunked bursts: us per turn at 65 536 envs
for worlds that are not the shipped examples (own channel counts, odd maps).  GPU only.
usage: python tools/jit_probe.py [variant ...]   variant = name:key=value;key=value (default: jit=0, burst=0 / 1 / 2)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
from _warm import timed_us
from generic_tables_probe_worlds import move_world

E = int(os.environ.get("E", 65536))
WORLDS = [("th 32x33 A8 r3", treasurehunt_spec(32, 33, 8, 3, spawn_prob=0.005)), ("th 24x24 A8 r3", treasurehunt_spec(24, 24, 8, 3, spawn_prob=0.005)),
          ("th 40x40 A8 r3", treasurehunt_spec(40, 40, 8, 3, spawn_prob=0.005)), ("th 20x20 A4 r4", treasurehunt_spec(20, 20, 4, 4, spawn_prob=0.005)),
          ("th 30x30 A8 r4", treasurehunt_spec(30, 30, 8, 4, spawn_prob=0.005)), ("th 30x26 A7 r5", treasurehunt_spec(30, 26, 7, 5, spawn_prob=0.005)),
          ("32x32x2 C8 A8 r3", move_world(32, 32, 2, 8, 8, 3)), ("32x32x2 C5 A8 r3", move_world(32, 32, 2, 5, 8, 3)),
          ("32x32x1 C4 A8 r3", move_world(32, 32, 1, 4, 8, 3)), ("32x32x3 C10 A8 r3", move_world(32, 32, 3, 10, 8, 3)),
          ("24x24x2 C8 A6 r4", move_world(24, 24, 2, 8, 6, 4)), ("40x40x2 C12 A8 r2", move_world(40, 40, 2, 12, 8, 2))]
variants = [("prebuilt", {"jit": 0}), ("auto", {}), ("whole", {"burst": 1}), ("chunks", {"burst": 2})]
if len(sys.argv) > 1:
    variants = []
    for a in sys.argv[1:]:
        name, _, kv = a.partition(":")
        variants.append((name, dict(x.split("=") for x in kv.split(";") if x)))
print(f"{'world':22s} " + " ".join(f"{n:>16s}" for n, _ in variants))
for name, spec in WORLDS:
    cells = []
    for vname, opts in variants:
        with N.options(**opts):
            eng = GridEngine(spec, E, device="cuda:0")
        eng.reset(0)
        for _ in range(150): eng.step(random_actions=True)
        us = timed_us(lambda: eng.step(random_actions=True), 100)
        by = spec.algorithmic_bytes_per_env_step() * E
        info = eng.launch_info()
        cells.append(f"{us:7.1f} ({by / us / 1e3 / 8000:.2f}){'*' if 'stage_agents=0' in info and 'obs_stage=0' not in info else ' '}")
        del eng
        torch.cuda.empty_cache()
    print(f"{name:22s} " + " ".join(f"{c:>16s}" for c in cells), flush=True)
print("(* = the whole env's windows leave in one burst)")

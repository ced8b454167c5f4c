#!/usr/bin/env python3
"""What sgw_turn_resolve's first pass is made of (config 5's shape): the same pass repeated on the same state (option resolve_diag bit 1:
nothing committed) with the window verification and / or the move resolution switched off.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from _warm import timed_us
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

h, w, A, r, E = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (128, 128, 64, 5, 2048)
spec = treasurehunt_spec(h, w, A, r, spawn_prob=0.05 if h > 64 else 0.005, seed=0, dense_prob=0.25 if h > 64 else 0.0)
eng = GridEngine(spec, E, device="cuda:0")
eng.reset(0)
for _ in range(20):
    eng.step(random_actions=True)
rows = eng.speculation_windows()
fresh = torch.randint(0, 4, (A * E,), device="cuda:0", dtype=torch.int64)
for diag, what in ((2, "whole first pass (nothing committed)"), (3, "... without the window verification"), (7, "... and without the move resolution / touch tests: loads + bookkeeping"),
                   (6, "windows of an empty check mask, no resolution")):
    N.set_option("resolve_diag", diag, engine=eng._h)
    eng.speculation_windows()
    us = timed_us(lambda: eng.turn_resolve(1, None, fresh), 300, ms=40.0)
    print(f"diag {diag}: {us:7.1f} us  {what}", flush=True)
N.set_option("resolve_diag", 2, engine=eng._h)
eng.speculation_windows()
eng.turn_resolve(1, None, fresh)
n1 = int(eng._spec_state[2].sum())
us = timed_us(lambda: eng.turn_resolve(2, None, None), 300, ms=40.0)
print(f"a later pass with nothing changed: {us:7.1f} us   (first pass marked {n1} of {A * E} rows dirty)")

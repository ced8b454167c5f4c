"""sgw_sweep_observe_rows (one launch) against sgw_step(sweep only) + sgw_observe_rows (two), config 3's shape: us per call pair, the
engine's own kernels only.  GPU box.  usage: python tools/sweep_rows_bench.py [envs ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

shape = tuple(int(v) for v in os.environ.get("SHAPE", "32,32,8,3").split(","))
for E in [int(v) for v in sys.argv[1:]] or [16384, 65536]:
    ws = treasurehunt_spec(*shape, spawn_prob=0.005, seed=0)
    eng = GridEngine(ws, E, device="cuda:0", allocate_obs=False) if "allocate_obs" in GridEngine.__init__.__code__.co_varnames else GridEngine(ws, E, device="cuda:0")
    eng.reset(0)
    Nw = int(np.prod(ws.obs_shape[1:]))
    dests = [torch.zeros((E, Nw), device="cuda:0") for _ in range(ws.num_agents)]
    rows = eng.window_rows(dests)
    assert eng.capabilities() & N.CAP_SWEEP_ROWS

    def two(t):
        eng.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t)
        eng.observe_rows(rows)

    def one(t):
        eng.sweep_observe_rows(rows, sweep=True, turn=t)

    for label, fn in (("two launches", two), ("one launch", one), ("two launches", two), ("one launch", one)):
        for t in range(1, 300):
            fn(t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        for t in range(300, 300 + n):
            fn(t)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n * 1e6
        bytes_ = E * (2 * ws.layers * ws.height * ws.width + ws.num_agents * Nw * 4)
        print(f"{shape} E={E:6d} {label:13s} {us:8.1f} us   ({bytes_ / us / 1e6:.2f} TB/s of grid read + write + windows)", flush=True)
    del eng, dests, rows
    torch.cuda.empty_cache()

"""Round 6: sgw_sweep_observe_rows (one launch) against sgw_step(sweep only) + sgw_observe_rows (two) on the kernels that got the fused launch in
round 6 -- step_big (config 5's shape, a Tag world above 4 KiB) and the chunk-staging wave-per-env instances (Cleanup as shipped 21x31x3 with its
12-element positional tail, Tag 11x11 with the "it" flag, a run-time-map Treasurehunt world).  us per turn-start, the engine's kernels only (HIP
events).  GPU box.  usage: python tools/rows_fused_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from _warm import timed_us
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
from tests import helpers as H


def tag(h, w, a, r):
    d, spec = H.load_golden("tag_9x9")
    ws = H.world_spec(spec)
    ws.height, ws.width, ws.num_agents, ws.vision_radius, ws.agent_type = h, w, a, r, [ws.agent_type[0]] * a
    return ws


def cleanup_default():
    d, spec = H.load_golden("cleanup_21x31_default")
    return H.world_spec(spec), d


CASES = [
    ("config 5: 128x128x2, 64 agents, 11x11", lambda: (treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=0, dense_prob=0.25), None), [2048, 8192], None),
    ("Tag 128x128, 64 agents, 9x9, it flag", lambda: (tag(128, 128, 64, 4), None), [2048], "it"),
    ("Cleanup 21x31x3, 10 agents, 11x11, 12-element code", cleanup_default, [16384, 65536], 12),
    ("Tag 11x11, 5 agents, 9x9, it flag", lambda: (tag(11, 11, 5, 4), None), [65536], "it"),
    ("Treasurehunt 33x35 (run-time map), 8 agents, 7x7", lambda: (treasurehunt_spec(33, 35, 8, 3, spawn_prob=0.005, seed=0), None), [65536], None),
]
for name, mk, sizes, tail in CASES:
    ws, d = mk()
    for E in sizes:
        eng = GridEngine(ws, E, device="cuda:0", allocate_obs=False)
        if d is not None and "grid0" in d:       # worlds populated by host code: every env starts from the fixture's grid
            g0, p0 = np.asarray(d["grid0"]), np.asarray(d["pos0"])
            g0, p0 = (g0[0] if g0.ndim == 4 else g0), (p0[0] if p0.ndim == 3 else p0)
            eng.grid.copy_(torch.from_numpy(np.broadcast_to(g0, (E,) + g0.shape).copy()).cuda())
            eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(p0, (E,) + p0.shape).copy()).cuda())
        else:
            eng.reset(0)
        if tail == "it":
            eng.bind_row_tail(N.TAIL_AGENT_IS_IT)
        elif tail:
            eng.bind_row_tail(N.TAIL_POSITION_TABLE, torch.randn((ws.height, ws.width, tail), device="cuda:0"))
        Nr = int(np.prod(ws.obs_shape[1:])) + eng.row_tail
        dests = [torch.zeros((E, Nr), device="cuda:0") for _ in range(ws.num_agents)]
        rows = eng.window_rows(dests)
        caps = eng.capabilities()
        if not (caps & N.CAP_SWEEP_ROWS) or not (caps & N.CAP_OBSERVE_ROWS):
            print(f"{name} E={E}: caps={caps}: skipped ({eng.launch_info()})", flush=True)
            continue
        turn = [0]

        def two():
            turn[0] += 1
            eng.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=turn[0])
            eng.observe_rows(rows)

        def one():
            turn[0] += 1
            eng.sweep_observe_rows(rows, sweep=True, turn=turn[0])

        res = {"two": [], "one": []}
        for _ in range(2):
            res["two"].append(timed_us(two, 100))
            res["one"].append(timed_us(one, 100))
        moved = E * (2 * ws.layers * ws.height * ws.width + ws.num_agents * Nr * 4)
        print(f"{name:58s} E={E:6d}  two launches {min(res['two']):8.1f} us   one launch {min(res['one']):8.1f} us   ({moved / min(res['one']) / 1e6:.2f} TB/s)   "
              f"{eng.launch_info().split('sweep_rows=')[1]}", flush=True)
        del eng, dests, rows
        torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Where a Cleanup sgw_act launch spends its time: per-launch GPU time of the ten acts of a turn with every agent moving /
every agent firing / random actions, windows repaired or not (run on the GPU box).  usage: tools/act_probe.py [E]"""
import os, sys
os.environ["MISC_ONLY"] = "none"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench_misc as BM
from sorrel_amd.engine import GridEngine

E = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
spec = BM.cleanup_spec(21, 31, 10, 5)
eng = GridEngine(spec, E, device="cuda:0")
g = np.zeros((3, 21, 31), np.uint8)
g[:, 0, :] = g[:, -1, :] = 2; g[:, :, 0] = g[:, :, -1] = 2
g[0, 1:7, 1:-1] = 3; g[0, 14:20, 1:-1] = 5; g[0, 7:14, 1:-1] = 1
pos = np.array([[8 + (i // 5) * 2, 3 + (i % 5) * 5] for i in range(10)], np.uint8)
for (y, x) in pos: g[1, y, x] = 11
eng.grid.copy_(torch.from_numpy(np.broadcast_to(g, (E,) + g.shape).copy()))
eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos, (E,) + pos.shape).copy()))
for _ in range(150): eng.step(random_actions=True)
ROWS = eng.window_rows(None)
rnd = eng.actions.clone()


def probe(label, acts, rows):
    eng.actions.copy_(acts)
    eng.set_timing(True)
    for _ in range(10):
        eng.step(eng.actions, sweep=True, no_move=True, advance_turn=False)
        for a in range(10):
            eng.act(a, rows)
        eng.turn += 1
    torch.cuda.synchronize()
    ms = eng.step_times_ms()
    eng.set_timing(False)
    per = [ms[i::11] for i in range(11)]
    mean = lambda v: sum(v) / len(v) * 1000
    print(f"  {label:44s} sweep + windows {mean(per[0]):7.1f} us | act of agent 0 {mean(per[1]):6.1f} us | agent 4 {mean(per[5]):6.1f} | agent 9 {mean(per[10]):6.1f} | ten acts {sum(mean(v) for v in per[1:]):7.1f} us")


print(f"cleanup 21x31x3 A10 r5 E={E}: GPU time per launch (HIP events)")
probe("random actions, windows repaired", rnd, ROWS)
probe("random actions, no windows (rows = None)", rnd, None)
probe("every agent moves up, windows repaired", torch.zeros_like(rnd), ROWS)
probe("every agent fires (zap), windows repaired", torch.full_like(rnd, 5), ROWS)
probe("every agent fires (zap), no windows", torch.full_like(rnd, 5), None)
if os.environ.get("ACT_PROBE_MORE"):
    gen = torch.Generator(device="cpu").manual_seed(0)
    def rand_from(choices):
        idx = torch.randint(0, len(choices), tuple(rnd.shape), generator=gen)
        return torch.tensor(choices, dtype=torch.uint8)[idx].to(rnd.device)
    probe("random moves only (4 directions)", rand_from([0, 1, 2, 3]), ROWS)
    probe("random clean / zap", rand_from([4, 5]), ROWS)
    probe("every agent cleans", torch.full_like(rnd, 4), ROWS)
    probe("half the envs zap, half move up (by env parity)", torch.where((torch.arange(E, device=rnd.device) % 2 == 0)[:, None], torch.full_like(rnd, 5), torch.zeros_like(rnd)), ROWS)
    probe("zap in envs 0..7 of every 16, up in the rest (wave-uniform)", torch.where(((torch.arange(E, device=rnd.device) // 8) % 2 == 0)[:, None], torch.full_like(rnd, 5), torch.zeros_like(rnd)), ROWS)
assert eng.status() == 0

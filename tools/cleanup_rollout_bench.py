#!/usr/bin/env python3
"""sgw_rollout on the Cleanup example's shape (21x31x3, 10 agents, 11x11 windows): us per turn at 50 turns per call
against turn-by-turn sgw_step (run on the GPU box).  usage: tools/cleanup_rollout_bench.py [E]"""
import os, sys
os.environ["MISC_ONLY"] = "none"          # bench_misc runs nothing at import then
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


def timed(fn, reps):
    from _warm import warm
    warm(fn, 80.0, probe=3)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000


for _ in range(100): eng.step(random_actions=True)
print(f"cleanup 21x31x3 A10 r5 E={E}  {eng.launch_info().split(' threads')[0]}")
print(f"  turn by turn (sgw_step)   {timed(lambda: [eng.step(random_actions=True) for _ in range(50)], 3) / 50:8.1f} us/turn")
print(f"  sgw_rollout, 50 turns     {timed(lambda: eng.rollout(50), 3) / 50:8.1f} us/turn")


def phased_turn():          # what a policy-driven Environment.take_turn launches: sweep + pov of agent 0, then one launch per agent
    eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, obs_next=True, advance_turn=False)
    for a in range(10):
        eng.step(eng.actions, sweep=False, write_obs=False, agent_begin=a, agent_end=a + 1, obs_next=a < 9, advance_turn=False)
    eng.turn += 1


print(f"  policy-driven, 1 + A launches (round 2: a window per launch) {timed(lambda: [phased_turn() for _ in range(20)], 3) / 20:8.1f} us/turn")

ROWS = eng.window_rows(None)


def patched_turn():         # round 3: sweep + every window in ONE launch, then per agent sgw_act (CleanupAgent.act + repair of later windows)
    eng.step(eng.actions, sweep=True, no_move=True, advance_turn=False)
    for a in range(10):
        eng.act(a, ROWS)
    eng.turn += 1


print(f"  policy-driven, 1 + A launches (round 3: NO_MOVE + sgw_act)   {timed(lambda: [patched_turn() for _ in range(20)], 3) / 20:8.1f} us/turn")
assert eng.status() == 0

# round 4: every agent's window + its positional code into a row of its own (what its replay buffer row is) -- sweep alone, observe_rows
# with the row tail bound, then per agent sgw_act; no torch.cat of the window on the host
import torch as _t
from sorrel_amd import _native as N
if eng.capabilities() & N.CAP_OBSERVE_ROWS:
    table = _t.rand((21, 31, 12), device="cuda:0")
    eng.bind_row_tail(N.TAIL_POSITION_TABLE, table)
    nwin = int(np.prod(spec.obs_shape[1:]))
    dests = [_t.zeros((E, nwin + 12), device="cuda:0") for _ in range(10)]
    TROWS = eng.window_rows(dests)

    def tailed_turn():
        eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, advance_turn=False)
        eng.observe_rows(TROWS)
        for a in range(10):
            eng.act(a, TROWS)
        eng.turn += 1

    def parts():
        t_sweep = timed(lambda: eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, advance_turn=False), 20)
        t_rows = timed(lambda: eng.observe_rows(TROWS), 20)
        return t_sweep, t_rows

    print(f"  policy-driven, 2 + A launches (round 4: sweep, observe_rows into tailed rows, sgw_act) {timed(lambda: [tailed_turn() for _ in range(20)], 3) / 20:8.1f} us/turn"
          f"   (sweep alone {parts()[0]:.1f} us, observe_rows {parts()[1]:.1f} us)")
    assert eng.status() == 0

    # round 4, "lazy windows": the sweep alone, then per agent its OWN window rendered right before its policy reads it (observe_rows of
    # one agent: what earlier agents' beams and moves did is simply in the grid by then) and its act WITHOUT repairs.  2 A + 1 launches;
    # pays where an act changes many cells that many later windows contain (Cleanup's beams: 3 R cells x ~5 later windows x 2 channels
    # of lone 4-byte stores per env and act)
    def lazy_turn():
        eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, advance_turn=False)
        for a in range(10):
            eng.observe_rows(TROWS, a, a + 1)
            eng.act(a, None)
        eng.turn += 1

    print(f"  policy-driven, 1 + 2 A launches (round 4: sweep, then per agent observe_rows of ONE agent + sgw_act without repairs) {timed(lambda: [lazy_turn() for _ in range(20)], 3) / 20:8.1f} us/turn"
          f"   (one agent's window {timed(lambda: eng.observe_rows(TROWS, 3, 4), 20):.1f} us, an act without repairs {timed(lambda: eng.act(3, None), 20):.1f} us)")
    assert eng.status() == 0

#!/usr/bin/env python3
"""Cost of a POLICY-driven turn (the path any trained model takes): the entity sweep, then per agent observe -> (policy)
-> act, agent i+1 observing agent i's move.  Times the engine launches only (actions are precomputed; a real policy's
forward pass comes on top), config 3 by default, three ways:
  fused    one launch (random / given actions: what bench.py measures)
  1 + 2A   round 1: sweep; per agent sgw_observe + sgw_step
  1 + A    round 2: sweep + obs of agent 0; per agent ONE sgw_step that moves it and renders the next agent (OBS_NEXT)
  2 + A    round 3: sweep; every agent's window once (sgw_observe_rows); per agent sgw_act = move + repair of later windows
Run on the GPU box: python tools/phased_bench.py [H W A r E]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

h, w, A, r, E = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 32, 8, 3, 65536)
spec = treasurehunt_spec(h, w, A, r, spawn_prob=0.005, seed=0)
eng = GridEngine(spec, E, device="cuda:0")
eng.reset(0)
acts = eng.random_actions(turn=1)      # = eng.actions: the step consumes it in place (a policy writes its choices there; passing
K = 100                                # another tensor would add a copy kernel to every launch timed below)


def timed(fn, warm=30):
    from _warm import warm as warm_ms        # ~80 ms of uninterrupted launches first: a few ms of idle bring the chip's power ramp back
    warm_ms(fn, 80.0, probe=max(3, warm // 6))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / K * 1000


def fused():
    eng.step(acts)


def old():
    eng.turn += 1
    eng.step(acts, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=eng.turn)
    for a in range(A):
        eng.observe(a, a + 1)
        eng.step(acts, sweep=False, write_obs=False, agent_begin=a, agent_end=a + 1, turn=eng.turn)


def new():
    eng.turn += 1
    eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=eng.turn)
    for a in range(A):
        eng.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, obs_next=a + 1 < A, write_obs=False, turn=eng.turn)


ROWS = eng.window_rows(None)


def patched():
    eng.turn += 1
    eng.step(acts, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=eng.turn)
    eng.observe_rows(ROWS)
    for a in range(A):
        eng.act(a, ROWS)


def patched_tensor():
    eng.turn += 1
    eng.step(acts, sweep=True, no_move=True, turn=eng.turn)      # sweep + every window (into the tensor) in one launch
    for a in range(A):
        eng.act(a, ROWS)


print(f"{h}x{w} A{A} r{r} E={E}  {eng.launch_info().split(' threads')[0]}")
us = timed(patched_tensor)
print(f"  {f'1 + A = {1 + A} launches (round 3, NO_MOVE + sgw_act)':40s} {us:9.1f} us/turn  {E * A / us * 1e6:.3e} agent-steps/s  ({us / (1 + A):6.1f} us per launch)")
us = timed(patched)
print(f"  {f'2 + A = {2 + A} launches (round 3, sgw_act)':40s} {us:9.1f} us/turn  {E * A / us * 1e6:.3e} agent-steps/s  ({us / (2 + A):6.1f} us per launch)")
eng.set_timing(True)
for _ in range(20):
    patched()
torch.cuda.synchronize()
ms = eng.step_times_ms()
eng.set_timing(False)
per = [ms[i::2 + A] for i in range(2 + A)]
mean = lambda v: sum(v) / len(v) * 1000
acts_ms = [x for k in range(2, 2 + A) for x in per[k]]
print(f"  GPU time per launch (events): sweep {mean(per[0]):6.1f} us | all windows {mean(per[1]):6.1f} us | sgw_act {mean(acts_ms):6.1f} us | "
      f"sum per turn {sum(mean(v) for v in per):7.1f} us")
for name, fn, launches in (("fused (1 launch)", fused, 1), (f"1 + 2A = {1 + 2 * A} launches (round 1)", old, 1 + 2 * A),
                           (f"1 + A = {1 + A} launches (round 2, OBS_NEXT)", new, 1 + A)):
    us = timed(fn)
    print(f"  {name:40s} {us:9.1f} us/turn  {E * A / us * 1e6:.3e} agent-steps/s  ({us / launches:6.1f} us per launch)")

# per-launch GPU durations (HIP events around every launch) of the 1 + A form: the sweep launch, the phases that move an
# agent and render the next one, the last phase (moves only)
eng.set_timing(True)
for _ in range(20):
    new()
torch.cuda.synchronize()
ms = eng.step_times_ms()
eng.set_timing(False)
per = [ms[i::1 + A] for i in range(1 + A)]
mean = lambda v: sum(v) / len(v) * 1000
mid = [x for k in range(1, A) for x in per[k]]
print(f"  GPU time per launch (events): sweep + window 0 {mean(per[0]):6.1f} us | move + next window {mean(mid) if mid else 0.0:6.1f} us | "
      f"last move {mean(per[A]):6.1f} us | sum per turn {sum(mean(v) for v in per):7.1f} us")

"""BASELINE config 2 (16x16, 4 agents, 5x5 windows, 4 096 envs): what its ~10 us launch is made of.  HIP-event times of back-to-back
launches of (a) every kernel family that can serve it, (b) the same engine with parts of the turn switched off through the ABI's own
flags (no observations, no sweep, observe only, sweep only) and (c) an empty kernel of the same grid (the launch floor: what a
dependent launch costs on this stack whatever it does).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
from _warm import timed_us

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.005, seed=0)
by = spec.algorithmic_bytes_per_env_step() * E


def run(label, opts, fn=None):
    with N.options(**opts):
        eng = GridEngine(spec, E, device="cuda:0")
    eng.reset(0)
    call = (lambda: eng.step(random_actions=True)) if fn is None else (lambda: fn(eng))
    for _ in range(3000):
        call()
    us = min(timed_us(call, 500) for _ in range(3))
    print(f"{label:58s} {us:6.2f} us  {by / us / 1e3 / 8000:.3f} of 8 TB/s  {eng.launch_info().split(' threads')[0]}", flush=True)
    eng.close()
    return us


print(f"config 2: {E} envs, {by / 1e6:.1f} MB algorithmic per launch ({by / 8e6:.2f} us at 8 TB/s)")
run("whole turn, the dispatcher's choice", {})
run("whole turn, two envs per wave (packed, specialised)", {"group": 32})
run("whole turn, four envs per wave", {"group": 16})
run("whole turn, a wave per env (step_fast, static 16x16)", {"group": 64})
run("whole turn, a wave per env, prebuilt", {"group": 64, "jit": 0})
full = run("  ... its parts: whole turn again", {"group": 64})
run("  no observations (sweep + moves + grid write-back)", {"group": 64}, lambda e: e.step(random_actions=True, write_obs=False))
run("  no sweep (moves + windows)", {"group": 64}, lambda e: e.step(random_actions=True, sweep=False))
run("  observe only (sgw_observe: windows, nothing moves)", {"group": 64}, lambda e: e.observe())
run("  sweep only (no agents, no windows)", {"group": 64}, lambda e: e.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, advance_turn=False, turn=5))
run("  one agent's act (sgw_act: the floor of a dependent launch)", {"group": 64}, lambda e: e.act(0, None))

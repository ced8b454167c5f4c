#!/usr/bin/env python3
"""Diagnostic throughput of the shapes that run on the generic / Tag paths (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import numpy as np, torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import WorldSpec, treasurehunt_spec, action_deltas
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _warm import timed_us

def tag_spec(h, w, a, r):
    app = np.zeros((4, 4)); app[1, 1] = app[2, 2] = app[3, 3] = 1.0
    dy, dx = action_deltas(["up", "down", "left", "right"])
    return WorldSpec(height=h, width=w, layers=1, num_agents=a, vision_radius=r, num_channels=4, agent_layer=0,
                     default_type=0, fill_type=1, action_dy=dy, action_dx=dx, agent_type=[3] * a,
                     type_value=[0, -1, 0, 0], type_passable=[1, 0, 0, 0], type_rule=[0] * 4, spawn_prob=[0.0] * 4,
                     spawn_choices=[[]] * 4, appearance=app, seed=1, layer_fill_type=[0], layer_border_type=[1],
                     agent_rule=1, tag_it_type=2, tag_notit_type=3, tag_reward=10.0)

KW = dict(write_obs=os.environ.get("MISC_NO_OBS") != "1", sweep=os.environ.get("MISC_NO_SWEEP") != "1")   # diagnostic ablations
if os.environ.get("MISC_AGENTS"):
    KW["agent_end"] = int(os.environ["MISC_AGENTS"])
ONLY = os.environ.get("MISC_ONLY", "")


def run(name, spec, E, K=100):
    if ONLY and ONLY not in name:
        return
    eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
    for _ in range(200): eng.step(random_actions=True, **KW)
    us = timed_us(lambda: eng.step(random_actions=True, **KW), K)
    byt = spec.algorithmic_bytes_per_env_step() * E
    print(f"{name:34s} E={E:7d} {us:8.1f} us/step  {E*spec.num_agents/us*1e6:.3e} agent-steps/s  {byt/us/1e3:7.1f} GB/s ({byt/us/1e3/8000:.3f} of 8 TB/s)")

run("tag 11x11x1 A5 r4 (example default)", tag_spec(11, 11, 5, 4), 65536)
run("tag 32x32x1 A8 r3", tag_spec(32, 32, 8, 3), 65536)
run("treasurehunt 10x10x2 A2 r2 (ragged)", treasurehunt_spec(10, 10, 2, 2), 65536)
run("treasurehunt 21x21x2 A2 r2 (default)", treasurehunt_spec(21, 21, 2, 2), 65536)
run("treasurehunt 32x32x2 A8 r3 (fast)", treasurehunt_spec(32, 32, 8, 3), 65536)


def run_observe(name, spec, E, K=100):
    """K1 alone: sgw_observe of all agents (grid read + observation stores, no sweep, no moves)."""
    if ONLY and ONLY not in name:
        return
    eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
    us = timed_us(lambda: eng.observe(), K)
    byt = (spec.grid_bytes_per_env() + spec.num_agents * spec.num_channels * spec.window ** 2 * 4) * E
    print(f"{name:34s} E={E:7d} {us:8.1f} us/call  {byt/us/1e3:7.1f} GB/s")


def cleanup_spec(h, w, a, r):
    """The Cleanup tables of tests/golden/cleanup_15x16 at another size (types as in oracle/make_golden.py)."""
    import json
    d = np.load("tests/golden/cleanup_15x16.npz", allow_pickle=True)
    s = json.loads(str(d["spec_json"]))
    s.update(height=h, width=w, num_agents=a, vision_radius=r, agent_type=[11] * a, spawn_prob=[0, 0, 0, 0.009, 0, 0.002] + [0] * 6)
    names = WorldSpec.__dataclass_fields__.keys()
    kw = {k: v for k, v in s.items() if k in names}
    kw["appearance"] = np.asarray(s["appearance"], dtype=np.float64)
    kw["spawn_choices"] = s["spawn_choices"]
    return WorldSpec(**kw)


run_observe("observe only 32x32x2 A8 r3", treasurehunt_spec(32, 32, 8, 3), 65536)


def run_cleanup(E=16384, K=50):
    if ONLY and ONLY not in "cleanup":
        return
    spec = cleanup_spec(21, 31, 10, 5)
    eng = GridEngine(spec, E, device="cuda:0")
    # the reference's map: walls around, river on top, orchard at the bottom, agents on the sand in between
    g = np.zeros((3, 21, 31), np.uint8)
    g[:, 0, :] = g[:, -1, :] = 2; g[:, :, 0] = g[:, :, -1] = 2
    g[0, 1:7, 1:-1] = 3; g[0, 14:20, 1:-1] = 5; g[0, 7:14, 1:-1] = 1
    pos = np.array([[8 + (i // 5) * 2, 3 + (i % 5) * 5] for i in range(10)], np.uint8)
    for (y, x) in pos: g[1, y, x] = 11
    eng.grid.copy_(torch.from_numpy(np.broadcast_to(g, (E,) + g.shape).copy()))
    eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos, (E,) + pos.shape).copy()))
    for _ in range(100): eng.step(random_actions=True, **KW)
    us = timed_us(lambda: eng.step(random_actions=True, **KW), K)
    byt = spec.algorithmic_bytes_per_env_step() * E
    print(f"{'cleanup 21x31x3 A10 r5 (RULES kernel)':34s} E={E:7d} {us:8.1f} us/step  {E*10/us*1e6:.3e} agent-steps/s  {byt/us/1e3:7.1f} GB/s ({byt/us/1e3/8000:.3f} of 8 TB/s)  [{eng.launch_info().split(' group')[0]}]")
    assert eng.status() == 0


run_cleanup(int(os.environ.get('CLEANUP_E', '16384')))


def run_big_rule_worlds():
    """Tag and Cleanup worlds above 4 KiB (step_kernel<256>): only with MISC_ONLY=big."""
    if "big" not in ONLY:
        return
    for name, spec, E in (("big tag 72x72x1 A16 r4", tag_spec(72, 72, 16, 4), 8192),
                          ("big tag 128x128x1 A64 r4", tag_spec(128, 128, 64, 4), 2048)):
        eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
        for _ in range(100): eng.step(random_actions=True, **KW)
        us = timed_us(lambda: eng.step(random_actions=True, **KW), 100)
        byt = spec.algorithmic_bytes_per_env_step() * E
        print(f"{name:34s} E={E:7d} {us:8.1f} us/step  {E*spec.num_agents/us*1e6:.3e} agent-steps/s  {byt/us/1e3:7.1f} GB/s ({byt/us/1e3/8000:.3f} of 8 TB/s)  [{eng.launch_info().split(' group')[0]}]")
    spec = cleanup_spec(48, 48, 10, 5)
    E = 4096
    eng = GridEngine(spec, E, device="cuda:0")
    g = np.zeros((3, 48, 48), np.uint8)
    g[:, 0, :] = g[:, -1, :] = 2; g[:, :, 0] = g[:, :, -1] = 2
    g[0, 1:14, 1:-1] = 3; g[0, 34:47, 1:-1] = 5; g[0, 14:34, 1:-1] = 1
    pos = np.array([[18 + (i // 5) * 8, 5 + (i % 5) * 9] for i in range(10)], np.uint8)
    for (y, x) in pos: g[1, y, x] = 11
    eng.grid.copy_(torch.from_numpy(np.broadcast_to(g, (E,) + g.shape).copy()))
    eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos, (E,) + pos.shape).copy()))
    for _ in range(100): eng.step(random_actions=True, **KW)
    us = timed_us(lambda: eng.step(random_actions=True, **KW), 100)
    byt = spec.algorithmic_bytes_per_env_step() * E
    print(f"{'big cleanup 48x48x3 A10 r5':34s} E={E:7d} {us:8.1f} us/step  {E*10/us*1e6:.3e} agent-steps/s  {byt/us/1e3:7.1f} GB/s ({byt/us/1e3/8000:.3f} of 8 TB/s)  [{eng.launch_info().split(' group')[0]}]")
    assert eng.status() == 0


run_big_rule_worlds()

"""A/B in ONE process (same tensors): the eager policy turn through the generic Agent.transition loop and through
Environment.fast_policy_loop, alternating.  GPU box.  usage: python tools/fast_loop_ab.py <h> <w> <agents> <radius> <envs> [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import latency_bench as LB
from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
from sorrel_amd.examples.treasurehunt.main import make_config
from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld

h, w, a, r, E = (int(v) for v in sys.argv[1:6])
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 3
cfg = make_config(h, w, a, r, spawn_prob=0.005)
env = TreasurehuntEnv(TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=0), cfg, model_factory=LB.policy_factory(E))
turns = 2000 if E <= 4096 else 300
for k in range(rounds):
    for fast, fuse in ((False, False), (True, False), (True, True)):
        env.fast_policy_loop, env.fuse_sweep_and_rows = fast, fuse
        us = LB.time_turns(env, turns)
        # host time alone: the same loop without waiting for the device at the end is what the host needs to issue a turn
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(turns):
            env.take_turn()
        issue = (time.perf_counter() - t0) / turns * 1e6
        torch.cuda.synchronize()
        print(f"{h}x{w} A{a} r{r} E={E:6d} {('fast loop, sweep + rows in one launch' if fuse else 'fast loop, two launches            ') if fast else 'generic loop, two launches         '} {us:8.1f} us/turn   issued in {issue:8.1f} us/turn", flush=True)
env.raise_on_status()

#!/usr/bin/env python3
"""cProfile of the host side of a policy-driven Environment.take_turn() (run on the GPU box): where the ~80 us per agent
phase go when the batch is small enough for the GPU work not to matter."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.chdir(ROOT)
import torch
import latency_bench as LB  # noqa: F401  (defines policy_factory; its module-level sweep is skipped below)
from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
from sorrel_amd.examples.treasurehunt.main import make_config
from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld

E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = make_config(32, 32, 8, 3, spawn_prob=0.005)
world = TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=0)
env = TreasurehuntEnv(world, cfg, model_factory=LB.policy_factory(E))
for _ in range(50):
    env.take_turn()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    env.take_turn()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

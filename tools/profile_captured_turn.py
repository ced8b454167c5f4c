"""Kernel trace of a captured policy turn (run under rocprofv3 --kernel-trace): what one replay launches.  GPU only.
usage: python tools/profile_captured_turn.py [E] [h w a r] [values]      (values: the policies hand their action values to the act launch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from latency_bench import policy_factory, time_turns
from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
from sorrel_amd.examples.treasurehunt.main import make_config
from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h, w, a, r = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 and sys.argv[2].isdigit() else (32, 32, 8, 3)
cfg = make_config(h, w, a, r, spawn_prob=0.005)
world = TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=0)
env = TreasurehuntEnv(world, cfg, model_factory=policy_factory(E, values="values" in sys.argv))
assert env.capture_turn() is not None, env.capture_error
print(f"{h}x{w} A{a} r{r} E={E}: {time_turns(env, 200):.1f} us per replayed turn")

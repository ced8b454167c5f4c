"""cProfile of the eager policy-driven Environment.take_turn() (host side): where the ~45 us per agent go.  GPU box.
usage: python tools/host_profile.py [envs] [turns]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import latency_bench as LB
from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
from sorrel_amd.examples.treasurehunt.main import make_config
from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cfg = make_config(32, 32, 8, 3, spawn_prob=0.005)
env = TreasurehuntEnv(TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=0), cfg, model_factory=LB.policy_factory(E))
for _ in range(50):
    env.take_turn()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(T):
    env.take_turn()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

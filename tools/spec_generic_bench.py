#!/usr/bin/env python3
"""Round 6: wall time per Environment.take_turn() of the generic speculative turn (sgw_verify_rows: Tag, Cleanup, > 64 agents) against the eager
agent-after-agent loop -- the examples as shipped and variants with more agents; ONE linear policy and one replay ring shared by every agent.
GPU box.  usage: tools/spec_generic_bench.py [envs ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sorrel_amd.buffers import Buffer
from sorrel_amd.models import BaseModel


def make(which, E, speculate):
    one = []

    class Shared(BaseModel):
        def __init__(self, input_size, n_actions, A):
            n = int(np.prod(input_size))
            super().__init__((n,), n_actions, memory_size=0, num_envs=E, device="cuda:0")
            self.memory = Buffer(capacity=4 * A, obs_shape=(n,), num_envs=E, device="cuda:0")
            self.weight = torch.randn((n, n_actions), generator=torch.Generator().manual_seed(3)).cuda()

        def take_action(self, state):
            return (state.reshape(state.shape[0], -1) @ self.weight).argmax(dim=1)

    name, A = which
    def factory(input_size, n_actions):
        if not one:
            one.append(Shared(input_size, n_actions, A))
        return one[0]
    if name == "tag":
        from sorrel_amd.entities import EmptyEntity
        from sorrel_amd.examples.tag.env import TagEnv
        from sorrel_amd.worlds import Gridworld
        size = 11 if A <= 5 else 32
        cfg = {"agent": {"num_agents": A, "vision_radius": 4 if A <= 5 else 3, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 100}}
        env = TagEnv(Gridworld(size, size, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=31), cfg, model_factory=factory)
    elif name == "cleanup":
        from sorrel_amd.examples.cleanup.entities import EmptyEntity as CEmpty
        from sorrel_amd.examples.cleanup.env import CleanupEnv
        from sorrel_amd.examples.cleanup.main import make_config as cleanup_config
        from sorrel_amd.examples.cleanup.world import CleanupWorld
        cfg = cleanup_config(num_agents=A)
        env = CleanupEnv(CleanupWorld(cfg, CEmpty(), num_envs=E, device="cuda:0", seed=41), cfg, model_factory=factory)
    else:
        from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
        from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
        from sorrel_amd.examples.treasurehunt.main import make_config
        from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld
        cfg = make_config(64, 64, A, 3, spawn_prob=0.02)
        env = TreasurehuntEnv(TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=0), cfg, model_factory=factory)
    env.speculate_turns = "always" if speculate else False
    return env


def wall(env, turns=150):
    for _ in range(30):
        env.take_turn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(turns):
        env.take_turn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / turns * 1e6


for E in [int(v) for v in sys.argv[1:]] or [1024, 4096]:
    for which in (("tag", 5), ("tag", 16), ("cleanup", 10), ("plain", 96)):
        try:
            res = {}
            for speculate in (False, True):
                env = make(which, E, speculate)
                res[speculate] = wall(env)
                passes = getattr(env, "speculation_passes", None)
                loop = env.turn_plan()["loop"]
                del env
                torch.cuda.empty_cache()
            print(f"{which[0]:8s} {which[1]:3d} agents  E={E:6d}   eager {res[False]:8.1f} us   speculative (generic) {res[True]:8.1f} us   passes of the last turn {passes}   [{loop}]", flush=True)
        except Exception as exc:
            print(f"{which} E={E}: {exc!r}"[:300], flush=True)

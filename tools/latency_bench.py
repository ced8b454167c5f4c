#!/usr/bin/env python3
"""What a user who switches from the reference feels: wall time of ``Environment.take_turn()`` through the Python API
(run on the GPU box).  Treasurehunt example defaults (21x21, 2 agents, 5x5 window) and the headline shape, batches of
1 ... 65 536 envs; the device-random model (one launch per turn) and a small torch policy (1 + A launches + A forward
passes per turn, memories appended).  The reference's own step loop does one env at ~3 ms per turn on one core."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch
from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
from sorrel_amd.examples.treasurehunt.main import make_config
from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld
from sorrel_amd.models import BaseModel


def policy_factory(E, values=False):
    class LinearPolicy(BaseModel):
        """obs [E, F] -> argmax of one linear layer: the cheapest model that really reads the observation."""

        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=64, num_envs=E, device="cuda:0")
            g = torch.Generator(device="cpu").manual_seed(1)
            self.w = torch.randn(int(input_size[0]), action_space, generator=g).cuda()

            self.epsilon = 0.05 if values else 0.0

        def take_action(self, state):
            q = state.reshape(state.shape[0], -1) @ self.w
            return q if values else q.argmax(dim=1)     # values: the act launch takes the argmax / explores (SGW_ACT_QF32)

    return LinearPolicy


def time_turns(env, turns):
    for _ in range(20):
        env.take_turn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(turns):
        env.take_turn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / turns * 1e6


def run(h, w, a, r, E, policy, spawn_prob=0.005):
    cfg = make_config(h, w, a, r, spawn_prob=spawn_prob)
    world = TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=0)
    env = TreasurehuntEnv(world, cfg, model_factory=policy_factory(E, values=policy >= 3) if policy else None)
    env.write_obs_into_replay = os.environ.get("LAT_NO_DIRECT") != "1"      # A/B: windows through the observation tensor + a copy
    label = {0: "device-random (1 launch)", 1: "policy (1+A launches)", 2: "policy, captured turn", 3: "values -> act, eager", 4: "values -> act, captured"}[policy]
    if policy in (2, 4) and env.capture_turn() is None:                          # round 4: the whole turn recorded once, replayed
        label = f"policy, NOT capturable: {getattr(env, 'capture_error', None)!r}"[:60]
    turns = 2000 if E <= 4096 else 300
    us = time_turns(env, turns)
    env.raise_on_status()
    print(f"{h}x{w} A{a} r{r} E={E:6d} {label:26s} {us:9.1f} us/turn  {E * a / us * 1e6:.3e} agent-steps/s", flush=True)
    del env, world
    torch.cuda.empty_cache()


def main():
    for shape in ((21, 21, 2, 2), (32, 32, 8, 3)):
        for E in (1, 64, 1024, 16384, 65536):
            for policy in (0, 1, 2, 3, 4):
                run(*shape, E, policy)
    for policy in (0, 1, 2, 3, 4):       # BASELINE config 5's per-GPU share
        run(128, 128, 64, 5, 2048, policy, spawn_prob=0.05)


if __name__ == "__main__":
    main()

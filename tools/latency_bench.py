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
    # warm up by TIME: a chip that idled between two processes runs ~25 % slower for its first ~150 ms of work (32x32 / 8 agents at 65 536 envs:
    # 560-600 us per turn over the first 300 turns of a process, 435-450 once warm) -- the same ramp bench.py's prewarm launches are for
    t0 = time.perf_counter()
    n = 0
    while n < 20 or time.perf_counter() - t0 < float(os.environ.get("LAT_WARM_S", "0.5")):
        env.take_turn()
        n += 1
        if n % 50 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(turns):
        env.take_turn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / turns * 1e6


def run(h, w, a, r, E, policy, spawn_prob=0.005):
    cfg = make_config(h, w, a, r, spawn_prob=spawn_prob)
    world = TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=0)
    factory = policy_factory(E, values=policy in (3, 4)) if policy else None
    if policy in (5, 6):                  # ONE model object (and one replay ring of A rows per turn) shared by every agent
        cls, one = factory, [None]

        class Shared(cls):
            def __init__(self, input_size, action_space):
                super().__init__(input_size, action_space)
                from sorrel_amd.buffers import Buffer
                self.memory = Buffer(capacity=4 * a, obs_shape=tuple(input_size), num_envs=E, device="cuda:0")

        def factory(input_size, action_space):
            if one[0] is None:
                one[0] = Shared(input_size, action_space)
            return one[0]
    env = TreasurehuntEnv(world, cfg, model_factory=factory)
    env.speculate_turns = "always" if policy in (6, 7) else False    # (whatever the cost model says: this tool measures it)
    # round 5: batched policy evaluation + sgw_turn_resolve (Environment.speculate_turns)
    env.write_obs_into_replay = os.environ.get("LAT_NO_DIRECT") != "1"      # A/B: windows through the observation tensor + a copy
    env.fast_policy_loop = os.environ.get("LAT_GENERIC_LOOP") != "1"        # A/B: the generic Agent.transition loop (round 4's)
    env.fuse_sweep_and_rows = os.environ.get("LAT_NO_FUSE") != "1"           # A/B: the sweep alone + sgw_observe_rows (two launches)
    label = {0: "device-random (1 launch)", 1: "policy (1+A launches)", 2: "policy, captured turn", 3: "values -> act, eager", 4: "values -> act, captured",
             5: "shared policy, eager", 6: "shared policy, speculative", 7: "own policies, speculative"}[policy]
    if policy in (2, 4) and env.capture_turn() is None:                          # round 4: the whole turn recorded once, replayed
        label = f"policy, NOT capturable: {getattr(env, 'capture_error', None)!r}"[:60]
    turns = int(os.environ.get("LAT_TURNS", 2000 if E <= 4096 else 300))
    us = time_turns(env, turns)
    env.raise_on_status()
    extra = ""
    if policy in (6, 7):
        extra = f"  passes of the last turn: {env.speculation_passes}" if hasattr(env, "speculation_passes") else "  (a model per agent: Environment keeps the sequential turn)"
    print(f"{h}x{w} A{a} r{r} E={E:6d} {label:26s} {us:9.1f} us/turn  {E * a / us * 1e6:.3e} agent-steps/s{extra}", flush=True)
    del env, world
    torch.cuda.empty_cache()


def run_example(which, E, policy):
    """The shipped Tag (8x9... as configured below: 11x11 world, 5 agents, 9x9 windows + the "it" flag) and Cleanup (21x31x3, 10 agents, 11x11
    windows + positional code) examples with a one-layer policy over the finished row."""
    import numpy as np

    class Linear(BaseModel):
        def __init__(self, input_size, action_space):
            n = int(np.prod(input_size))
            super().__init__((n,), action_space, memory_size=64, num_envs=E, device="cuda:0")
            self.w = torch.randn(n, action_space, generator=torch.Generator(device="cpu").manual_seed(1)).cuda()

        def take_action(self, state):
            return (state.reshape(state.shape[0], -1) @ self.w).argmax(dim=1)

    if which == "tag":
        from sorrel_amd.entities import EmptyEntity as E0
        from sorrel_amd.examples.tag.env import TagEnv
        from sorrel_amd.worlds import Gridworld
        cfg = {"agent": {"num_agents": 5, "vision_radius": 4, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 100}}
        env = TagEnv(Gridworld(11, 11, 1, E0(), num_envs=E, device="cuda:0", seed=0), cfg, model_factory=Linear)
    else:
        sys.path.insert(0, ROOT)
        from tests.test_api_host import CLEANUP_CFG
        from sorrel_amd.examples.cleanup.entities import EmptyEntity as CEmpty
        from sorrel_amd.examples.cleanup.env import CleanupEnv
        from sorrel_amd.examples.cleanup.world import CleanupWorld
        env = CleanupEnv(CleanupWorld(CLEANUP_CFG, CEmpty(), num_envs=E, device="cuda:0", seed=0), CLEANUP_CFG, model_factory=Linear)
    label = "policy (eager)" if policy == 1 else "policy, captured turn"
    if policy == 2 and env.capture_turn() is None:
        label = f"NOT capturable: {getattr(env, 'capture_error', None)!r}"[:60]
    us = time_turns(env, 1000 if E <= 4096 else 200)
    env.raise_on_status()
    A = len(env.agents)
    print(f"{which:8s} A{A} E={E:6d} {label:26s} {us:9.1f} us/turn  {E * A / us * 1e6:.3e} agent-steps/s", flush=True)
    del env
    torch.cuda.empty_cache()


def main():
    if len(sys.argv) > 4 and sys.argv[1] == "examples":         # one case (for a kernel trace): examples <tag|cleanup> <envs> <1|2>
        run_example(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "examples":
        for which in ("tag", "cleanup"):
            for E in (1, 1024, 16384):
                for policy in (1, 2):
                    run_example(which, E, policy)
        return
    if len(sys.argv) > 7 and sys.argv[1] == "one":              # one case (for a kernel trace): one <h> <w> <agents> <radius> <envs> <policy>
        h, w, a, r, E, policy = (int(v) for v in sys.argv[2:8])
        run(h, w, a, r, E, policy, spawn_prob=0.05 if h > 64 else 0.005)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "spec":             # the speculative policy turn beside the eager / recorded ones
        for policy in (1, 2, 5, 6, 7):
            run(128, 128, 64, 5, 2048, policy, spawn_prob=0.05)
        for E in (1024, 16384, 65536):
            for policy in (1, 5, 6, 7):
                run(32, 32, 8, 3, E, policy)
        return
    for shape in ((21, 21, 2, 2), (32, 32, 8, 3)):
        for E in (1, 64, 1024, 16384, 65536):
            for policy in (0, 1, 2, 3, 4):
                run(*shape, E, policy)
    for policy in (0, 1, 2, 3, 4):       # BASELINE config 5's per-GPU share
        run(128, 128, 64, 5, 2048, policy, spawn_prob=0.05)


if __name__ == "__main__":
    main()

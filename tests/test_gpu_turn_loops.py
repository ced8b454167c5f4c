"""The turn loops of ``Environment.take_turn`` against each other (round 6: grouped by component; the cases of rounds 3-5 follow at the
end of this file, bodies unchanged).  Reference: ``Agent.transition``, ``sorrel/agents/agent.py:155-173``; exploration
``sorrel/models/pytorch/iqn.py:294-309``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


@pytest.mark.parametrize("shared", [1, 2])
def test_speculative_turns_with_action_values_explore_like_the_sequential_turn(torch_cuda, shared):
    """A policy that returns action VALUES with epsilon > 0 (round-5 advisor finding: the speculative turn took a plain argmax): the eager
    loops hand the values to ``sgw_act`` (``SGW_ACT_QF32``), which explores in-kernel with the engine's keyed draw for (env, turn, agent);
    the speculative turn takes the same choice through ``sgw_choose_actions`` -- state, step outputs and replay rings are equal turn after
    turn, with an epsilon that decays (and differs per model) on the way, and exploration really happens."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

    E, A = 48, 6

    class Values(BaseModel):
        def __init__(self, input_size, action_space, memory, eps):
            super().__init__(input_size, action_space, memory_size=memory, num_envs=E, device="cuda:0")
            g = torch.Generator().manual_seed(5)
            self.weight = torch.randn((int(np.prod(input_size)), action_space), generator=g).cuda()
            self.epsilon = eps

        def take_action(self, state):
            return state.reshape(state.shape[0], -1) @ self.weight          # [n, n_actions]: the engine takes the argmax / explores

    envs = []
    for speculate in (False, True):
        made, models = [], []

        def factory(input_size, action_space):
            k = len(made) * shared // A
            made.append(k)
            if k >= len(models):
                models.append(Values(input_size, action_space, 4 * A, 0.5 if k == 0 else 0.25))
            return models[k]

        env = make_env(14, 17, A, 3, E, p=0.06, seed=11, model_factory=factory)
        env.speculate_turns = "always" if speculate else False
        envs.append((env, models))
    (eager, em), (spec, sm) = envs
    greedy_differs = 0
    for t in range(10):
        if t == 6:
            eager.reset()
            spec.reset()
        for m in em + sm:
            m.epsilon *= 0.9                                                  # a decaying epsilon reaches both loops
        eager.take_turn()
        spec.take_turn()
        assert spec.turn_plan()["loop"] == "speculative"
        torch.cuda.synchronize()
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(eager.world, name), getattr(spec.world, name)), (t, name)
        assert torch.equal(eager.rewards, spec.rewards) and torch.equal(eager.actions, spec.actions), t
        # what a greedy choice over the windows the agents acted on would have been: exploration must show
        for a in range(A):
            greedy = (spec.obs_of(a).reshape(E, -1) @ spec.agents[a].model.weight).argmax(dim=1)
            greedy_differs += int((greedy != spec.actions[:, a].to(torch.int64)).sum())
    assert greedy_differs > 0.1 * 10 * A * E * 0.2          # (epsilon 0.2-0.45 x 3/4 of the uniform draws differ from the argmax)
    for a in range(A):
        ma, mb = eager.agents[a].model.memory, spec.agents[a].model.memory
        assert (ma.idx, ma.size) == (mb.idx, mb.size)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(ma, name), getattr(mb, name)), (a, name)
    eager.raise_on_status()
    spec.raise_on_status()


def test_choose_actions_is_the_choice_of_the_act_launch(torch_cuda):
    """``sgw_choose_actions`` row by row against ``sgw_act(SGW_ACT_QF32)``: NaN counts as the maximum, ties take the first index, epsilon 1
    always explores with the action SGW_STEP_RANDOM_ACTIONS would draw, epsilon 0 never, a list of rows picks (agent, env) pairs."""
    torch = torch_cuda
    from sorrel_amd.engine import GridEngine
    from sorrel_amd.spec import treasurehunt_spec

    E, A = 200, 5
    ws = treasurehunt_spec(12, 12, A, 2, spawn_prob=0.02, seed=3)
    eng = GridEngine(ws, E, device="cuda:0")
    eng.reset(epoch=0)
    g = torch.Generator().manual_seed(2)
    q = torch.randn((A * E, ws.num_actions), generator=g).cuda()
    q[5, 1] = float("nan")
    q[6] = 0.25                                                   # a tie: index 0
    q[7, 2] = q[7, 3] = 9.0
    eng.turn_set(2, 6)                                            # epoch 2, six turns completed: the turn in flight is 7
    eng.turn_epsilon(0.0)
    greedy = eng.choose_actions(q, None, 2, 7)
    ref = q.argmax(dim=1)
    ref[5], ref[6], ref[7] = 1, 0, 2
    assert torch.equal(greedy, ref)
    eng.turn_epsilon(1.0)
    eng.epoch, eng.turn = 2, 6
    eng.random_actions()                                          # actions[E][A] <- the draws of turn 7 (SGW_STREAM_ACTION)
    explored = eng.choose_actions(q, None, 2, 7)
    assert torch.equal(explored.view(A, E).t().contiguous(), eng.actions.to(torch.int64))
    eng.turn_epsilon(0.0)
    eng.turn_epsilon(1.0, 3)                                      # only agent 3 explores
    idx = torch.tensor([3 * E + 17, 0 * E + 17, 3 * E + 199, 4 * E + 0], dtype=torch.int64, device="cuda:0")
    picked = eng.choose_actions(q[idx].contiguous(), idx, 2, 7)
    want = torch.stack([eng.actions[17, 3].to(torch.int64), ref[17], eng.actions[199, 3].to(torch.int64), ref[4 * E]])
    assert torch.equal(picked, want)
    with pytest.raises(ValueError):
        eng.choose_actions(q[:, :2], None, 2, 7)
    eng.close()


@pytest.mark.parametrize("values", [False, True], ids=["int_actions", "action_values_eps"])
@pytest.mark.parametrize("which", ["tag", "cleanup", "tag_shared_two_models", "plain_80_agents"])
def test_generic_speculative_turn_equals_the_sequential_turn(torch_cuda, which, values):
    """``Environment.speculate_turns`` for agent rules the resolve kernel does not know (round 6, ``sgw_verify_rows``): Tag ("it" flag in the row,
    tags flip victims' flags and cells), Cleanup (beams on the layer above, facing, all-layer rewards, a 12-element positional tail) and plain
    movers beyond 64 agents -- one model shared by all agents (or two) with a shared replay ring.  World, agent state, step outputs and every
    replay row (tails included) equal the eager agent-after-agent loop's, turn after turn, through ring wrap-arounds and a reset; with action
    values and epsilon > 0 the exploration draws are the sequential turn's."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel

    E = 21
    n_models = 2 if which == "tag_shared_two_models" else 1

    def make(speculate):
        made = []

        class Shared(BaseModel):
            def __init__(self, input_size, n_actions, cap):
                n = int(np.prod(input_size))
                super().__init__((n,), n_actions, memory_size=0, num_envs=E, device="cuda:0")
                self.memory = Buffer(capacity=cap, obs_shape=(n,), num_envs=E, device="cuda:0")
                self.weight = torch.randn((n, n_actions), generator=torch.Generator().manual_seed(3 + n + len(made))).cuda()
                self.epsilon = 0.3 if values else 0.0

            def take_action(self, state):
                q = state.reshape(state.shape[0], -1) @ self.weight
                return q if values else q.argmax(dim=1)

        count = [0]

        def factory(input_size, n_actions):
            A = agents_of[which]
            k = count[0] * n_models // A
            count[0] += 1
            while len(made) <= k:
                made.append(Shared(input_size, n_actions, 3 * A + 1))
            return made[k]

        if which.startswith("tag"):
            from sorrel_amd.entities import EmptyEntity
            from sorrel_amd.examples.tag.env import TagEnv
            from sorrel_amd.worlds import Gridworld

            cfg = {"agent": {"num_agents": 6, "vision_radius": 2, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 50}}
            env = TagEnv(Gridworld(8, 9, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=31), cfg, model_factory=factory)
        elif which == "cleanup":
            from tests.test_api_host import make_cleanup_env

            env = make_cleanup_env(E=E, seed=7, device="cuda:0", model_factory=factory)
        else:
            from tests.gpu_common import make_env

            env = make_env(26, 27, 80, 2, E, p=0.05, seed=9, model_factory=factory)
        env.speculate_turns = "always" if speculate else False
        return env, made

    agents_of = {"tag": 6, "tag_shared_two_models": 6, "cleanup": None, "plain_80_agents": 80}
    if which == "cleanup":
        from tests.test_api_host import make_cleanup_env

        agents_of["cleanup"] = len(make_cleanup_env(E=2, seed=7, device="cuda:0").agents)
    (a, ma), (b, mb) = make(False), make(True)
    A = len(a.agents)
    assert len(ma) == len(mb) == n_models
    passes = []
    for t in range(14):
        if t == 9:
            a.reset(); b.reset()
        for m in ma + mb:
            m.epsilon *= 0.95
        a.take_turn()
        b.take_turn()
        assert b.turn_plan()["loop"] == "speculative", b.turn_plan()
        passes.append(b.speculation_passes)
        torch.cuda.synchronize()
        names = ("grid", "agent_pos", "total_reward") + (("agent_state",) if which.startswith("tag") else ()) + (("agent_dir",) if which == "cleanup" else ())
        for name in names:
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert torch.equal(a.rewards, b.rewards) and torch.equal(a.actions, b.actions), t
    assert b._spec_generic is True and max(passes) >= 2 and max(passes) <= A + 1
    for x, y in zip(ma, mb):
        assert (x.memory.idx, x.memory.size) == (y.memory.idx, y.memory.size)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(x.memory, name), getattr(y.memory, name)), name
    assert float(b.world.total_reward.abs().sum()) > 0
    a.raise_on_status()
    b.raise_on_status()


# ------------------------------------------------------------------ moved here from the by-round files of rounds 2-5 (bodies unchanged)
import json  # noqa: E402,F401
import os  # noqa: E402,F401
import subprocess  # noqa: E402,F401
import sys  # noqa: E402,F401

from oracle import gridstep_oracle as O  # noqa: E402,F401
from sorrel_amd import _native as N  # noqa: E402,F401
from tests import helpers as H  # noqa: E402,F401
from tests.gpu_common import *  # noqa: E402,F401,F403


def test_environment_policy_turn_protocols_agree_and_overridden_take_turn_is_called(torch_cuda, tmp_path):
    """Environment.take_turn with policy models: the patched-window protocol == the 1 + A protocol, turn after turn; a
    world mutated by host code in the middle of a turn falls back to rendering on demand; and a subclass that overrides
    take_turn gets it called every turn by run_experiment / generate_memories even with device-random models (the
    reference's loop always goes through take_turn, sorrel/environment.py:160-166)."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel, RandomModel
    from tests.gpu_common import make_env

    E = 21

    class Policy(BaseModel):
        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=6, num_envs=E, device="cuda:0")

        def take_action(self, state):
            s = state.reshape(state.shape[0], -1)
            return (s.sum(dim=1).long() * 3 + (s[:, ::7].sum(dim=1).long())) % 4

    a, b = (make_env(13, 15, 5, 2, E, p=0.05, model_factory=Policy) for _ in range(2))
    b.patch_windows = False
    for t in range(9):
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        assert torch.equal(a.world.grid, b.world.grid) and torch.equal(a.world.agent_pos, b.world.agent_pos), t
        assert torch.equal(a.rewards, b.rewards) and torch.equal(a.actions, b.actions) and torch.equal(a.world.total_reward, b.world.total_reward), t
        for x, y in zip(a.agents, b.agents):
            assert torch.equal(x.model.memory.states, y.model.memory.states), t
    a.raise_on_status()

    # host code that edits the world between two agents' transitions: the remaining agents render on demand
    class Meddler(type(a.agents[0])):
        def act(self, world, action):
            out = super().act(world, action)
            if self.slot == 1:
                world.mutations += 1           # what world.add / remove / move do
            return out

    c, d2 = (make_env(13, 15, 5, 2, E, p=0.05, model_factory=Policy) for _ in range(2))
    c.agents[1].__class__ = Meddler
    d2.patch_windows = False
    for t in range(5):
        c.take_turn()
        d2.take_turn()
        torch.cuda.synchronize()
        assert torch.equal(c.world.grid, d2.world.grid) and torch.equal(c.rewards, d2.rewards), t
        for x, y in zip(c.agents, d2.agents):
            assert torch.equal(x.model.memory.states, y.model.memory.states), t

    calls = []

    def counting(env):
        orig = type(env).take_turn

        class Counting(type(env)):
            def take_turn(self, actions=None):
                calls.append(self.turn)
                return orig(self, actions)

        env.__class__ = Counting
        return env

    e1 = counting(make_env(12, 12, 2, 2, 16, max_turns=7))
    assert all(isinstance(ag.model, RandomModel) for ag in e1.agents)
    e1.run_experiment(epochs=1, logging=False)
    assert len(calls) == 2 * 7, calls
    calls.clear()
    e1.generate_memories(num_games=2, output_dir=tmp_path)
    assert len(calls) == 2 * 7, calls


def test_agents_with_different_specs_vs_the_reference_fixture_device_random(torch_cuda):
    """Radius 2 / radius 4 / full_view / another entity list and fill kind / a float map, three action lists, RandomModel on every
    agent: sweep + per agent (window, act) on the handle of its own specs.  Every turn of the reference's run, envs 0 / 3 / 11."""
    torch = torch_cuda
    env, (d, base, views, full, defs) = make_mixed_env(12, "cuda:0")
    ids = [int(e) for e in d["env_ids"]]
    shapes = [d[f"obs_a{a}"].shape[2:] for a in range(len(defs))]
    eng = env._ensure_engine()
    assert env._mixed and len(env._group_engines) == 5
    torch.cuda.synchronize()
    for n, e in enumerate(ids):
        assert np.array_equal(to_fixture_ids(env, env.world.grid[e].cpu().numpy()), d["grid0"][n]) and np.array_equal(env.world.agent_pos[e].cpu().numpy(), d["pos0"][n])
    for t in range(d["grid"].shape[0]):
        env.take_turn()
        _compare_turn(torch, env, d, t, ids, shapes)
    env.raise_on_status()
    assert env.capture_turn() is None and "different" in str(env.capture_error)
    assert isinstance(env.obs, list) and len(env.obs) == 5


def test_agents_with_different_specs_policy_driven_with_replay_memories(torch_cuda):
    """The same world with a policy on every agent (pov -> get_action -> act -> add_memory, agent after agent): the policies replay the
    oracle's actions for all 12 envs; windows, rewards and the replay memories against the oracle's mixed rollout, the three fixture
    envs against the reference's own arrays."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel

    E = 12
    d, base, views, full, defs = H.load_mixed()
    T = d["grid"].shape[0]
    want = O.rollout_mixed(views, full, list(range(E)), T)
    clock = {"t": 0}

    class Replay(BaseModel):
        def __init__(self, input_size, action_space, slot):
            super().__init__(input_size, action_space, memory_size=T + 2, num_envs=E, device="cuda:0")
            self.slot = slot
            self.seen = []

        def take_action(self, state):
            self.seen.append(state.clone())
            return torch.from_numpy(want["actions"][clock["t"], :, self.slot].astype(np.int64)).to("cuda:0")

    env, _ = make_mixed_env(E, "cuda:0", model_factory=Replay)
    ids = [int(e) for e in d["env_ids"]]
    shapes = [d[f"obs_a{a}"].shape[2:] for a in range(len(defs))]
    for t in range(T):
        clock["t"] = t
        env.take_turn()
        _compare_turn(torch, env, d, t, ids, shapes)
        torch.cuda.synchronize()
        assert np.array_equal(env.rewards.cpu().numpy(), want["rewards"][t]) and np.array_equal(to_fixture_ids(env, env.world.grid.cpu().numpy()), want["grid"][t])
        for a, agent in enumerate(env.agents):
            assert np.array_equal(agent.model.seen[t].cpu().numpy().reshape((E,) + shapes[a]), want[f"obs_a{a}"][t]), (t, a)
    for a, agent in enumerate(env.agents):          # what add_memory stored: float32 windows, int64 actions, float32 rewards, done 0
        mem = agent.model.memory
        assert mem.size == T
        assert np.array_equal(mem.states[:T].cpu().numpy().reshape((T, E) + shapes[a]), want[f"obs_a{a}"])
        assert np.array_equal(mem.actions[:T].cpu().numpy().reshape(T, E), want["actions"][:, :, a].astype(np.int64))
        assert np.array_equal(mem.rewards[:T].cpu().numpy().reshape(T, E), want["rewards"][:, :, a])
        assert float(mem.dones.sum()) == 0.0
    env.raise_on_status()


def test_agents_with_different_specs_given_actions_and_a_bad_index(torch_cuda):
    """``take_turn(actions)``: indices into each agent's OWN list (agent 4 has three actions: index 3 is a KeyError there and only
    there), and two agents that share their specs share one handle."""
    torch = torch_cuda
    d, base, views, full, defs = H.load_mixed()
    defs = [defs[0], defs[4], defs[0], defs[2]]
    env, _ = make_mixed_env(6, "cuda:0", defs=defs)
    eng = env._ensure_engine()
    assert len(env._group_engines) == 3 and env._agent_engine[0] is env._agent_engine[2]
    v = [views[0], views[4], views[0], views[2]]
    f = [False, False, False, True]
    import dataclasses
    v = [dataclasses.replace(x, num_agents=4, agent_type=[6] * 4) for x in v]
    states = [O.reset_env(v[0], e, 0) for e in range(6)]
    rng = np.random.default_rng(5)
    for t in range(1, 9):
        acts = np.stack([rng.integers(0, [4, 3, 4, 5]) for _ in range(6)]).astype(np.uint8)
        env.take_turn(torch.from_numpy(acts).to("cuda:0"))
        torch.cuda.synchronize()
        for e in range(6):
            o, a_, r = O.step_env_mixed(v, f, states[e], e, 0, t, actions=acts[e])
            for a in range(4):
                assert np.array_equal(env.obs_of(a)[e].cpu().numpy(), o[a]), (t, e, a)
            assert np.array_equal(env.rewards[e].cpu().numpy(), r)
        assert np.array_equal(to_fixture_ids(env, env.world.grid.cpu().numpy()), np.stack([s.grid for s in states]))
    env.raise_on_status()
    bad = torch.zeros((6, 4), dtype=torch.uint8, device="cuda:0")
    bad[:, 1] = 3                                    # agent 1's list has three names
    env.take_turn(bad)
    with pytest.raises(KeyError):
        env.raise_on_status()


def test_environment_speculative_turns_equal_the_eager_loop(torch_cuda):
    """Environment.speculate_turns: agents that share one model (one batched forward pass per pass, one shared replay ring filled in
    agent order) and agents with a model each -- grids, totals, step outputs and every replay row equal the eager agent-after-agent
    loop's after 12 turns and a reset."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

    E, A = 40, 6

    class Linear(BaseModel):
        def __init__(self, input_size, action_space, memory=0):
            super().__init__(input_size, action_space, memory_size=memory, num_envs=E, device="cuda:0")
            g = torch.Generator().manual_seed(99)
            self.weight = torch.randn((int(np.prod(input_size)), action_space), generator=g).cuda()
            self.calls = 0

        def take_action(self, state):
            self.calls += 1
            return (state.reshape(state.shape[0], -1) @ self.weight).argmax(dim=1)

    for shared, cap in ((1, 4 * A), (1, 4 * A + 1), (2, 3 * A)):             # (4 A: the turn's rows of the shared ring are contiguous -- the windows are
        envs = []                                                             # rendered straight into them; 4 A + 1: they are not, add_batch copies;
        for speculate in (False, True):                                       # 2: two models of three agents each -- two batches per pass)
            made = []

            def factory(input_size, action_space):
                k = len(made) * shared // A
                made.append(k)
                if k >= len(models):
                    models.append(Linear(input_size, action_space, memory=cap))
                return models[k]

            models = []

            env = make_env(14, 17, A, 3, E, p=0.06, seed=7, model_factory=factory)
            env.speculate_turns = "always" if speculate else False      # ("always": also where the cost model would keep the sequential loop)
            envs.append(env)
        eager, spec = envs
        for t in range(12):
            if t == 7:
                eager.reset()
                spec.reset()
            eager.take_turn()
            spec.take_turn()
        torch.cuda.synchronize()
        assert spec.speculation_passes >= 1 and spec._speculation_groups(spec._engine) is not None
        assert len(spec._speculation_groups(spec._engine)) == shared
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(eager.world, name), getattr(spec.world, name)), (shared, name)
        assert torch.equal(eager.rewards, spec.rewards) and torch.equal(eager.actions, spec.actions)
        for a in range(A):
            ma, mb = eager.agents[a].model.memory, spec.agents[a].model.memory
            assert (ma.idx, ma.size) == (mb.idx, mb.size)
            assert torch.equal(ma.states, mb.states) and torch.equal(ma.actions, mb.actions) and torch.equal(ma.rewards, mb.rewards)
            assert torch.equal(ma.dones, mb.dones)
        assert spec.agents[0].model.calls < eager.agents[0].model.calls          # one forward pass per PASS, not per agent
        for a in range(A):          # obs_of: the window each agent acted on (the eager loop's lives in its replay row; the last add is the last turn's)
            mem = eager.agents[a].model.memory
            k = sum(1 for b in range(a + 1, A) if eager.agents[b].model.memory is mem)
            last = (mem.idx - 1 - k) % mem.capacity
            assert torch.equal(spec.obs_of(a).reshape(E, -1), mem.states[last].reshape(E, -1)), (shared, a)
        eager.raise_on_status()
        spec.raise_on_status()


@pytest.mark.parametrize("case", ["own_rings", "shared_ring", "values_and_ints", "model_edits_the_world", "subclass_with_own_pov"])
def test_fast_policy_loop_equals_the_generic_transition_loop(torch_cuda, case):
    """Environment.fast_policy_loop (agents with the standard hooks stepped without the generic hooks in between) against the
    Agent.transition loop it replaces: grids, positions, totals, step outputs and every replay row after 11 turns and a reset --
    for a ring per agent, one shared ring whose rows wrap mid-turn, agents that return action values (in-kernel argmax / exploration)
    or a plain int, a model that edits the world between two agents (windows rendered on demand from there on), and a subclass that
    overrides pov (the fast loop must not take it)."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.examples.treasurehunt.agents import TreasurehuntAgent
    from sorrel_amd.examples.treasurehunt.entities import Wall
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

    E, A = 37, 5

    class Linear(BaseModel):
        def __init__(self, input_size, action_space, k=0, memory=9):
            super().__init__(input_size, action_space, memory_size=memory, num_envs=E, device="cuda:0")
            g = torch.Generator().manual_seed(50 + k)
            self.weight = torch.randn((int(np.prod(input_size)), action_space), generator=g).cuda()
            self.k, self.env, self.turns = k, None, 0
            self.epsilon = 0.25 if (case == "values_and_ints" and k == 1) else 0.0

        def take_action(self, state):
            q = state.reshape(state.shape[0], -1) @ self.weight
            if case == "values_and_ints" and self.k in (1, 3):
                return q                                   # action values: the act launch chooses
            if case == "values_and_ints" and self.k == 2:
                return 1                                   # a plain int for every env
            if case == "values_and_ints" and self.k == 4:
                return q.argmax(dim=1).to(torch.int32)
            if case == "model_edits_the_world" and self.k == 2:
                self.turns += 1
                if self.turns % 3 == 0:
                    self.env.world.add((1 + self.turns % 5, 2, 0), Wall(), env=None)
            return q.argmax(dim=1)

    envs = []
    for fast in (False, True):
        made = []

        def factory(input_size, action_space):
            if case == "shared_ring":
                if not made:
                    made.append(Linear(input_size, action_space, 0, memory=0))
                    made[0].memory = Buffer(capacity=2 * A + 3, obs_shape=tuple(input_size), num_envs=E, device="cuda:0")
                return made[0]
            made.append(Linear(input_size, action_space, len(made)))
            return made[-1]

        env = make_env(13, 16, A, 2, E, p=0.07, seed=11, model_factory=factory)
        for m in made:
            m.env = env
        if case == "subclass_with_own_pov":
            class Dimmed(TreasurehuntAgent):
                def pov(self, world):
                    return super().pov(world) * 0.5
            env.agents[3].__class__ = Dimmed
        env.fast_policy_loop = fast
        envs.append(env)
    generic, quick = envs
    for t in range(11):
        if t == 6:
            generic.reset()
            quick.reset()
        generic.take_turn()
        quick.take_turn()
    torch.cuda.synchronize()
    plan = quick._fast_plan(quick._engine)
    assert (plan is None) == (case == "subclass_with_own_pov") and generic.__dict__.get("_fast_plan_cache") is None
    for name in ("grid", "agent_pos", "total_reward"):
        assert torch.equal(getattr(generic.world, name), getattr(quick.world, name)), name
    assert torch.equal(generic.rewards, quick.rewards) and torch.equal(generic.actions, quick.actions)
    assert float(quick.world.total_reward.abs().sum()) > 0
    for a in range(A):
        ma, mb = generic.agents[a].model.memory, quick.agents[a].model.memory
        assert (ma.idx, ma.size, ma._dones_dirty) == (mb.idx, mb.size, mb._dones_dirty)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(ma, name), getattr(mb, name)), (a, name)
    generic.raise_on_status()
    quick.raise_on_status()


@pytest.mark.parametrize("case", range(int(os.environ.get("SGW_SOAK", "16")) // 2))
def test_environment_turn_loops_soak(torch_cuda, case):
    """Random Treasurehunt environments through the Python API, policy-driven, four ways: the generic Agent.transition loop, the fast
    loop, the speculative turn (where the agents share few enough models) and a recorded turn -- same seeds, same policies: every world
    tensor, step output and replay row equal after a few turns, a reset and a few more."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

    rng = np.random.default_rng(52000 + case)
    h, w = int(rng.integers(7, 70)), int(rng.integers(7, 70))
    A = int(min(rng.integers(1, 25), max(1, (h - 2) * (w - 2) // 4)))
    r = min(int(rng.integers(1, 6)), (min(h, w) - 1) // 2)
    E = int(rng.integers(1, 50))
    n_models = int(rng.choice([1, 1, 2, A]))                      # agents per model: all share one, two groups, or a model each
    n_models = max(1, min(n_models, A))
    cap = int(rng.integers(A, 4 * A + 3))                          # (a shared ring takes A rows per turn: wraps mid-turn unless a multiple)
    p, seed, T = float(rng.choice([0.0, 0.02, 0.2])), int(rng.integers(0, 2**31)), int(rng.integers(3, 8))

    class Linear(BaseModel):
        def __init__(self, input_size, action_space, k):
            super().__init__(input_size, action_space, memory_size=0, num_envs=E, device="cuda:0")
            self.memory = Buffer(capacity=cap, obs_shape=tuple(input_size), num_envs=E, device="cuda:0")
            g = torch.Generator().manual_seed(1000 * case + k)
            self.weight = torch.randn((int(np.prod(input_size)), action_space), generator=g).cuda()

        def take_action(self, state):
            return (state.reshape(state.shape[0], -1) @ self.weight).argmax(dim=1)

    def build(mode):
        models, made = [], []

        def factory(input_size, action_space):
            k = len(made) * n_models // A
            made.append(k)
            if k >= len(models):
                models.append(Linear(input_size, action_space, k))
            return models[k]

        env = make_env(h, w, A, r, E, p=p, seed=seed % 1000, model_factory=factory)
        env.fast_policy_loop = mode != "generic"
        env.speculate_turns = "always" if mode == "speculative" else False
        if mode == "recorded":
            env.capture_turn(warmup=1)                             # (may decline: the eager loop then plays, which is as good a check)
        return env

    envs = {mode: build(mode) for mode in ("generic", "fast", "speculative", "recorded")}
    envs["generic"].take_turn()
    envs["fast"].take_turn()
    envs["speculative"].take_turn()
    if envs["recorded"]._captured is None:
        envs["recorded"].take_turn()
    for t in range(2 * T):
        for env in envs.values():
            if t == T:
                env.reset()
            env.take_turn()
    torch.cuda.synchronize()
    ref = envs["generic"]
    ctx = f"case {case}: {h}x{w}, {A} agents on {n_models} models, r {r}, {E} envs, ring of {cap}"
    _LOOPS_SEEN["cases"] += 1
    _LOOPS_SEEN["speculative"] += int(getattr(envs["speculative"], "speculation_passes", 0) > 0)
    _LOOPS_SEEN["recorded"] += int(envs["recorded"]._captured is not None and envs["recorded"]._captured.turns_replayed > 0)
    _LOOPS_SEEN["fast"] += int(envs["fast"]._fast_plan(envs["fast"]._engine) is not None)
    for mode, env in envs.items():
        if mode == "generic":
            continue
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(ref.world, name), getattr(env.world, name)), (ctx, mode, name)
        assert torch.equal(ref.rewards, env.rewards) and torch.equal(ref.actions, env.actions), (ctx, mode)
        for a in range(A):
            ma, mb = ref.agents[a].model.memory, env.agents[a].model.memory
            assert (ma.idx, ma.size) == (mb.idx, mb.size), (ctx, mode, a)
            for name in ("states", "actions", "rewards", "dones"):
                assert torch.equal(getattr(ma, name), getattr(mb, name)), (ctx, mode, a, name)
        env.raise_on_status()


def test_environment_turn_loops_soak_reached_every_loop():
    """... and the soak above did run what it names (a case whose agents have a model each keeps the sequential turn; a capture may decline)."""
    seen = _LOOPS_SEEN
    if seen["cases"] < 8:
        pytest.skip("the soak did not run in this session")
    assert seen["fast"] == seen["cases"] and seen["speculative"] >= seen["cases"] // 4 and seen["recorded"] >= seen["cases"] // 2, seen


def test_speculate_turns_true_follows_the_cost_model(torch_cuda):
    """``speculate_turns = True`` speculates only where the measured cost model says it is the faster turn: many agents on one model yes, few
    agents over a large batch no (the sequential loop is device-bound there); "always" speculates wherever it is possible."""
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

    def env_of(h, w, A, r, E):
        one = []

        class Shared(BaseModel):
            def __init__(self, input_size, action_space):
                super().__init__(input_size, action_space, memory_size=2 * A, num_envs=E, device="cuda:0")

            def take_action(self, state):
                return state.reshape(state.shape[0], -1).sum(dim=1).long() % 4

        def factory(input_size, action_space):
            if not one:
                one.append(Shared(input_size, action_space))
            return one[0]

        return make_env(h, w, A, r, E, p=0.02, seed=3, model_factory=factory)

    many = env_of(24, 24, 20, 2, 64)
    many.speculate_turns = True
    many.take_turn()
    assert many._speculation_groups(many._engine) is not None and many.speculation_passes >= 1
    few = env_of(32, 32, 8, 3, 16384)                      # 154 MB of windows, eight agents: 406 us speculative against 292 sequential
    few.speculate_turns = True
    few.take_turn()
    assert few._speculation_groups(few._engine) is None and not hasattr(few, "speculation_passes")
    few.speculate_turns = "always"
    few.take_turn()
    assert few.speculation_passes >= 1
    many.raise_on_status()
    few.raise_on_status()


@pytest.mark.parametrize("which", ["tag", "cleanup"])
def test_fast_policy_loop_on_the_tag_and_cleanup_examples(torch_cuda, which):
    """The shipped Tag and Cleanup agents -- pov = the engine's row (window + the "it" flag / the positional code), get_action =
    model.take_action -- go through the fast eager loop too: 30 turns across ring wrap-arounds and a reset leave exactly what the generic
    Agent.transition loop leaves (world, agent state, step outputs, every replay row incl. its tail)."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel

    E = 23

    class Policy(BaseModel):
        def __init__(self, input_size, n_actions):
            n = int(np.prod(input_size))
            super().__init__((n,), n_actions, memory_size=6, num_envs=E, device="cuda:0")
            self.weight = torch.randn((n, n_actions), generator=torch.Generator().manual_seed(3 + n)).cuda()

        def take_action(self, state):
            return (state.reshape(state.shape[0], -1) @ self.weight).argmax(dim=1)

    def make(fast):
        if which == "tag":
            from sorrel_amd.entities import EmptyEntity
            from sorrel_amd.examples.tag.env import TagEnv
            from sorrel_amd.worlds import Gridworld

            cfg = {"agent": {"num_agents": 6, "vision_radius": 2, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 50}}
            env = TagEnv(Gridworld(8, 9, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=31), cfg, model_factory=Policy)
        else:
            from tests.test_api_host import make_cleanup_env

            env = make_cleanup_env(E=E, seed=7, device="cuda:0", model_factory=Policy)
        env.fast_policy_loop = fast
        return env

    a, b = make(False), make(True)
    for t in range(30):
        if t == 17:
            a.reset(); b.reset()
        a.take_turn()
        b.take_turn()
    torch.cuda.synchronize()
    eng = b._engine
    assert eng.row_tail == (1 if which == "tag" else 12)
    assert b._fast_plan(eng) is not None and a.__dict__.get("_fast_plan_cache") is None
    for name in ("grid", "agent_pos", "total_reward") + (("agent_state",) if which == "tag" else ("agent_dir",)):
        assert torch.equal(getattr(a.world, name), getattr(b.world, name)), name
    assert torch.equal(a.rewards, b.rewards) and torch.equal(a.actions, b.actions)
    assert float(b.world.total_reward.abs().sum()) > 0
    for x, y in zip(a.agents, b.agents):
        mx, my = x.model.memory, y.model.memory
        assert (mx.idx, mx.size) == (my.idx, my.size)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(mx, name), getattr(my, name)), name
    a.raise_on_status()
    b.raise_on_status()


def test_turn_plan_names_the_loop_that_plays(torch_cuda):
    """Environment.turn_plan(): the diagnostic agrees with what take_turn() then does."""
    from sorrel_amd.examples.treasurehunt.agents import TreasurehuntAgent
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env
    from tests.gpu_common import _policy_env

    rnd = make_env(14, 14, 3, 2, 16)                                   # RandomModel agents
    assert rnd.turn_plan()["loop"] == "fused"
    env = _policy_env(64, shape=(32, 32, 8, 3), memory=4)
    plan = env.turn_plan()
    assert plan["loop"] == "fast" and plan["one_launch_windows"] is True and plan["launches"] == 9, plan
    env.fuse_sweep_and_rows = False
    assert env.turn_plan()["launches"] == 10
    env.fast_policy_loop = False
    assert env.turn_plan()["loop"] == "generic" and "switched off" in env.turn_plan()["fast"]
    env.fast_policy_loop = True

    class Own(TreasurehuntAgent):
        def get_action(self, state):
            return super().get_action(state)

    env.agents[2].__class__ = Own
    env.__dict__.pop("_fast_plan_cache", None)
    assert env.turn_plan()["loop"] == "generic"
    env.agents[2].__class__ = TreasurehuntAgent
    env.__dict__.pop("_fast_plan_cache", None)
    assert env.capture_turn() is not None
    assert env.turn_plan()["loop"] == "recorded"
    env.take_turn()
    mixed = make_mixed_env(9, "cuda:0")[0]
    assert mixed.turn_plan()["loop"] == "per-agent handles" and mixed.turn_plan()["handles"] >= 2


@pytest.mark.parametrize("case", range(max(6, int(os.environ.get("SGW_SOAK", "0")) // 8)))
def test_generic_speculative_turn_soak_random_examples(torch_cuda, case):
    """The Tag and Cleanup examples at random sizes (map, agents, vision, beam radius, batch), one shared linear policy + ring, action values with exploration
    in every other case: the generic speculative turn (``"always"``) against the eager loop -- world, agent state, step outputs, the ring -- over 6 turns."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel

    rng = np.random.default_rng(4000 + case)
    which = ("tag", "cleanup")[case % 2]
    values = (case // 2) % 2 == 1
    E = int(rng.integers(3, 30))
    if which == "tag":
        h, w = int(rng.integers(7, 20)), int(rng.integers(7, 20))
        A = int(rng.integers(3, min(12, (h - 2) * (w - 2) // 3) + 1))            # (three agents per model at least: below that the turn is not speculated)
        r = int(rng.integers(1, min(4, (min(h, w) - 1) // 2) + 1))
    else:
        h, w = int(rng.integers(11, 24)), int(rng.integers(12, 30))
        A = int(rng.integers(3, 9))
        r = int(rng.integers(1, min(5, (min(h, w) - 1) // 2) + 1))
    beam = int(rng.integers(1, 4))
    seed = int(rng.integers(0, 1000))

    def make(speculate):
        one = []

        class Shared(BaseModel):
            def __init__(self, input_size, n_actions):
                n = int(np.prod(input_size))
                super().__init__((n,), n_actions, memory_size=0, num_envs=E, device="cuda:0")
                self.memory = Buffer(capacity=2 * A + 1, obs_shape=(n,), num_envs=E, device="cuda:0")
                self.weight = torch.randn((n, n_actions), generator=torch.Generator().manual_seed(77 + case)).cuda()
                self.epsilon = 0.25 if values else 0.0

            def take_action(self, state):
                q = state.reshape(state.shape[0], -1) @ self.weight
                return q if values else q.argmax(dim=1)

        def factory(input_size, n_actions):
            if not one:
                one.append(Shared(input_size, n_actions))
            return one[0]

        if which == "tag":
            from sorrel_amd.entities import EmptyEntity
            from sorrel_amd.examples.tag.env import TagEnv
            from sorrel_amd.worlds import Gridworld

            cfg = {"agent": {"num_agents": A, "vision_radius": r, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 50}}
            env = TagEnv(Gridworld(h, w, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=seed), cfg, model_factory=factory)
        else:
            from sorrel_amd.examples.cleanup.entities import EmptyEntity as CEmpty
            from sorrel_amd.examples.cleanup.env import CleanupEnv
            from sorrel_amd.examples.cleanup.main import make_config
            from sorrel_amd.examples.cleanup.world import CleanupWorld

            cfg = make_config(height=h, width=w, num_agents=A, vision=r, beam_radius=beam)
            env = CleanupEnv(CleanupWorld(cfg, CEmpty(), num_envs=E, device="cuda:0", seed=seed), cfg, model_factory=factory)
        env.speculate_turns = "always" if speculate else False
        return env, one

    (a, ma), (b, mb) = make(False), make(True)
    ctx = f"case {case}: {which} {h}x{w} A={A} r={r} E={E} values={values}"
    for t in range(6):
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        assert b.turn_plan()["loop"] == "speculative", (ctx, b.turn_plan())
        names = ("grid", "agent_pos", "total_reward") + (("agent_state",) if which == "tag" else ("agent_dir",))
        for name in names:
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (ctx, t, name)
        assert torch.equal(a.rewards, b.rewards) and torch.equal(a.actions, b.actions), (ctx, t)
    for name in ("states", "actions", "rewards", "dones"):
        assert torch.equal(getattr(ma[0].memory, name), getattr(mb[0].memory, name)), (ctx, name)
    a.raise_on_status()
    b.raise_on_status()

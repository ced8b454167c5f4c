"""Round 5, GPU: agents that differ in their observation / action specs (one engine handle per distinct pair over the same world
tensors) against the reference-generated fixture and the oracle; step_big at eight waves per SIMD."""
import numpy as np
import pytest

from tests import helpers as H
from tests.mixed_env import make_mixed_env, to_fixture_ids
from oracle import gridstep_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no silent CPU fallback)")
    return torch


def _compare_turn(torch, env, d, t, ids, shapes):
    torch.cuda.synchronize()
    for n, e in enumerate(ids):
        for a in range(len(shapes)):
            got = env.obs_of(a)[e].cpu().numpy().reshape(shapes[a])
            assert np.array_equal(got, d[f"obs_a{a}"][t, n]), f"turn {t + 1} env {e}: window of agent {a}"
        assert np.array_equal(env.actions[e].cpu().numpy(), d["actions"][t, n]), (t, e)
        assert np.array_equal(env.rewards[e].cpu().numpy(), d["rewards"][t, n]), (t, e)
        assert float(env.total_reward[e]) == d["total_reward"][t, n]
        assert np.array_equal(to_fixture_ids(env, env.world.grid[e].cpu().numpy()), d["grid"][t, n])
        assert np.array_equal(env.world.agent_pos[e].cpu().numpy(), d["pos"][t, n])


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

"""GPU tests through the Sorrel-shaped API (Environment / Gridworld / Agent / ObservationSpec):
the classes compile to engine tables, take_turn runs the HIP kernels, results equal the oracle."""
import numpy as np
import pytest

from tests import helpers as H
from oracle import gridstep_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no silent CPU fallback)")
    return torch


def make_env(h, w, a, r, E, p=0.02, seed=5, model_factory=None, dense=0.0):
    from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
    from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
    from sorrel_amd.examples.treasurehunt.main import make_config
    from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld

    cfg = make_config(h, w, a, r, spawn_prob=p)
    if dense:
        cfg["world"]["dense_prob"] = dense
    world = TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=seed)
    return TreasurehuntEnv(world, cfg, model_factory=model_factory)


def test_treasurehunt_env_fused_random_vs_oracle(torch_cuda):
    torch = torch_cuda
    E, T = 48, 10
    env = make_env(16, 16, 4, 2, E, dense=0.2)
    ospec = H.oracle_spec(env.compile_spec())
    states = [O.reset_env(ospec, e, epoch=0) for e in range(E)]
    torch.cuda.synchronize()
    assert np.array_equal(env.world.grid.cpu().numpy(), np.stack([s.grid for s in states]))
    for t in range(1, T + 1):
        env.take_turn()
        assert env.turn == t
        torch.cuda.synchronize()
        obs, rew = env.obs.cpu().numpy(), env.rewards.cpu().numpy()
        for e in range(E):
            o, a, r = O.step_env(ospec, states[e], e, 0, t)
            assert np.array_equal(obs[e], o) and np.array_equal(rew[e], r)
            assert np.array_equal(env.actions[e].cpu().numpy(), a)
        assert np.array_equal(env.world.grid.cpu().numpy(), np.stack([s.grid for s in states]))
        assert np.array_equal(env.world.total_reward.cpu().numpy(), np.array([s.total_reward for s in states]))
    assert not env.dones.any()
    # reference-style accessors on the batched world
    a0 = env.agents[0]
    assert a0.location == (int(states[0].pos[0, 0]), int(states[0].pos[0, 1]), 1)
    assert env.world.observe(a0.location).kind == "TreasurehuntAgent"
    assert env.world.map[0, 0, 1].kind == "Wall" and env.world.map[0, 0, 0].kind == "EmptyEntity"


def test_reset_starts_a_new_epoch(torch_cuda):
    torch = torch_cuda
    env = make_env(12, 12, 3, 2, 16)
    ospec = H.oracle_spec(env.compile_spec())
    for _ in range(3):
        env.take_turn()
    env.reset()
    assert env.turn == 0 and env.epoch == 1
    torch.cuda.synchronize()
    states = [O.reset_env(ospec, e, epoch=1) for e in range(16)]
    assert np.array_equal(env.world.grid.cpu().numpy(), np.stack([s.grid for s in states]))
    assert float(env.world.total_reward.abs().sum()) == 0.0
    env.take_turn()
    torch.cuda.synchronize()
    for e in range(16):
        O.step_env(ospec, states[e], e, 1, 1)
    assert np.array_equal(env.world.grid.cpu().numpy(), np.stack([s.grid for s in states]))


def policy_np(obs, a):
    """Deterministic toy policy of the flattened observation (and the agent slot)."""
    return int((int(obs.sum()) * 7 + int(obs[1].sum()) * 3 + int(obs[5].sum()) + a) % 4)


def test_policy_driven_agents_take_the_phased_path(torch_cuda):
    """Agents with a real model: sweep, then per agent observe -> policy -> act, in list order."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel

    class ToyPolicy(BaseModel):
        slot_counter = [0]

        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=8, num_envs=24, device="cuda:0")
            self.slot = ToyPolicy.slot_counter[0]
            ToyPolicy.slot_counter[0] += 1

        def take_action(self, state):
            s = state.reshape(state.shape[0], 6, -1)
            return ((s.sum(dim=(1, 2)).long() * 7 + s[:, 1].sum(dim=1).long() * 3 + s[:, 5].sum(dim=1).long() + self.slot) % 4)

    ToyPolicy.slot_counter[0] = 0
    E, T = 24, 8
    env = make_env(14, 14, 3, 2, E, p=0.05, model_factory=ToyPolicy)
    ospec = H.oracle_spec(env.compile_spec())
    states = [O.reset_env(ospec, e, epoch=0) for e in range(E)]
    for t in range(1, T + 1):
        env.take_turn()
        torch.cuda.synchronize()
        for e in range(E):
            o, a, r = O.step_env_policy(ospec, states[e], e, 0, t, policy_np)
            assert np.array_equal(env.actions[e].cpu().numpy(), a), (t, e)
            assert np.array_equal(env.rewards[e].cpu().numpy(), r)
        assert np.array_equal(env.world.grid.cpu().numpy(), np.stack([s.grid for s in states]))
        assert np.array_equal(env.world.total_reward.cpu().numpy(), np.array([s.total_reward for s in states]))
    # the replay memory received what the reference would store: obs f32, action i64, reward f32, done 0
    mem = env.agents[1].model.memory
    assert mem.size == T and mem.states.dtype == torch.float32 and mem.actions.dtype == torch.int64
    assert float(mem.dones.sum()) == 0.0


def test_observe_from_arbitrary_cell_and_full_view(torch_cuda):
    torch = torch_cuda
    from sorrel_amd.observation.observation_spec import OneHotObservationSpec
    from sorrel_amd.examples.treasurehunt.env import ENTITY_LIST

    env = make_env(12, 12, 2, 2, 6, dense=0.3)
    ospec = H.oracle_spec(env.compile_spec())
    states = [O.reset_env(ospec, e, epoch=0) for e in range(6)]
    spec = env.agents[0].observation_spec
    got = spec.observe(env.world, (0, 11, 1)).cpu().numpy()          # a corner: mostly out of bounds
    for e in range(6):
        assert np.array_equal(got[e], O.visual_field(ospec, states[e].grid, 0, 11).astype(np.float32))
    full = OneHotObservationSpec(ENTITY_LIST, full_view=True, env_dims=(12, 12))
    fv = full.observe(env.world).cpu().numpy()
    app = ospec.appearance
    for e in range(6):
        want = app[states[e].grid].sum(axis=0).transpose(2, 0, 1)
        assert np.array_equal(fv[e], want)


def test_other_observation_specs_and_visual_field_on_demand(torch_cuda):
    """A spec that differs from the agents' own (radius, entity map, fill kind) gets its own compiled
    handle over the same grid; ``visual_field`` takes the same road (reference signature)."""
    torch = torch_cuda
    from sorrel_amd.observation.observation_spec import OneHotObservationSpec
    from sorrel_amd.observation.visual_field import visual_field
    from sorrel_amd.examples.treasurehunt.env import ENTITY_LIST

    E = 5
    env = make_env(12, 12, 2, 2, E, dense=0.3)
    base = H.oracle_spec(env.compile_spec())
    states = [O.reset_env(base, e, epoch=0) for e in range(E)]
    wide = OneHotObservationSpec(ENTITY_LIST, full_view=False, vision_radius=4, fill_entity_kind="Gem")
    ospec_wide = H.oracle_spec(env.compile_spec(wide))
    assert ospec_wide.vision_radius == 4 and ospec_wide.fill_type != base.fill_type
    got = wide.observe(env.world, env.agents[1]).cpu().numpy()
    own = env.agents[1].observation_spec.observe(env.world, env.agents[1]).cpu().numpy()
    assert got.shape == (E, 6, 9, 9) and own.shape == (E, 6, 5, 5)
    for e in range(E):
        y, x = states[e].pos[1]
        assert np.array_equal(got[e], O.visual_field(ospec_wide, states[e].grid, int(y), int(x)).astype(np.float32))
        assert np.array_equal(own[e], O.visual_field(base, states[e].grid, int(y), int(x)).astype(np.float32))
    # visual_field(world, entity_map, vision, location, fill_entity_kind): a float table, from a cell
    emap = {k: np.array([i * 0.5, 1.0]) for i, k in enumerate(ENTITY_LIST)}
    vf = visual_field(env.world, emap, vision=3, location=(2, 9, 1), fill_entity_kind="Wall").cpu().numpy()
    adhoc = OneHotObservationSpec(ENTITY_LIST, full_view=False, vision_radius=3)
    adhoc.override_entity_map(emap)
    ospec_f = H.oracle_spec(env.compile_spec(adhoc))
    assert vf.shape == (E, 2, 7, 7)
    for e in range(E):
        assert np.array_equal(vf[e], O.visual_field(ospec_f, states[e].grid, 2, 9).astype(np.float32))
    full = visual_field(env.world, emap).cpu().numpy()              # vision None -> whole map
    for e in range(E):
        assert np.array_equal(full[e], ospec_f.appearance[states[e].grid].sum(axis=0).transpose(2, 0, 1))
    with pytest.raises(KeyError):                                    # a placed kind the map does not know
        visual_field(env.world, {"EmptyEntity": np.zeros(2), "Wall": np.ones(2)}, vision=2, location=(3, 3, 1))
    # the step engine is untouched by all this
    env.take_turn()
    for e in range(E):
        o, a, r = O.step_env(base, states[e], e, 0, 1)
        assert np.array_equal(env.obs[e].cpu().numpy(), o)


def test_rgb_observation_spec_through_the_api(torch_cuda):
    """config model.observation_spec = "rgb" (examples/treasurehunt/main.py:24): clip(sum, 0, 255) / 255."""
    torch = torch_cuda
    from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
    from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
    from sorrel_amd.examples.treasurehunt.main import make_config
    from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld

    cfg = make_config(12, 12, 3, 2, spawn_prob=0.05)
    cfg["model"]["observation_spec"] = "rgb"
    env = TreasurehuntEnv(TreasurehuntWorld(cfg, EmptyEntity(), num_envs=10, device="cuda:0", seed=2), cfg)
    ospec = H.oracle_spec(env.compile_spec())
    assert ospec.obs_post == 1 and ospec.num_channels == 3
    states = [O.reset_env(ospec, e, epoch=0) for e in range(10)]
    for t in range(1, 7):
        env.take_turn()
        torch.cuda.synchronize()
        for e in range(10):
            o, a, r = O.step_env(ospec, states[e], e, 0, t)
            assert np.array_equal(env.obs[e].cpu().numpy(), o)
    assert 0.0 < float(env.obs.max()) <= 1.0


def test_tag_env_through_the_api(torch_cuda):
    """examples/tag through the class API: fused random turns and the phased policy path."""
    torch = torch_cuda
    from sorrel_amd.entities import EmptyEntity
    from sorrel_amd.examples.tag.env import TagEnv
    from sorrel_amd.worlds import Gridworld

    cfg = {"agent": {"num_agents": 5, "vision_radius": 2, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 5}}
    E = 20
    env = TagEnv(Gridworld(9, 9, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=31), cfg)
    ospec = H.oracle_spec(env.compile_spec())
    assert ospec.agent_rule == O.AGENT_RULE_TAG
    states = [O.reset_env(ospec, e, epoch=0) for e in range(E)]
    for t in range(1, 16):
        env.take_turn()
        torch.cuda.synchronize()
        for e in range(E):
            o, a, r = O.step_env(ospec, states[e], e, 0, t)
            assert np.array_equal(env.obs[e].cpu().numpy(), o) and np.array_equal(env.rewards[e].cpu().numpy(), r)
        assert np.array_equal(env.world.grid.cpu().numpy(), np.stack([s.grid for s in states]))
        assert np.array_equal(env.world.agent_state.cpu().numpy(), np.stack([s.agent_state for s in states]))
    its = torch.stack([a.its for a in env.agents], dim=1)
    assert bool((its.sum(dim=1) == 1).all())
    pov = env.agents[2].pov(env.world)
    assert pov.shape == (E, 4 * 25 + 1) and torch.equal(pov[:, -1].bool(), env.agents[2].its)


def test_cleanup_env_through_the_api(torch_cuda):
    """examples/cleanup through the class API: fused random turns, then the phased policy path
    (entity sweep, then per agent observe -> policy -> act), both against the C oracle."""
    torch = torch_cuda
    from tests.test_api_host import make_cleanup_env

    class BeamHappyPolicy:
        """Deterministic function of the observation; favours the beam actions."""
        memory = None
        device_random = False

        def __init__(self, input_size, n_actions):
            self.n = n_actions
            self.seen = []

        def reset(self): pass
        def start_epoch_action(self, **kw): pass
        def end_epoch_action(self, **kw): pass

        def take_action(self, state):
            self.seen.append(state.clone())
            k = (state[:, :441].sum(dim=1).long() * 5 + (state[:, 441:] > 0).sum(dim=1).long()) % 8
            return torch.where(k >= 6, k - 2, k % self.n)           # clean / zap twice as likely

    E = 33
    env = make_cleanup_env(E=E, seed=7, device="cuda:0")
    ws = env.compile_spec()
    co = H.COracle(ws, E)
    co.grid[...] = env.world.grid.cpu().numpy()
    co.pos[...] = env.world.agent_pos.cpu().numpy()
    for t in range(1, 31):                                           # fused: on-device random actions
        env.take_turn()
        assert co.step(0, t, random_actions=True) == 0
        torch.cuda.synchronize()
        assert np.array_equal(env.obs.cpu().numpy(), co.obs), f"obs turn {t}"
        assert np.array_equal(env.world.grid.cpu().numpy(), co.grid), f"grid turn {t}"
        assert np.array_equal(env.rewards.cpu().numpy(), co.rewards)
        assert np.array_equal(env.world.total_reward.cpu().numpy(), co.total)
        assert np.array_equal(env.world.agent_dir.cpu().numpy(), co.agent_dir)
    assert [a.direction for a in env.agents] == co.agent_dir[0].tolist()
    pol = make_cleanup_env(E=E, seed=7, device="cuda:0", model_factory=BeamHappyPolicy)
    co = H.COracle(ws, E)
    co.grid[...] = pol.world.grid.cpu().numpy()
    co.pos[...] = pol.world.agent_pos.cpu().numpy()
    tab = None
    for t in range(1, 13):                                           # phased: policy-driven
        pol.take_turn()
        torch.cuda.synchronize()
        assert co.step(0, t, actions=pol.actions.cpu().numpy()) == 0  # same actions, whole turn at once
        assert np.array_equal(pol.world.grid.cpu().numpy(), co.grid), f"phased grid turn {t}"
        assert np.array_equal(pol.rewards.cpu().numpy(), co.rewards)
        assert np.array_equal(pol.world.total_reward.cpu().numpy(), co.total)
        assert np.array_equal(pol.world.agent_dir.cpu().numpy(), co.agent_dir)
        for a, agent in enumerate(pol.agents):                       # what each policy saw = the oracle's pov-time view
            seen = agent.model.seen[-1].cpu().numpy()
            assert np.array_equal(seen[:, :441], co.obs[:, a].reshape(E, -1)), (t, a)
    assert (co.grid[:, 2] != ws.default_type).any() or (co.total != 0).any()


def test_host_built_template_world(torch_cuda):
    """populate_environment with plain world.add(...) calls (the reference's imperative style): the same
    template in every env, basic entities, a user-defined MovingAgent subclass, explicit actions."""
    torch = torch_cuda
    from sorrel_amd.action.action_spec import ActionSpec
    from sorrel_amd.agents import MovingAgent
    from sorrel_amd.entities import EmptyEntity, Gem, Wall
    from sorrel_amd.environment import Environment
    from sorrel_amd.models import RandomModel
    from sorrel_amd.observation.observation_spec import OneHotObservationSpec
    from sorrel_amd.worlds import Gridworld

    class Walker(MovingAgent):
        def reset(self): ...
        def pov(self, world): return self.observation_spec.observe(world, self)
        def get_action(self, state): return self.model.take_action(state)
        def is_done(self, world): return world.is_done

    class BasicEnv(Environment):
        def setup_agents(self):
            self.agents = []
            for _ in range(3):
                o = OneHotObservationSpec(["EmptyEntity", "Wall", "Gem", "Walker"], full_view=False, vision_radius=2)
                self.agents.append(Walker(o, ActionSpec(["up", "down", "left", "right", "wait"]), RandomModel((100,), 5)))

        def populate_environment(self):
            H, W = self.world.height, self.world.width
            for y in range(H):
                for x in range(W):
                    if y in (0, H - 1) or x in (0, W - 1):
                        self.world.add((y, x, 0), Wall())
            for (y, x, v) in ((2, 2, 3.5), (2, 5, -2), (4, 4, 7), (5, 2, 1)):
                self.world.add((y, x, 0), Gem(v))
            for agent, loc in zip(self.agents, ((1, 1, 0), (3, 4, 0), (6, 6, 0))):
                self.world.add(loc, agent)

    E = 12
    env = BasicEnv(Gridworld(8, 9, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=1), {"experiment": {"epochs": 1}})
    spec = env.compile_spec()
    ospec = H.oracle_spec(spec)
    g0 = env.world.grid.cpu().numpy()
    assert (g0 == g0[0]).all()                                        # one template, every env
    states = [O.EnvState(grid=g0[e].copy(), pos=env.world.agent_pos[e].cpu().numpy().astype(np.int64), total_reward=0.0)
              for e in range(E)]
    rng = np.random.default_rng(0)
    for t in range(1, 13):
        acts = rng.integers(0, 5, size=(E, 3))
        env.take_turn(torch.from_numpy(acts.astype(np.uint8)).cuda())
        torch.cuda.synchronize()
        for e in range(E):
            o, a, r = O.step_env(ospec, states[e], e, 0, t, actions=acts[e])
            assert np.array_equal(env.obs[e].cpu().numpy(), o) and np.array_equal(env.rewards[e].cpu().numpy(), r)
        assert np.array_equal(env.world.grid.cpu().numpy(), np.stack([s.grid for s in states]))
        assert np.array_equal(env.world.total_reward.cpu().numpy(), np.array([s.total_reward for s in states]))
    env._ensure_engine().raise_on_status()
    assert env.agents[0].location == (int(states[0].pos[0, 0]), int(states[0].pos[0, 1]), 0)


def test_collect_writes_observations_straight_into_the_turn_buffer(torch_cuda):
    """Environment.collect: the step kernel's observation output is the ring slot itself (staged burst path for the
    config-3 shape, direct stores otherwise); contents equal the oracle's, ring wraps."""
    torch = torch_cuda
    from sorrel_amd.buffers import TurnBuffer

    for (h, w, a, r) in ((32, 32, 8, 3), (12, 12, 3, 2)):
        E, T, cap = 9, 7, 5
        env = make_env(h, w, a, r, E, p=0.05)
        ospec = H.oracle_spec(env.compile_spec())
        states = [O.reset_env(ospec, e, epoch=0) for e in range(E)]
        buf = TurnBuffer(cap, E, env.compile_spec().obs_shape, device="cuda:0")
        env.collect(T, buf)
        torch.cuda.synchronize()
        assert len(buf) == cap and buf.idx == T % cap
        want = {}
        for t in range(1, T + 1):
            for e in range(E):
                want[(t, e)] = O.step_env(ospec, states[e], e, 0, t)
        for t in range(T - cap + 1, T + 1):                     # the last `cap` turns are still in the ring
            slot = (t - 1) % cap
            for e in range(E):
                o, act, rew = want[(t, e)]
                assert np.array_equal(buf.obs[slot, e].cpu().numpy(), o), (h, t, e)
                assert np.array_equal(buf.actions[slot, e].cpu().numpy(), act)
                assert np.array_equal(buf.rewards[slot, e].cpu().numpy(), rew)
        st, ac, rw, dn = buf.agent_view(1)
        assert st.shape == (cap, E) + tuple(env.compile_spec().obs_shape[1:]) and float(dn.sum()) == 0.0
        assert np.array_equal(env.world.grid.cpu().numpy(), np.stack([s.grid for s in states]))


def test_run_experiment_thin_loop(torch_cuda):
    env = make_env(10, 10, 2, 2, 32)
    hist = env.run_experiment(epochs=1, max_turns=5, all_reduce=False)
    assert len(hist) == 2 and hist[0]["envs"] == 32.0
    ospec = H.oracle_spec(env.compile_spec())
    tot = 0.0
    for e in range(32):
        st = O.reset_env(ospec, e, epoch=2)       # ctor = epoch 0, first reset() = 1, second = 2
        for t in range(1, 6):
            O.step_env(ospec, st, e, 2, t)
        tot += st.total_reward
    assert hist[1]["sum_total_reward"] == tot


def test_example_mains_run_with_default_device(torch_cuda):
    """The runnable examples build their worlds without an explicit device (bare "cuda"): worlds, engines and
    buffers must agree on the indexed device."""
    import runpy

    from sorrel_amd.buffers import Buffer
    from sorrel_amd.entities import EmptyEntity
    from sorrel_amd.examples.tag.env import TagEnv
    from sorrel_amd.worlds import Gridworld

    cfg = {"experiment": {"epochs": 1, "max_turns": 3}, "agent": {"num_agents": 3, "vision_radius": 2}}
    world = Gridworld(9, 9, 1, EmptyEntity(), num_envs=8)                     # no device argument
    assert world.device.index is not None and Buffer(4, (3,), num_envs=2).device == world.device
    env = TagEnv(world, cfg)
    env.take_turn()
    assert env.obs.device == world.device
    for mod in ("sorrel_amd.examples.treasurehunt.main", "sorrel_amd.examples.tag.main", "sorrel_amd.examples.cleanup.main"):
        runpy.run_module(mod, run_name="__main__")                             # 4096 envs, 2-3 short epochs each

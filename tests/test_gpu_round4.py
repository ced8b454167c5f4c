"""Round 4 (GPU): a whole policy turn as one graph (device-side turn state), the specialiser's cache and fallback, long horizons on
every kernel family."""
import os

import numpy as np
import pytest

from tests import helpers as H
from sorrel_amd import _native as N
from tests.test_gpu_parity import torch_cuda, make_engine, assert_same  # noqa: F401

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ sgw_turn_*: the device counts turns and replay rows
@pytest.mark.parametrize("layout", ["tensor", "rows"])
def test_turn_protocol_counts_on_the_device_vs_oracle(torch_cuda, layout):
    """sgw_turn_begin / sgw_turn_act / sgw_turn_end (windows in the observation tensor, copied into the replay rows at the end of the turn)
    and sgw_turn_begin_rows / sgw_turn_act_rows (windows in per-agent rows, the replay rows written alongside) with the SAME arguments every turn: the turn number, the epoch and each agent's
    replay row come from device memory the engine advances itself.  Every window an agent's policy would read, the rewards, the
    state and the rows of the rings (windows, int64 actions, float32 rewards, zeroed dones) against the C oracle, across a ring
    wrap-around and an epoch change."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(18, 14, 5, 2, spawn_prob=0.06, seed=21, dense_prob=0.1)
    E, A, CAP = 33, 5, 4
    eng, co = make_engine(ws, E, first=5), H.COracle(ws, E, first_env_id=5)
    N_ = int(np.prod(ws.obs_shape[1:]))
    # agents 0..2 own a ring each; agents 3 and 4 SHARE one (two rows per turn)
    def ring(cap):
        return dict(states=torch.full((cap, E, N_), -3.0, device="cuda:0"), rewards=torch.full((cap, E), -3.0, device="cuda:0"),
                    actions=torch.full((cap, E), -3, dtype=torch.int64, device="cuda:0"), dones=torch.full((cap, E), 9.0, device="cuda:0"))
    own = [ring(CAP) for _ in range(3)]
    shared = ring(2 * CAP)
    rings = [(r["states"], r["rewards"], r["actions"], r["dones"], 1, 1) for r in own] + \
            [(shared["states"], shared["rewards"], shared["actions"], shared["dones"], 2 + k, 2) for k in range(2)]
    eng.turn_bind(rings)
    dests = [torch.zeros((E, N_), device="cuda:0") for _ in range(A)]
    rows = eng.window_rows(dests)
    rng = np.random.default_rng(3)
    epoch = 4
    eng.reset(epoch)
    co.reset(epoch)
    eng.turn_set(epoch, 0)
    row_own, row_sh = 1, 2
    for t in range(1, 12):
        if t == 7:          # Environment.reset: a new epoch, the turn counter restarts
            epoch += 1
            eng.reset(epoch)
            co.reset(epoch)
            eng.turn_set(epoch, 0)
        turn = t if t < 7 else t - 6
        acts = rng.integers(0, len(ws.action_dy), size=(E, A), dtype=np.uint8)
        assert co.step(epoch, turn, actions=acts) == 0
        policy = torch.from_numpy(acts.astype(np.int64)).cuda()
        seen = torch.zeros_like(eng.obs)
        if layout == "tensor":
            eng.obs.fill_(-9.0)
            eng.turn_begin()
            for a in range(A):
                seen[:, a] = eng.obs[:, a]
                eng.turn_act(a, policy[:, a].contiguous())
            eng.turn_end()
        else:
            for d in dests:
                d.fill_(-9.0)
            eng.turn_begin_rows(rows)
            for a in range(A):
                seen[:, a] = dests[a].view(E, *ws.obs_shape[1:])
                eng.turn_act_rows(a, rows, policy[:, a].contiguous())
            eng.turn_end(commit_windows=False)
        torch.cuda.synchronize()
        assert eng.turn_state()[:2] == (epoch, turn)
        assert np.array_equal(seen.cpu().numpy(), co.obs), f"turn {t}: windows at pov time"
        assert_same(eng, co, ("grid", "pos", "total", "rewards"), ctx=f"turn {t}")
        for a in range(A):
            r, row = (own[a], row_own) if a < 3 else (shared, (row_sh + (a - 3)) % (2 * CAP))
            assert np.array_equal(r["states"][row].cpu().numpy(), co.obs[:, a].reshape(E, N_)), f"turn {t}: ring row of agent {a}"
            assert np.array_equal(r["actions"][row].cpu().numpy(), acts[:, a].astype(np.int64))
            assert np.array_equal(r["rewards"][row].cpu().numpy(), co.rewards[:, a])
            assert not r["dones"][row].any()
        row_own, row_sh = (row_own + 1) % CAP, (row_sh + 2) % (2 * CAP)
        assert eng.turn_state()[2] == [row_own] * 3 + [row_sh, (row_sh + 1) % (2 * CAP)]
    assert eng.status() == 0


def _policy_env(E, shape=(13, 15, 5, 2), memory=6, seed=5):
    import torch
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    h, w, a, r = shape

    class Policy(BaseModel):
        """A fixed linear layer + argmax: deterministic, capturable (no host synchronisation)."""

        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=memory, num_envs=E, device="cuda:0")
            n = int(np.prod(input_size))
            g = torch.Generator().manual_seed(1234 + n)
            self.weight = torch.randn((n, action_space), generator=g).cuda()

        def take_action(self, state):
            return (state.reshape(state.shape[0], -1) @ self.weight).argmax(dim=1)

    return make_env(h, w, a, r, E, p=0.05, seed=seed, model_factory=Policy)


@pytest.mark.parametrize("layout", ["rows", "tensor"])
def test_captured_turn_equals_the_eager_turn_and_the_oracle(torch_cuda, layout):
    """Environment.capture_turn(): sweep + every window + A x (policy forward, act) + the copy into the replay rows recorded ONCE and
    replayed -- 60 turns across two epoch resets and several wrap-arounds of the 6-row rings equal the eager loop (state, step
    outputs, every buffer row, the buffers' idx / size), and the C oracle stepping the actions the policies chose."""
    torch = torch_cuda
    E = 37
    a, b = _policy_env(E), _policy_env(E)
    b.capture_layout = layout
    cap = b.capture_turn(warmup=2)
    assert cap is not None, getattr(b, "capture_error", None)
    assert (b._capture_rows is not None) == (layout == "rows")
    for _ in range(2):                      # the warm-up turns were real turns
        a.take_turn()
    ws = a._engine.spec
    co = H.COracle(ws, E, first_env_id=0)
    co.grid[...] = a.world.grid.cpu().numpy()
    co.pos[...] = a.world.agent_pos.cpu().numpy()
    co.total[...] = a.world.total_reward.cpu().numpy()
    for t in range(60):
        if t in (20, 41):
            a.reset()
            b.reset()
            co.grid[...] = a.world.grid.cpu().numpy()
            co.pos[...] = a.world.agent_pos.cpu().numpy()
            co.total[...] = a.world.total_reward.cpu().numpy()
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        assert (a.turn, a.epoch) == (b.turn, b.epoch)
        assert co.step(a.epoch, a.turn, actions=a.actions.cpu().numpy()) == 0
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert torch.equal(a.rewards, b.rewards) and torch.equal(a.actions, b.actions), t
        assert np.array_equal(b.world.grid.cpu().numpy(), co.grid) and np.array_equal(b.world.agent_pos.cpu().numpy(), co.pos), t
        assert np.array_equal(b.rewards.cpu().numpy(), co.rewards) and np.array_equal(b.world.total_reward.cpu().numpy(), co.total), t
        for k, (x, y) in enumerate(zip(a.agents, b.agents)):
            mx, my = x.model.memory, y.model.memory
            assert (mx.idx, mx.size) == (my.idx, my.size), t
            last = (my.idx - 1) % my.capacity
            assert np.array_equal(my.states[last].cpu().numpy().reshape(E, -1), co.obs[:, k].reshape(E, -1)), (t, k)
            for name in ("states", "actions", "rewards", "dones"):
                assert torch.equal(getattr(mx, name), getattr(my, name)), (t, k, name)
    assert cap.turns_replayed == 60
    assert b._engine.turn_state()[:2] == (b.epoch, b.turn)
    b.raise_on_status()
    # an agent class that overrides transition cannot be recorded: the eager loop stays
    c = _policy_env(8)

    class Custom(type(c.agents[0])):
        def transition(self, world):
            return super().transition(world)

    c.agents[0].__class__ = Custom
    assert c.capture_turn() is None
    c.take_turn()
    c.raise_on_status()


# ------------------------------------------------------------------ the specialiser: cache, fallback, plan == what is created
def test_specialised_instances_are_cached_and_fall_back(torch_cuda, tmp_path):
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(19, 23, 5, 3, spawn_prob=0.05, seed=8)
    N.set_option("jit_cache_dir", str(tmp_path))
    s0 = N.jit_stats()
    e1 = make_engine(ws, 40)
    s1 = N.jit_stats()
    want = N.plan(e1.config)["kernel"]
    assert "specialised=1" in e1.launch_info() and e1.launch_info().startswith(want), e1.launch_info()
    files = sorted(os.listdir(tmp_path))
    assert s1["compiled"] + s1["mem_hits"] + s1["disk_hits"] > s0["compiled"] + s0["mem_hits"] + s0["disk_hits"] and (files or s1["mem_hits"] > s0["mem_hits"])
    e2 = make_engine(ws, 40)                               # the same world again: the loaded function is reused, nothing is compiled
    s2 = N.jit_stats()
    assert s2["compiled"] == s1["compiled"] and s2["mem_hits"] > s1["mem_hits"]
    N.set_option("jit", 0)
    e3 = make_engine(ws, 40)                               # the prebuilt run-time-shape instance
    assert "specialised=0" in e3.launch_info() and "step_fast<true, 2, 6, 0, 0, 0" in e3.launch_info(), e3.launch_info()
    N.set_option("jit", 1)
    N.set_option("jit_cache_dir", "/proc/this/cannot/be/written")   # no disk cache: compiles (or reuses) all the same
    e4 = make_engine(treasurehunt_spec(19, 25, 5, 3, spawn_prob=0.05, seed=8), 40)
    assert "specialised=1" in e4.launch_info()
    engines = [e1, e2, e3]
    co = H.COracle(ws, 40, first_env_id=0)
    for e in engines:
        e.reset(0)
    co.reset(0)
    for t in range(1, 6):
        assert co.step(0, t, random_actions=True) == 0
        for e in engines:
            e.step(random_actions=True)
            assert_same(e, co, ctx=f"turn {t} {e.launch_info().split(' group')[0]}")
    with pytest.raises(ValueError):
        N.set_option("no_such_key", 1)
    with pytest.raises(ValueError):
        N.set_option("group", 48)
    with pytest.raises(ValueError):
        N.set_option("group", 16, engine=e1._h)           # a plan-shaping key on a live engine
    N.set_option("rows_mode", 1, engine=e1._h)             # a live key


# ------------------------------------------------------------------ long horizons, one per kernel family
def _long_horizon(torch, ws, E, T, check=(1, 2, 3, 10, 50, 100, 200, 350), first=3, epoch=1, expect=None, start=None):
    """T turns of random actions against the C oracle: every tensor (and the agents' types / facings where the rule keeps them) at
    the check points and at the end, then the same T turns as ONE sgw_rollout call (final state and last turn's outputs).
    ``start``: (grid, pos) of one env to begin every env from (worlds whose map the reset kernel does not build: Cleanup)."""
    def begin(eng, co):
        if start is None:
            eng.reset(epoch)
            if co is not None:
                co.reset(epoch)
            return
        g0, p0 = start
        eng.epoch = epoch
        eng.grid.copy_(torch.from_numpy(np.broadcast_to(g0, (E,) + g0.shape).copy()))
        eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(p0, (E,) + p0.shape).copy()))
        eng.total_reward.zero_()
        if co is not None:
            co.grid[...], co.pos[...], co.total[...] = g0, p0, 0

    def same(eng, co, ctx):
        assert_same(eng, co, ctx=ctx)
        if eng.agent_state is not None:
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), ctx + ": agent_state"
            assert np.array_equal(eng.state_at_pov.cpu().numpy(), co.state_at_pov), ctx + ": state_at_pov"
        if eng.agent_dir is not None:
            assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), ctx + ": agent_dir"

    eng, co = make_engine(ws, E, first=first), H.COracle(ws, E, first_env_id=first)
    if expect:
        assert expect in eng.launch_info(), eng.launch_info()
    begin(eng, co)
    marks = set(check) | {T}
    for t in range(1, T + 1):
        eng.step(random_actions=True)
        assert co.step(epoch, t, random_actions=True) == 0
        if t in marks:
            same(eng, co, f"turn {t}")
    assert eng.status() == 0
    roll = make_engine(ws, E, first=first)
    begin(roll, None)
    roll.rollout(T)
    same(roll, co, f"sgw_rollout of {T} turns")
    assert roll.status() == 0
    return eng


def _tag_spec(h, w, a, r):
    d, spec = H.load_golden("tag_9x9")
    ws = H.world_spec(spec)
    ws.height, ws.width, ws.num_agents, ws.vision_radius, ws.agent_type = h, w, a, r, [ws.agent_type[0]] * a
    return ws


@pytest.mark.parametrize("variant", ["plain", "walking", "staged"])
def test_long_horizon_config5_shape_on_step_big(torch_cuda, variant):
    """Config 5's shape (128x128x2, 64 agents, 11x11 windows, dense entities) for 500 turns on step_big: the plain instance, the
    walking workgroups, the staged windows."""
    from sorrel_amd.spec import treasurehunt_spec

    if variant == "walking":
        N.set_option("big_walk_blocks", 5)
    if variant == "staged":
        N.set_option("big_stage", 1)
    ws = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=2, dense_prob=0.25)
    eng = _long_horizon(torch_cuda, ws, 32, 500, expect="step_big<true, 2, 6, 5")
    info = eng.launch_info()
    assert ("false, true>" in info.split(" group")[0]) == (variant == "walking"), info
    assert ("big_stage=0" not in info) == (variant == "staged"), info


def test_long_horizon_tag_packed_and_big(torch_cuda):
    """Tag 11x11 / 5 agents / 9x9 on the packed kernel (two envs per wave) and Tag 128x128 / 64 agents on step_big<..., TAG>: 500
    turns -- the "it" token changes hands hundreds of times."""
    N.set_option("group", 32)
    _long_horizon(torch_cuda, _tag_spec(11, 11, 5, 4), 64, 500, expect="step_kernel<32, true, 1, 4, 1, 4, 11, 11>")
    N.set_option("group", None)
    _long_horizon(torch_cuda, _tag_spec(128, 128, 64, 4), 32, 500, expect="step_big<true, 1, 4, 4, false, false, true")


def test_long_horizon_cleanup_rules_kernel(torch_cuda):
    """Cleanup as shipped (21x31x3, 10 agents, 11x11 windows) on the RULES kernel for 500 turns: beam timers, pollution and apple
    cycles (sorrel/examples/cleanup/entities.py:43-105), facing, all-layer rewards."""
    d, spec = H.load_golden("cleanup_21x31_default")
    ws = H.world_spec(spec)
    _long_horizon(torch_cuda, ws, 48, 500, epoch=0, expect="step_fast<true, 3, 9, 5, 21, 31, false, true", start=(d["grid0"][0], d["pos0"][0]))


def test_long_horizon_treasurehunt_packed_and_own_entities(torch_cuda):
    """Treasurehunt 21x21 four envs to a wave, and a world with its own entity set (5 channels, 3 layers) on the instance
    specialised for it: 500 turns each."""
    from sorrel_amd.spec import treasurehunt_spec
    from tests.test_gpu_round3 import _move_world

    N.set_option("group", 16)
    _long_horizon(torch_cuda, treasurehunt_spec(21, 21, 2, 2, spawn_prob=0.02, seed=9), 64, 500, expect="step_kernel<16, true, 2, 6, 0, 2, 21, 21>")
    N.set_option("group", None)
    _long_horizon(torch_cuda, _move_world(26, 30, 3, 5, 7, 3, seed=11), 48, 500, expect="step_fast<true, 3, 5, 3, 26, 30")


# ------------------------------------------------------------------ row tails: what pov() appends, written by the engine
@pytest.mark.parametrize("which", ["tag", "cleanup"])
@pytest.mark.parametrize("memory", [0, 5], ids=["no_buffers", "replay_rows"])
def test_row_tails_written_by_the_engine_equal_the_host_concatenation(torch_cuda, which, memory):
    """TagAgent.pov appends the "it" flag, CleanupObservation.observe the positional code (sorrel/examples/tag/agents.py:57-65,
    sorrel/examples/cleanup/agents.py:52-60).  With sgw_bind_row_tail the engine writes them behind the window in each agent's row
    (its replay row where it has one) and sgw_act keeps Tag's flag current; the policy reads the finished row.  Everything a policy
    saw, every replay row and the state equal the host-concatenation path, turn after turn; Tag's flag equals the oracle's
    state_at_pov; the windows equal the oracle's."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel

    E = 29

    def factory():
        class Policy(BaseModel):
            def __init__(self, input_size, n_actions):
                super().__init__((int(np.prod(input_size)),), n_actions, memory_size=memory, num_envs=E, device="cuda:0")
                self.seen = []

            def take_action(self, state):
                self.seen.append(state.clone())
                s = state.reshape(state.shape[0], -1)
                return (s[:, ::3].sum(dim=1).long() * 7 + (s[:, -1] * 3).long() + (s > 0).sum(dim=1)) % self.action_space

        return Policy

    def make(in_kernel):
        if which == "tag":
            from sorrel_amd.entities import EmptyEntity
            from sorrel_amd.examples.tag.env import TagEnv
            from sorrel_amd.worlds import Gridworld

            cfg = {"agent": {"num_agents": 6, "vision_radius": 2, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 50}}
            env = TagEnv(Gridworld(8, 9, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=31), cfg, model_factory=factory())
        else:
            from tests.test_api_host import make_cleanup_env

            env = make_cleanup_env(E=E, seed=7, device="cuda:0", model_factory=factory())
        env.row_tails_in_kernel = in_kernel
        env._bind_row_tail()              # (the constructor has built the engine already)
        return env

    a, b = make(True), make(False)
    eng = a._ensure_engine()
    assert eng.row_tail == (1 if which == "tag" else 12) and b._ensure_engine().row_tail == 0
    co = H.COracle(a.compile_spec(), E)
    co.grid[...] = a.world.grid.cpu().numpy()
    co.pos[...] = a.world.agent_pos.cpu().numpy()
    if which == "tag":
        co.agent_state[...] = a.world.agent_state.cpu().numpy()
    nwin = int(np.prod(eng.spec.obs_shape[1:]))
    for t in range(1, 15):
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        assert co.step(0, t, actions=a.actions.cpu().numpy()) == 0
        assert torch.equal(a.actions, b.actions) and torch.equal(a.rewards, b.rewards), t
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert np.array_equal(a.world.grid.cpu().numpy(), co.grid) and np.array_equal(a.rewards.cpu().numpy(), co.rewards), t
        for k, (x, y) in enumerate(zip(a.agents, b.agents)):
            sx, sy = x.model.seen[-1], y.model.seen[-1]
            assert sx.shape == sy.shape == (E, nwin + eng.row_tail) and torch.equal(sx, sy), (t, k)
            assert np.array_equal(sx[:, :nwin].cpu().numpy(), co.obs[:, k].reshape(E, -1)), (t, k)
            if which == "tag":
                assert np.array_equal(sx[:, -1].cpu().numpy() != 0, co.state_at_pov[:, k] == eng.spec.tag_it_type), (t, k)
            if memory:
                assert torch.equal(x.model.memory.states, y.model.memory.states) and torch.equal(x.model.memory.actions, y.model.memory.actions), (t, k)
    a.raise_on_status()


def test_specialised_step_big_with_more_than_64_kib_of_lds(torch_cuda):
    """A 180x200x2 world keeps 72 KB per env in LDS: the specialised step_big instance (loaded as a module function) gets that much
    dynamic LDS, and agrees with the oracle and with the prebuilt instance."""
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(180, 200, 6, 4, spawn_prob=0.03, seed=12, dense_prob=0.1)
    for jit in (1, 0):
        N.set_option("jit", jit)
        eng, co = make_engine(ws, 6, first=1), H.COracle(ws, 6, first_env_id=1)
        assert "step_big<" in eng.launch_info() and f"specialised={jit}" in eng.launch_info() and int(eng.launch_info().split("lds=")[1].split()[0]) > 65536, eng.launch_info()
        eng.reset(0)
        co.reset(0)
        for t in range(1, 4):
            eng.step(random_actions=True)
            assert co.step(0, t, random_actions=True) == 0
            assert_same(eng, co, ctx=f"jit={jit} turn {t}")
        assert eng.status() == 0


# ------------------------------------------------------------------ SGW_ACT_QF32: the act takes the argmax of the policy's action values / explores
def _values_and_expected(rng, ws, E, A, first, epoch, turn, eps):
    """Random action values with ties, NaNs and infinities in some rows, and the actions the oracle's value_action takes from them."""
    from oracle import gridstep_oracle as O

    nact = len(ws.action_dy)
    q = rng.standard_normal((A, E, nact)).astype(np.float32)
    q[:, 0::7] = np.round(q[:, 0::7])                        # ties: the FIRST maximum wins
    q[:, 3::11, 1] = np.nan                                  # NaN counts as the maximum (np.argmax / torch.argmax)
    q[:, 5::13, nact - 1] = np.inf
    q[:, 6::17] = -np.inf
    spec = H.oracle_spec(ws)
    acts = np.zeros((E, A), dtype=np.uint8)
    for a in range(A):
        for e in range(E):
            acts[e, a] = O.value_action(spec, first + e, epoch, turn, a, q[a, e], eps[a])
    return q, acts


@pytest.mark.parametrize("world", ["treasurehunt", "tag"])
def test_action_values_argmax_and_exploration_in_the_act_vs_oracle(torch_cuda, world):
    """``agent_action`` = the policy's action values (float32 [E][num_actions], SGW_ACT_QF32): the act takes np.argmax of each row --
    ties, NaN, +-inf -- or, with probability epsilon[agent] (sgw_turn_epsilon: 0, 0.3, 1, ...), the engine's own uniform action for
    (env, turn, agent), as oracle.value_action (iqn.py:294-309 with the counter RNG) says; through sgw_turn_act (rings get the int64
    action taken) across an epoch change, and through sgw_act with the turn state set by the caller."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(14, 17, 5, 2, spawn_prob=0.05, seed=8, dense_prob=0.1) if world == "treasurehunt" else _tag_spec(12, 12, 5, 3)
    E, A, first, CAP = 45, 5, 11, 3
    eng, co = make_engine(ws, E, first=first), H.COracle(ws, E, first_env_id=first)
    nact = len(ws.action_dy)
    rings = [(None, torch.zeros((CAP, E), device="cuda:0"), torch.full((CAP, E), -1, dtype=torch.int64, device="cuda:0"), None, 0, 1) for _ in range(A)]
    eng.turn_bind(rings)
    eps = [0.0, 0.3, 1.0, 0.05, 0.7]
    for a in range(A):
        eng.turn_epsilon(eps[a], a)
    rng = np.random.default_rng(17)
    epoch = 2
    eng.reset(epoch); co.reset(epoch)
    eng.turn_set(epoch, 0)
    explored = 0
    for t in range(1, 11):
        if t == 6:
            epoch += 1
            eng.reset(epoch); co.reset(epoch)
            eng.turn_set(epoch, 0)
            eng.turn_epsilon(0.5)                    # every agent at once
            eps = [0.5] * A
        turn = t if t < 6 else t - 5
        q, acts = _values_and_expected(rng, ws, E, A, first, epoch, turn, eps)
        explored += int((acts != np.nanargmax(np.where(np.isnan(q), np.inf, q), axis=2).T).sum())
        assert co.step(epoch, turn, actions=acts) == 0
        qd = torch.from_numpy(q).cuda()
        row = eng.turn_state()[2][0]
        eng.turn_begin()
        for a in range(A):
            eng.turn_act(a, qd[a])
        eng.turn_end(commit_windows=False)
        assert_same(eng, co, ("grid", "pos", "actions", "rewards", "total"), ctx=f"{world} turn {t}")
        if eng.agent_state is not None:
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), t
        for a in range(A):
            assert np.array_equal(rings[a][2][row].cpu().numpy(), acts[:, a].astype(np.int64)), (t, a)
    assert explored > 50                              # (the epsilon branch was taken, and differs from the argmax, often)
    # sgw_act: no turn of its own -- the caller sets the device's turn state; epsilon 0 = plain argmax
    eng.turn_bind(None)
    for turn in (6, 7):
        for mode in ("explore", "greedy"):
            eps = [0.4] * A if mode == "explore" else [0.0] * A
            eng.turn_epsilon(eps[0])
            q, acts = _values_and_expected(rng, ws, E, A, first, epoch, turn, eps)
            if mode == "greedy":
                assert np.array_equal(acts, np.nanargmax(np.where(np.isnan(q), np.inf, q), axis=2).T.astype(np.uint8))
            assert co.step(epoch, turn, actions=acts, sweep=False) == 0
            qd = torch.from_numpy(q).cuda()
            eng.turn_set(epoch, turn - 1)
            eng.step(sweep=False, no_move=True, turn=turn)
            for a in range(A):
                eng.act(a, eng.window_rows(None), action=qd[a])
            assert_same(eng, co, ("grid", "pos", "actions", "rewards", "total"), ctx=f"{world} sgw_act turn {turn} {mode}")
    assert eng.status() == 0
    with pytest.raises(ValueError):
        eng.turn_act(0, torch.zeros((E, nact + 1), device="cuda:0"))
    with pytest.raises(ValueError):
        eng.turn_epsilon(1.5)


def test_captured_turn_with_action_values_follows_a_decaying_epsilon(torch_cuda):
    """Agents whose get_action returns the model's action VALUES: Environment hands them to the act launch (no argmax launch, exploration
    in-kernel at the model's epsilon).  A recorded turn -- one node less per agent -- equals the eager loop and the oracle over 40 turns
    while epsilon decays every few turns and across an epoch reset; the buffers hold the actions TAKEN."""
    torch = torch_cuda
    from oracle import gridstep_oracle as O
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E = 29

    class ValuePolicy(BaseModel):
        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=5, num_envs=E, device="cuda:0")
            n = int(np.prod(input_size))
            self.weight = torch.randn((n, action_space), generator=torch.Generator().manual_seed(99 + n)).cuda()
            self.epsilon = 0.6

        def take_action(self, state):
            return state.reshape(state.shape[0], -1) @ self.weight          # [E, n_actions] float32: values, not actions

    a, b = (make_env(12, 16, 4, 2, E, p=0.05, seed=3, model_factory=ValuePolicy) for _ in range(2))
    cap = b.capture_turn(warmup=2)
    assert cap is not None, getattr(b, "capture_error", None)
    for _ in range(2):
        a.take_turn()
    ws = a._engine.spec
    spec = H.oracle_spec(ws)
    co = H.COracle(ws, E, first_env_id=0)

    def sync_oracle():
        co.grid[...] = a.world.grid.cpu().numpy()
        co.pos[...] = a.world.agent_pos.cpu().numpy()
        co.total[...] = a.world.total_reward.cpu().numpy()

    sync_oracle()
    took_random = 0
    for t in range(40):
        if t == 25:
            a.reset(); b.reset()
            sync_oracle()
        if t % 4 == 0:
            for env in (a, b):
                for ag in env.agents:
                    ag.model.epsilon *= 0.8
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        # the oracle, agent by agent: the window -> the same linear values (float32 on the host) -> value_action -> act
        assert co.step(a.epoch, a.turn, actions=b.actions.cpu().numpy()) == 0
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert torch.equal(a.actions, b.actions) and torch.equal(a.rewards, b.rewards), t
        assert np.array_equal(b.world.grid.cpu().numpy(), co.grid) and np.array_equal(b.rewards.cpu().numpy(), co.rewards), t
        for k, (x, y) in enumerate(zip(a.agents, b.agents)):
            mx, my = x.model.memory, y.model.memory
            assert (mx.idx, mx.size) == (my.idx, my.size), t
            for name in ("states", "actions", "rewards", "dones"):
                assert torch.equal(getattr(mx, name), getattr(my, name)), (t, k, name)
            last = (my.idx - 1) % my.capacity
            taken = my.actions[last].cpu().numpy().reshape(-1)
            assert np.array_equal(taken, b.actions[:, k].cpu().numpy().astype(np.int64)), (t, k)
            # ... and they are what value_action takes from the values of the window the agent saw
            q = (my.states[last].reshape(E, -1) @ y.model.weight).cpu().numpy()
            want = np.array([O.value_action(spec, e, b.epoch, b.turn, k, q[e], y.model.epsilon) for e in range(E)])
            greedy = q.argmax(axis=1)
            margin = np.sort(q, axis=1)
            sure = (margin[:, -1] - margin[:, -2]) > 1e-3          # (the host's matmul may round differently from the device's: skip near-ties)
            assert np.array_equal(taken[sure], want[sure]), (t, k)
            took_random += int((taken[sure] != greedy[sure]).sum())
    assert took_random > 20
    assert cap.turns_replayed == 40
    b.raise_on_status()


def test_run_experiment_with_recorded_turns_equals_the_eager_loop(torch_cuda):
    """``Environment.capture_turns = True``: run_experiment records the policy turn in its first epoch and replays it for every later
    turn of every epoch -- across resets, a model that clears its memory at the start of some epochs (the rings are bound again) and an
    epsilon that decays per epoch (in-kernel exploration follows it) -- with the history, world and buffers of the eager loop."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E = 21

    class Model(BaseModel):
        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=7, num_envs=E, device="cuda:0")
            n = int(np.prod(input_size))
            self.weight = torch.randn((n, action_space), generator=torch.Generator().manual_seed(5 + n)).cuda()
            self.epsilon = 0.5
            self.trained = 0

        def take_action(self, state):
            return state.reshape(state.shape[0], -1) @ self.weight

        def start_epoch_action(self, epoch=0, **kw):
            if epoch % 2 == 1:
                self.memory.clear()

        def train_step(self):
            self.trained += 1
            self.weight.mul_(0.97)            # in place: the recorded forward pass reads the same storage
            return float(self.memory.rewards.sum())

    envs = []
    for capture in (False, True):
        env = make_env(13, 12, 3, 2, E, p=0.06, seed=9, model_factory=Model, max_turns=11, extra_model={"epsilon_decay": 0.2})
        env.capture_turns = capture
        hist = env.run_experiment(epochs=3, logging=False, all_reduce=False)
        torch.cuda.synchronize()
        envs.append((env, hist))
    (a, ha), (b, hb) = envs
    assert b._captured is not None and b._captured.turns_replayed == 4 * 11 - 2, getattr(b, "capture_error", None)
    assert a._captured is None
    assert ha == hb
    for name in ("grid", "agent_pos", "total_reward"):
        assert torch.equal(getattr(a.world, name), getattr(b.world, name)), name
    assert torch.equal(a.actions, b.actions) and torch.equal(a.rewards, b.rewards)
    for x, y in zip(a.agents, b.agents):
        assert x.model.trained == y.model.trained == 4 and x.model.epsilon == y.model.epsilon < 0.5
        mx, my = x.model.memory, y.model.memory
        assert (mx.idx, mx.size) == (my.idx, my.size)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(mx, name), getattr(my, name)), name
    b.raise_on_status()


def test_captured_turn_with_frame_stacks(torch_cuda):
    """Memories with ``n_frames = 3`` (the reference's Cleanup / IQN configs stack frames: ``Buffer.current_state``,
    sorrel/buffers.py:143-154, in front of the window): in a recorded turn the previous frames are gathered by the device's own row
    count (sgw_turn_prev_rows) -- 30 turns with wrap-arounds of the 5-row rings, an epoch reset and an ``add_empty`` at its start equal
    the eager loop; the gather itself against the ring for every position of the row counter."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E = 19

    class Stacked(BaseModel):
        n_frames = 3

        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=0, num_envs=E, device="cuda:0")
            n = int(np.prod(input_size))
            self.memory = Buffer(capacity=5, obs_shape=(n,), n_frames=3, num_envs=E, device="cuda:0")
            self.weight = torch.randn((3 * n, action_space), generator=torch.Generator().manual_seed(7 + n)).cuda()

        def take_action(self, state):
            assert state.shape[1] == self.weight.shape[0]
            return (state @ self.weight).argmax(dim=1)

        def start_epoch_action(self, **kw):
            self.memory.add_empty()

    a, b = (make_env(12, 13, 3, 2, E, p=0.05, seed=4, model_factory=Stacked) for _ in range(2))
    cap = b.capture_turn(warmup=2)
    assert cap is not None, getattr(b, "capture_error", None)
    for _ in range(2):
        a.take_turn()
    for t in range(30):
        if t == 17:
            for env in (a, b):
                env.reset()
                for ag in env.agents:
                    ag.model.start_epoch_action()
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert torch.equal(a.actions, b.actions) and torch.equal(a.rewards, b.rewards), t
        for k, (x, y) in enumerate(zip(a.agents, b.agents)):
            mx, my = x.model.memory, y.model.memory
            assert (mx.idx, mx.size) == (my.idx, my.size), t
            for name in ("states", "actions", "rewards", "dones"):
                assert torch.equal(getattr(mx, name), getattr(my, name)), (t, k, name)
            # the gather, against the host's indices
            got = b._engine.turn_prev_rows(k, 2, torch.empty_like(my.states[:2]))
            assert torch.equal(got, my.states[[(my.idx - 2) % 5, (my.idx - 1) % 5]]), (t, k)
    assert cap.turns_replayed == 30
    b.raise_on_status()
    eng = b._engine
    with pytest.raises(ValueError):
        eng.turn_prev_rows(0, 6, torch.empty((6, E, my.states.shape[2]), device="cuda:0"))      # more rows than the ring has
    # agents that SHARE a frame-stacking ring: recorded only where the windows reach the ring as the turn goes (test_gpu_round5.py)
    c = make_env(12, 13, 3, 2, E, p=0.05, seed=4, model_factory=Stacked)
    for ag in c.agents[1:]:
        ag.model.memory = c.agents[0].model.memory
    c.capture_layout = "tensor"
    assert c.capture_turn() is None and "rows" in str(c.capture_error)
    c.take_turn()


@pytest.mark.parametrize("which", ["tag", "cleanup"])
def test_captured_turn_of_the_tag_and_cleanup_examples(torch_cuda, which):
    """The shipped Tag and Cleanup agents -- whose pov appends to the window -- in a recorded turn: the engine writes window + tail into
    the row each policy reads AND into its replay row (the "it" flag of an agent tagged before its own pov in both), so nothing is
    concatenated or copied on the host.  35 turns across ring wrap-arounds and a reset equal the eager loop; the rows a policy read
    equal the oracle's windows, Tag's flag the oracle's state_at_pov."""
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

    def make():
        if which == "tag":
            from sorrel_amd.entities import EmptyEntity
            from sorrel_amd.examples.tag.env import TagEnv
            from sorrel_amd.worlds import Gridworld

            cfg = {"agent": {"num_agents": 6, "vision_radius": 2, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 50}}
            return TagEnv(Gridworld(8, 9, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=31), cfg, model_factory=Policy)
        from tests.test_api_host import make_cleanup_env

        return make_cleanup_env(E=E, seed=7, device="cuda:0", model_factory=Policy)

    a, b = make(), make()
    eng = b._ensure_engine()
    tail = 1 if which == "tag" else 12
    assert eng.row_tail == tail
    cap = b.capture_turn(warmup=2)
    assert cap is not None, getattr(b, "capture_error", None)
    assert b._capture_rows is not None
    for _ in range(2):
        a.take_turn()
    co = H.COracle(a.compile_spec(), E)

    def sync_oracle():
        co.grid[...] = a.world.grid.cpu().numpy()
        co.pos[...] = a.world.agent_pos.cpu().numpy()
        co.total[...] = a.world.total_reward.cpu().numpy()
        if which == "tag":
            co.agent_state[...] = a.world.agent_state.cpu().numpy()
        else:
            co.agent_dir[...] = a.world.agent_dir.cpu().numpy()

    sync_oracle()
    nwin = int(np.prod(eng.spec.obs_shape[1:]))
    for t in range(35):
        if t == 20:
            a.reset(); b.reset()
            sync_oracle()
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        assert co.step(a.epoch, a.turn, actions=a.actions.cpu().numpy()) == 0
        assert torch.equal(a.actions, b.actions) and torch.equal(a.rewards, b.rewards), t
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert np.array_equal(b.world.grid.cpu().numpy(), co.grid) and np.array_equal(b.rewards.cpu().numpy(), co.rewards), t
        for k, (x, y) in enumerate(zip(a.agents, b.agents)):
            mx, my = x.model.memory, y.model.memory
            assert (mx.idx, mx.size) == (my.idx, my.size), t
            for name in ("states", "actions", "rewards", "dones"):
                assert torch.equal(getattr(mx, name), getattr(my, name)), (t, k, name)
            row = my.states[(my.idx - 1) % my.capacity].reshape(E, -1)
            assert row.shape[1] == nwin + tail
            assert np.array_equal(row[:, :nwin].cpu().numpy(), co.obs[:, k].reshape(E, -1)), (t, k)
            if which == "tag":
                assert np.array_equal(row[:, -1].cpu().numpy() != 0, co.state_at_pov[:, k] == eng.spec.tag_it_type), (t, k)
    assert cap.turns_replayed == 35
    b.raise_on_status()


def test_generate_memories_with_recorded_turns_writes_the_same_files(torch_cuda, tmp_path):
    """``capture_turns`` in generate_memories (sorrel/environment.py:213-300): the per-agent replay files of three games of nine turns
    -- states, actions, rewards, dones -- are byte for byte what the eager loop writes."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E = 11

    class Model(BaseModel):
        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=40, num_envs=E, device="cuda:0")
            n = int(np.prod(input_size))
            self.weight = torch.randn((n, action_space), generator=torch.Generator().manual_seed(11 + n)).cuda()

        def take_action(self, state):
            return (state.reshape(state.shape[0], -1) @ self.weight).argmax(dim=1)

    files = []
    for capture in (False, True):
        env = make_env(11, 12, 3, 2, E, p=0.06, seed=2, model_factory=Model, max_turns=9)
        env.capture_turns = capture
        paths = env.generate_memories(num_games=3, output_dir=tmp_path / ("rec" if capture else "eager"))
        assert (env._captured is not None) == capture, getattr(env, "capture_error", None)
        files.append([dict(np.load(p)) for p in paths])
    for fa, fb in zip(*files):
        assert set(fa) == set(fb)
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), k
    assert files[0][0]["states"].shape[0] > 0

"""Round-3 GPU tests: the row-load phase kernel (``phase_rows``) that serves policy-driven phases of one-hot plain-move
worlds of any size; the long horizon at the benchmark's own shape; the whole-map (``full_view``) observation."""
import os

import numpy as np
import pytest

from tests import helpers as H
from sorrel_amd import _native as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no silent CPU fallback)")
    return torch


def make_engine(ws, E, first=0, **kw):
    from sorrel_amd.engine import GridEngine

    return GridEngine(ws, E, device="cuda:0", first_env_id=first, **kw)


def _move_world(h, w, layers, channels, a, r, seed, zA=None):
    """A plain-mover world with `layers` layers and `channels` one-hot channels (every instance of phase_rows is keyed by
    layers x ceil(channels / 4) x radius): walls around every layer, a spawner, a few pick-ups with values."""
    from sorrel_amd.spec import WorldSpec, action_deltas

    T = max(6, min(channels + 1, 12))
    app = np.zeros((T, channels))
    for t in range(1, T):
        app[t, (t * 5 + 1) % channels] = 1.0
    zA = layers - 1 if zA is None else zA
    dy, dx = action_deltas(["up", "down", "left", "right", "stay"])
    rule = [1] + [0] * (T - 1)
    return WorldSpec(height=h, width=w, layers=layers, num_agents=a, vision_radius=r, num_channels=channels, agent_layer=zA,
                     default_type=0, fill_type=1, action_dy=dy, action_dx=dx, agent_type=[T - 1] * a,
                     type_value=[0.0, -1.0, 10.0, 5.0, -10.0] + [1.0] * (T - 6) + [0.0],
                     type_passable=[1, 0, 1, 1, 1] + [1] * (T - 6) + [0], type_rule=rule,
                     spawn_prob=[0.05] + [0.0] * (T - 1), spawn_choices=[[2, 3, 4]] + [[] for _ in range(T - 1)],
                     appearance=app, seed=seed, layer_fill_type=[0] * layers, layer_border_type=[1] * layers,
                     dense_prob=0.3, dense_choices=[2, 3, 4])


ROWS_CASES = [
    # (h, w, layers, channels, agents, radius, envs)   -- envs deliberately not multiples of the envs a workgroup carries
    (32, 32, 2, 6, 8, 3, 77),      # BASELINE configs 3 / 4
    (16, 16, 2, 6, 4, 2, 201),     # BASELINE config 2
    (128, 128, 2, 6, 24, 5, 7),    # BASELINE config 5's shape
    (7, 7, 2, 6, 9, 3, 65),        # the smallest world a 7x7 window allows, crowded: every window hangs over every edge
    (5, 6, 2, 6, 4, 2, 33),        # rows shorter than one 8-byte load
    (9, 13, 1, 4, 6, 4, 50),       # one layer, one counter word, 9x9 windows (two 8-byte chunks per row)
    (11, 11, 1, 3, 5, 5, 19),      # maximum radius for the size
    (3, 3, 1, 2, 1, 1, 130),       # the smallest world there is (9 cells)
    (12, 10, 3, 7, 5, 2, 41),      # three layers, agents on the middle one
    (21, 31, 3, 8, 6, 3, 23),
    (24, 20, 2, 3, 7, 1, 300),     # 3x3 windows: 16 envs per wave
    (40, 36, 2, 8, 10, 4, 29),
]


@pytest.mark.parametrize("case", ROWS_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}_C{c[3]}_A{c[4]}_r{c[5]}" for c in ROWS_CASES])
def test_phase_rows_policy_turn_vs_oracle(torch_cuda, case, monkeypatch):
    """A policy-driven turn in 1 + A launches (sweep + window of agent 0; then per agent: move it, render the next) on the
    row-load phase kernel: every window an agent's policy would read, every reward, the grid, positions and totals against
    the C oracle, turn after turn; the packed destination (a replay row) receives the same windows; a phase that renders
    nothing, the plain per-agent step (own window BEFORE the move) and sgw_observe of one agent take the kernel too."""
    torch = torch_cuda
    h, w, layers, channels, a_, r_, E = case
    ws = _move_world(h, w, layers, channels, a_, r_, seed=h * 100 + w, zA=1 if layers == 3 else None)
    A = ws.num_agents
    eng, co = make_engine(ws, E, first=11), H.COracle(ws, E, first_env_id=11)
    assert "phase_rows<" in eng.launch_info(), eng.launch_info()
    eng.reset(0)
    co.reset(0)
    per_env = int(np.prod(ws.obs_shape[1:]))
    rng = np.random.default_rng(5)
    for t in range(1, 7):
        acts_np = rng.integers(0, len(ws.action_dy), size=(E, A), dtype=np.uint8)
        acts = torch.from_numpy(acts_np).cuda()
        assert co.step(0, t, actions=acts_np) == 0
        seen = torch.zeros_like(eng.obs)
        rew = torch.zeros_like(eng.rewards)
        packed = t % 2 == 0
        rows = [torch.full((E, per_env), -3.0, device="cuda:0") for _ in range(A)] if packed else None
        eng.obs.fill_(-7.0)
        eng.step(acts, sweep=True, agent_begin=0, agent_end=0, turn=t, obs_next=True, obs_next_out=rows[0] if packed else None)
        for a in range(A):
            seen[:, a] = rows[a].view(E, *ws.obs_shape[1:]) if packed else eng.obs[:, a]
            nxt = a + 1 < A
            eng.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, turn=t, obs_next=nxt, write_obs=False,
                     obs_next_out=rows[a + 1] if (packed and nxt) else None)
            rew[:, a] = eng.rewards[:, a]
        torch.cuda.synchronize()
        assert np.array_equal(seen.cpu().numpy(), co.obs), f"turn {t}: windows differ from the oracle"
        assert np.array_equal(rew.cpu().numpy(), co.rewards), f"turn {t}: rewards"
        assert np.array_equal(eng.grid.cpu().numpy(), co.grid), f"turn {t}: grid"
        assert np.array_equal(eng.agent_pos.cpu().numpy(), co.pos), f"turn {t}: positions"
        assert np.array_equal(eng.total_reward.cpu().numpy(), co.total), f"turn {t}: total_reward"
        if packed:
            assert bool((eng.obs == -7.0).all()), "the packed calls must not touch the observation tensor"
    # the plain per-agent step: ONE call writes the mover's own (pre-move) window and moves it; sgw_observe of one agent
    for t in range(7, 10):
        acts_np = rng.integers(0, len(ws.action_dy), size=(E, A), dtype=np.uint8)
        assert co.step(0, t, actions=acts_np) == 0
        acts = torch.from_numpy(acts_np).cuda()
        eng.obs.fill_(-3.0)
        eng.step(acts, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t)
        for a in range(A):
            eng.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, turn=t)
        torch.cuda.synchronize()
        assert np.array_equal(eng.obs.cpu().numpy(), co.obs) and np.array_equal(eng.grid.cpu().numpy(), co.grid), t
        assert np.array_equal(eng.total_reward.cpu().numpy(), co.total) and np.array_equal(eng.agent_pos.cpu().numpy(), co.pos), t
    eng.obs.fill_(-1.0)
    co.obs.fill(-1.0)
    for a in (0, A - 1):
        eng.observe(a, a + 1)
        co.observe(a, a + 1)
    torch.cuda.synchronize()
    assert np.array_equal(eng.obs.cpu().numpy(), co.obs), "sgw_observe of one agent"
    assert eng.status() == 0


def test_phase_rows_flags_bad_input_like_the_other_kernels(torch_cuda):
    """Bad action index, a move off an un-walled map edge and a garbage position raise the same status bits on the
    row-load phase kernel; nothing is written outside the env."""
    torch = torch_cuda
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.0, seed=1)
    E = 50
    eng = make_engine(ws, E)
    eng.reset(0)
    guard = eng.grid.clone()
    acts = torch.full((E, 4), 9, dtype=torch.uint8, device="cuda:0")           # no such action
    eng.step(acts, sweep=False, agent_begin=1, agent_end=2, obs_next=True, write_obs=False, turn=1)
    assert eng.status() & N.STATUS_BAD_ACTION
    assert torch.equal(eng.grid, guard)
    eng.agent_pos[:, 2, 0] = 200                                               # garbage row
    eng.step(torch.zeros_like(acts), sweep=False, agent_begin=1, agent_end=2, obs_next=True, write_obs=False, turn=1)
    assert eng.status() & N.STATUS_BAD_POS
    eng.reset(0)
    eng.grid[:, 1, 0, :] = 0                                                   # open the top wall, put agent 0 on the edge row
    eng.grid[:, 1][torch.arange(E), eng.agent_pos[:, 0, 0].long(), eng.agent_pos[:, 0, 1].long()] = 0
    eng.agent_pos[:, 0, 0] = 0
    eng.agent_pos[:, 0, 1] = 5
    eng.grid[:, 1, 0, 5] = ws.agent_type[0]
    eng.step(torch.zeros_like(acts), sweep=False, agent_begin=0, agent_end=1, obs_next=True, write_obs=False, turn=1)   # "up" off the map
    assert eng.status() & N.STATUS_OOB_MOVE
    assert bool((eng.agent_pos[:, 0, 0] == 0).all())


# ------------------------------------------------------------------ sgw_observe_rows + sgw_act: windows rendered once, repaired by the movers
PATCH_CASES = ROWS_CASES + [
    (14, 18, 2, 5, 6, 3, 37, "float"),     # a non one-hot appearance table: sgw_observe renders, sgw_act repairs in float64
    (20, 16, 2, 6, 7, 2, 45, "u8"),        # compact uint8 windows
    (26, 22, 2, 6, 40, 2, 11, "plain"),    # 40 agents: a wave per env in sgw_act
    (12, 30, 1, 4, 20, 3, 14, "plain"),    # 20 agents: 32 lanes per env
]


@pytest.mark.parametrize("case", PATCH_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}_C{c[3]}_A{c[4]}_r{c[5]}{'_' + c[7] if len(c) > 7 else ''}" for c in PATCH_CASES])
def test_windows_rendered_once_and_repaired_by_sgw_act_vs_oracle(torch_cuda, case):
    """The policy-driven turn of round 3: the sweep alone, every agent's window once (sgw_observe_rows into replay-like rows
    or tensor slots; sgw_observe where there is no row-load instance), then per agent sgw_act = move + repair of the later
    agents' windows.  What each agent's policy would read (its window at the moment BEFORE its own act), every reward,
    the grid, positions and totals equal the C oracle's sequential take_turn."""
    torch = torch_cuda
    from sorrel_amd import _native as N

    h, w, layers, channels, a_, r_, E = case[:7]
    kind = case[7] if len(case) > 7 else "plain"
    ws = _move_world(h, w, layers, channels, a_, r_, seed=7 * h + w, zA=1 if layers == 3 else None)
    if kind == "float":
        ws.appearance = ws.appearance * 1.0
        ws.appearance[2, 0] = 2.5
        ws.appearance[3, 1] = 0.25
        ws.appearance[ws.num_types - 1, 2] = 3.0          # the agents themselves: every move changes float cells
    A = ws.num_agents
    kw = dict(obs_dtype=torch.uint8) if kind == "u8" else {}
    eng, co = make_engine(ws, E, first=3, **kw), H.COracle(ws, E, first_env_id=3)
    caps = eng.capabilities()
    assert caps & N.CAP_ACT
    assert bool(caps & N.CAP_OBSERVE_ROWS) == (kind == "plain"), (caps, eng.launch_info())
    eng.reset(0)
    co.reset(0)
    per_env = int(np.prod(ws.obs_shape[1:]))
    rng = np.random.default_rng(9)
    for t in range(1, 8):
        acts_np = rng.integers(0, len(ws.action_dy), size=(E, A), dtype=np.uint8)
        assert co.step(0, t, actions=acts_np) == 0
        eng.actions.copy_(torch.from_numpy(acts_np))
        own_rows = (caps & N.CAP_OBSERVE_ROWS) and t % 2 == 0
        dests = [torch.full((E, per_env), -3.0, device="cuda:0") for _ in range(A)] if own_rows else None
        eng.obs.fill_(99 if kind == "u8" else -7.0)
        rows = eng.window_rows(dests)
        if t % 3 == 0 and not own_rows:
            eng.step(eng.actions, sweep=True, no_move=True, turn=t)                                 # sweep + every window in ONE launch (SGW_STEP_NO_MOVE)
        else:
            eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t)  # the sweep alone
            if caps & N.CAP_OBSERVE_ROWS:
                eng.observe_rows(rows)
            else:
                eng.observe()
        seen = torch.zeros_like(eng.obs)
        rew = torch.zeros_like(eng.rewards)
        want_actions = eng.actions.clone()
        for a in range(A):
            seen[:, a] = dests[a].view(E, *ws.obs_shape[1:]) if own_rows else eng.obs[:, a]      # what agent a's policy reads
            if a % 2:       # the policy's own output tensor (int64 / int32), rewards and actions also into replay-like rows
                mine = want_actions[:, a].to(torch.int64 if a % 4 == 1 else torch.int32).contiguous()
                eng.actions[:, a] = 77
                rrow, arow = torch.full((E,), -1.0, device="cuda:0"), torch.full((E,), -1, dtype=torch.int64, device="cuda:0")
                out = eng.act(a, rows, action=mine, reward_row=rrow, action_row=arow)
                assert out.data_ptr() == rrow.data_ptr()
                assert torch.equal(rrow, eng.rewards[:, a]) and torch.equal(arow, want_actions[:, a].long())
            else:
                eng.act(a, rows)
            rew[:, a] = eng.rewards[:, a]
        torch.cuda.synchronize()
        assert torch.equal(eng.actions, want_actions), "actions[:, a] records what was taken"
        assert np.array_equal(seen.cpu().numpy().astype(np.float32), co.obs), f"turn {t}: windows differ from the oracle"
        assert np.array_equal(rew.cpu().numpy(), co.rewards), f"turn {t}: rewards"
        assert np.array_equal(eng.grid.cpu().numpy(), co.grid), f"turn {t}: grid"
        assert np.array_equal(eng.agent_pos.cpu().numpy(), co.pos), f"turn {t}: positions"
        assert np.array_equal(eng.total_reward.cpu().numpy(), co.total), f"turn {t}: total_reward"
        if own_rows:
            assert bool((eng.obs == -7.0).all()), "per-agent destinations: the observation tensor must stay untouched"
    assert eng.status() == 0


def test_sgw_act_and_observe_rows_reject_what_they_cannot_serve(torch_cuda):
    torch = torch_cuda
    from sorrel_amd import _native as N

    d, spec = H.load_golden("tag_9x9")
    tag = make_engine(H.world_spec(spec), 8)
    assert tag.capabilities() & ~N.CAP_SWEEP_ROWS == N.CAP_ACT | N.CAP_OBSERVE_ROWS   # (round 4: observe_rows renders windows whatever the agents' act rule is; round 6: where the
    # world is on a wave-per-env instance -- a batch of 8 envs of a 9x9 map may be packed instead -- the fused sweep + rows launch as well; never the speculative resolve: Tag agents)
    ws = _move_world(16, 16, 2, 6, 4, 2, seed=1)
    u8 = make_engine(ws, 8, obs_dtype=torch.uint8)
    assert u8.capabilities() == N.CAP_ACT                # the row-load kernels write float32 windows only
    with pytest.raises(ValueError):
        u8.observe_rows(u8.window_rows(None))
    eng = make_engine(ws, 8)
    per_env = int(np.prod(ws.obs_shape[1:]))
    with pytest.raises(ValueError):                       # a destination that is not exactly one window per env
        eng.window_rows([torch.zeros((8, per_env + 1), device="cuda:0") for _ in range(4)])
    with pytest.raises(ValueError):
        eng.window_rows([torch.zeros((8, per_env), device="cuda:0")] * 3)
    rows = eng.window_rows(None)
    with pytest.raises(ValueError):
        eng.act(4, rows)
    with pytest.raises(ValueError):
        eng.observe_rows(rows, 2, 2)


def test_environment_policy_turn_protocols_agree_and_overridden_take_turn_is_called(torch_cuda, tmp_path):
    """Environment.take_turn with policy models: the patched-window protocol == the 1 + A protocol, turn after turn; a
    world mutated by host code in the middle of a turn falls back to rendering on demand; and a subclass that overrides
    take_turn gets it called every turn by run_experiment / generate_memories even with device-random models (the
    reference's loop always goes through take_turn, sorrel/environment.py:160-166)."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel, RandomModel
    from tests.test_gpu_round2 import make_env

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


@pytest.mark.parametrize("mode", ["flat", "pairs", "singles", "runs", "offset_rows", "agent_range"])
def test_observe_rows_emit_modes(torch_cuda, mode, monkeypatch):
    """The staged windows leave as one contiguous aligned run per wave (tensor slots: agents of consecutive envs; per-agent
    rows: consecutive envs of one agent), as float2 runs per window, or as single floats -- whatever the alignment of the
    destinations allows; every form writes the same windows and nothing else."""
    torch = torch_cuda
    if mode == "pairs":
        N.set_option("rows_mode", 2)
    if mode == "singles":
        N.set_option("rows_mode", 1)
    if mode == "runs":        # (round 4: what unaligned destinations take by default -- aligned float4 runs, the ends element by element)
        N.set_option("rows_mode", 3)
    for (h, w, layers, channels, a_, r_, E) in [(32, 32, 2, 6, 8, 3, 77), (16, 16, 2, 6, 4, 2, 201), (9, 13, 1, 3, 5, 1, 50), (40, 36, 2, 8, 3, 5, 13)]:
        ws = _move_world(h, w, layers, channels, a_, r_, seed=3)
        A = ws.num_agents
        eng, co = make_engine(ws, E), H.COracle(ws, E)
        eng.reset(0)
        co.reset(0)
        for t in range(1, 3):
            eng.step(random_actions=True, turn=t)
            co.step(0, t, random_actions=True)
        co.obs.fill(-1.0)
        co.observe()
        per_env = int(np.prod(ws.obs_shape[1:]))
        # tensor slots
        eng.obs.fill_(-5.0)
        if mode == "agent_range":
            eng.observe_rows(eng.window_rows(None), 1, A - 1)
            torch.cuda.synchronize()
            got = eng.obs.cpu().numpy()
            assert np.array_equal(got[:, 1:A - 1], co.obs[:, 1:A - 1]) and (got[:, 0] == -5.0).all() and (got[:, A - 1] == -5.0).all()
            continue
        eng.observe_rows(eng.window_rows(None))
        torch.cuda.synchronize()
        assert np.array_equal(eng.obs.cpu().numpy(), co.obs), (mode, h, w)
        # per-agent rows, with a guard element on either side of every destination
        pad = 1 if mode == "offset_rows" else 4
        store = [torch.full((E * per_env + 2 * pad,), -9.0, device="cuda:0") for _ in range(A)]
        dests = [s[pad:pad + E * per_env].view(E, per_env) for s in store]
        eng.observe_rows(eng.window_rows(dests))
        torch.cuda.synchronize()
        for a in range(A):
            assert np.array_equal(dests[a].view(E, *ws.obs_shape[1:]).cpu().numpy(), co.obs[:, a]), (mode, h, w, a)
            assert bool((store[a][:pad] == -9.0).all()) and bool((store[a][-pad:] == -9.0).all()), "wrote outside the destination"


# ------------------------------------------------------------------ long horizon at the benchmark's own shape
def test_config3_shape_long_horizon_vs_oracle(torch_cuda):
    """bench.py times launches at turns ~1 500 of a saturated world; this plays 512 envs of the config-3 shape (32x32x2,
    8 agents, 7x7 windows, spawn 0.005) for 1 600 turns: turn by turn against the C oracle at check points (every tensor,
    observations included) and at the end, and once more as ONE sgw_rollout call."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=0)
    E, T = 512, 1600
    eng, co = make_engine(ws, E, first=1000), H.COracle(ws, E, first_env_id=1000)
    one_call = make_engine(ws, E, first=1000)
    for e in (eng, one_call):
        e.reset(0)
    co.reset(0)
    for t in range(1, T + 1):
        eng.step(random_actions=True, turn=t)
        assert co.step(0, t, random_actions=True) == 0
        if t % 200 == 0 or t in (1, 2, T - 1):
            torch.cuda.synchronize()
            for name, mine, ref in (("grid", eng.grid, co.grid), ("pos", eng.agent_pos, co.pos), ("actions", eng.actions, co.actions),
                                    ("obs", eng.obs, co.obs), ("rewards", eng.rewards, co.rewards), ("total", eng.total_reward, co.total)):
                assert np.array_equal(mine.cpu().numpy(), ref), f"turn {t}: {name} differs from the oracle"
    one_call.rollout(T)
    torch.cuda.synchronize()
    for name in ("grid", "agent_pos", "total_reward", "obs", "rewards", "actions"):
        assert torch.equal(getattr(eng, name), getattr(one_call, name)), f"sgw_rollout({T}) vs turn by turn: {name}"
    # a world that has been running this long is saturated: most interior cells hold something
    filled = float((eng.grid[:, 1, 1:-1, 1:-1] >= 3).float().mean())
    assert filled > 0.25, filled
    assert eng.status() == 0 and one_call.status() == 0


# ------------------------------------------------------------------ full_view observations in the engine
def test_full_view_kernel_vs_reference_fixture_and_oracle(torch_cuda):
    """sgw_observe_full == what the reference's own OneHot / RGB specs with full_view=True observed (fixture generated by
    running the reference), also as uint8; and through the API: ``OneHotObservationSpec(full_view=True).observe(world)``."""
    torch = torch_cuda
    import copy

    d, spec = H.load_golden("full_view_treasurehunt")
    E, T = d["grid0"].shape[0], d["grid"].shape[0]
    ws = H.world_spec(spec)
    for dtype in (torch.float32, torch.uint8):
        eng = make_engine(ws, E, obs_dtype=dtype)
        for t in range(T):
            eng.grid.copy_(torch.from_numpy(d["grid"][t]))
            got = eng.observe_full().cpu().numpy()
            assert got.shape == d["full_onehot"][t].shape and np.array_equal(got.astype(np.float64), d["full_onehot"][t]), (dtype, t)
    rgb = copy.deepcopy(spec)
    rgb.appearance = d["rgb_table"][[0, 0, 1, 2, 3, 4, 5]].astype(np.float64)
    rgb.num_channels, rgb.obs_post = 3, 1
    eng = make_engine(H.world_spec(rgb), E)
    for t in range(T):
        eng.grid.copy_(torch.from_numpy(d["grid"][t]))
        got = eng.observe_full().cpu().numpy()
        assert np.array_equal(got, d["full_rgb"][t].astype(np.float32)), t
    with pytest.raises(ValueError):
        eng.observe_full(out=torch.zeros((E, 3, 9, 12), device="cuda:0"))
    # a bigger batch against the oracle's restatement (ragged world: 21x21x2 = 882 bytes per env in a padded stride)
    from sorrel_amd.spec import treasurehunt_spec

    ws2 = treasurehunt_spec(21, 21, 3, 2, spawn_prob=0.1, seed=2, dense_prob=0.3)
    eng2, co = make_engine(ws2, 300), H.COracle(ws2, 300)
    eng2.reset(0)
    co.reset(0)
    for t in range(1, 4):
        eng2.step(random_actions=True, turn=t)
        co.step(0, t, random_actions=True)
    got = eng2.observe_full().cpu().numpy()
    osp = H.oracle_spec(ws2)
    for e in (0, 1, 150, 299):
        assert np.array_equal(got[e].astype(np.float64), O_full(osp, co.grid[e])), e


def O_full(spec, grid):
    from oracle import gridstep_oracle as O

    return O.full_view(spec, grid)


# ------------------------------------------------------------------ Tag on its compile-time-shape instances
@pytest.mark.parametrize("jit", [1, 0], ids=["specialised", "prebuilt"])
@pytest.mark.parametrize("shape", [(32, 32, 8, 3, 150, "step_fast<true, 1, 4, 3, 32, 32, true>", None),     # wave per env, static 32x32 map
                                   (32, 32, 8, 3, 65536, "step_fast<true, 1, 4, 3, 32, 32, true>", None),   # ... also for big batches (not packed)
                                   (20, 24, 6, 3, 40000, "step_kernel<32, true, 1, 4, SGW_AGENT_RULE_TAG, 3>", "step_kernel<32, true, 1, 4, 1, 3, 20, 24>"),     # packed, static 7x7 window
                                   (32, 32, 20, 3, 90, "step_fast<true, 1, 4, 3, 32, 32, true>", None),     # crowded: many tags per turn
                                   (30, 30, 6, 4, 70000, "step_fast<true, 0, 0, 0, 0, 0, true, false, true, false, true>", "step_fast<true, 1, 4, 4, 30, 30, true>"),   # wave per env: 3-bit packed counters prebuilt, the whole-env burst specialised
                                   (48, 48, 10, 4, 130, "step_fast<true, 0, 0, 0, 0, 0, true, false, true, false, true>", "step_fast<true, 1, 4, 4, 48, 48, true>"),    # (six workgroups per CU: still the whole-env burst)
                                   (17, 61, 21, 5, 77, "step_fast<true, 0, 0, 0, 0, 0, true, false, true, false, true>", "step_fast<true, 1, 4, 5, 17, 61, true, false, true, false, true>")],    # ragged, crowded, 11x11 windows
                         ids=["static_32x32", "static_32x32_full_batch", "packed_static_radius", "static_32x32_crowded", "p3_30x30_full_batch",
                              "p3_48x48", "p3_ragged_crowded"])
def test_tag_static_instances_vs_oracle(torch_cuda, shape, jit):
    """TagAgent.act on the instances round 3 added (only the agent that is "it" looks at its neighbours; compile-time
    32x32 map on the wave-per-env kernel; compile-time 7x7 window on the packed kernel) and on the instances specialised for the
    engine's own map (round 4): every tensor, the agents' types and what they were when they observed, against the C oracle;
    the 1 + A phased form too."""
    torch = torch_cuda
    h, w, a_, r_, E, prebuilt, special = shape
    N.set_option("jit", jit)
    kernel = (special or prebuilt) if jit else prebuilt
    d, spec = H.load_golden("tag_9x9")
    ws = H.world_spec(spec)
    ws.height, ws.width, ws.num_agents, ws.vision_radius, ws.agent_type = h, w, a_, r_, [ws.agent_type[0]] * a_
    eng = make_engine(ws, E, first=9)
    assert kernel in eng.launch_info() and f"specialised={1 if (jit and special) else 0}" in eng.launch_info(), eng.launch_info()
    Ec = min(E, 400)                                         # the oracle replays the first envs of the batch
    co = H.COracle(ws, Ec, first_env_id=9)
    eng.reset(0)
    co.reset(0)
    assert np.array_equal(eng.agent_state[:Ec].cpu().numpy(), co.agent_state)
    for t in range(1, 13):
        co.step(0, t, random_actions=True)
        if t % 4 == 0 and E <= 400:                          # policy-driven form: sweep-less launches, one per agent, the next agent's window each
            acts = torch.from_numpy(co.actions.copy())
            eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
            for a in range(a_):
                eng.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, obs_next=a + 1 < a_, write_obs=False, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            what = ("grid", "agent_pos", "total_reward", "rewards")
        else:
            eng.step(random_actions=True, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            what = ("grid", "agent_pos", "total_reward", "rewards", "obs", "actions")
            assert np.array_equal(eng.state_at_pov[:Ec].cpu().numpy(), co.state_at_pov), t
        ref = dict(grid=co.grid, agent_pos=co.pos, total_reward=co.total, rewards=co.rewards, obs=co.obs, actions=co.actions)
        for k in what:
            assert np.array_equal(getattr(eng, k)[:Ec].cpu().numpy(), ref[k]), f"turn {t}: {k}"
        assert np.array_equal(eng.agent_state[:Ec].cpu().numpy(), co.agent_state), t
    assert ((eng.agent_state == ws.tag_it_type).sum(dim=1) == 1).all()      # exactly one "it" per env, always
    assert eng.status() == 0


# ------------------------------------------------------------------ Tag worlds above 4 KiB on step_big<..., TAG>
@pytest.mark.parametrize("case", ["tag_70x80", "tag_crowded_66x64", "tag_128x128", "tag_72x72_r3_float", "tag_phased", "tag_u8"])
def test_big_tag_worlds_on_the_workgroup_per_env_kernel_vs_oracle(torch_cuda, case):
    """TagAgent.act on step_big: moves resolved in registers, the "it" token walked by wave 0, tags undone along with moves
    when the windows are rendered.  Sparse and crowded worlds (many tags per turn, flags handed on within a turn), the
    Tag example's tables at 128x128 / 64 agents, a float appearance table, the phased 1 + A form, uint8 windows, and
    sgw_observe; every tensor, the agents' types and their types at observation time against the C oracle."""
    torch = torch_cuda
    import dataclasses

    d, spec = H.load_golden("tag_11x11_default")
    ws = H.world_spec(spec)
    kw = {}
    h, w, a, r, E, T = {"tag_70x80": (70, 80, 12, 4, 13, 25), "tag_crowded_66x64": (66, 64, 64, 4, 7, 14), "tag_128x128": (128, 128, 64, 4, 9, 8),
                        "tag_72x72_r3_float": (72, 72, 30, 3, 6, 10), "tag_phased": (80, 64, 9, 4, 5, 8), "tag_u8": (70, 70, 40, 2, 6, 8)}[case]
    ws = dataclasses.replace(ws, height=h, width=w, num_agents=a, vision_radius=r, agent_type=[ws.agent_type[0]] * a)
    if case == "tag_72x72_r3_float":
        ws.appearance = ws.appearance * 1.0
        ws.appearance[ws.tag_it_type, 0] = 2.5           # not one-hot: the float64 layer-sum path
    if case == "tag_u8":
        kw["obs_dtype"] = torch.uint8
    eng = make_engine(ws, E, first=17, **kw)
    name = eng.launch_info().split(" group")[0]
    assert name.startswith("step_big<") and (name.endswith(", true>") or name.endswith(", true, 256>")), eng.launch_info()   # the TAG instance (256 threads up to 32 agents)
    co = H.COracle(ws, E, first_env_id=17)
    eng.reset(0)
    co.reset(0)
    assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state)
    ref = lambda: dict(grid=co.grid, agent_pos=co.pos, total_reward=co.total, rewards=co.rewards, obs=co.obs, actions=co.actions)
    for t in range(1, T + 1):
        assert co.step(0, t, random_actions=True) == 0
        if case == "tag_phased":
            acts = torch.from_numpy(co.actions.copy()).cuda()
            seen = torch.zeros_like(eng.obs)
            eng.obs.fill_(-3.0)
            eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
            for i in range(a):
                seen[:, i] = eng.obs[:, i]
                eng.step(acts, sweep=False, agent_begin=i, agent_end=i + 1, obs_next=i + 1 < a, write_obs=False, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            assert np.array_equal(seen.cpu().numpy(), co.obs), f"{case} turn {t}: phased windows"
            what = ("grid", "agent_pos", "total_reward")
        else:
            eng.step(random_actions=True, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            what = ("grid", "agent_pos", "total_reward", "rewards", "actions", "obs")
            assert np.array_equal(eng.state_at_pov.cpu().numpy(), co.state_at_pov), f"{case} turn {t}: state_at_pov"
        for k in what:
            assert np.array_equal(getattr(eng, k).cpu().numpy().astype(ref()[k].dtype), ref()[k]), f"{case} turn {t}: {k}"
        assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), f"{case} turn {t}: agent_state"
    eng.obs.zero_()
    eng.observe()
    co.observe()
    torch.cuda.synchronize()
    assert np.array_equal(eng.obs.cpu().numpy().astype(np.float32), co.obs), "sgw_observe"
    assert ((eng.agent_state == ws.tag_it_type).sum(dim=1) == 1).all()
    if case == "tag_crowded_66x64":
        assert len(np.unique(eng.agent_state.cpu().numpy(), axis=0)) > 1        # the flag really moved around
    assert eng.status() == 0


# ------------------------------------------------------------------ sgw_act for Tag and Cleanup agents
@pytest.mark.parametrize("case", ["tag_11x11", "tag_crowded_12x9", "tag_big_70x66", "cleanup_15x16", "cleanup_21x31", "cleanup_big_40x48", "cleanup_u8"])
def test_sgw_act_tag_and_cleanup_vs_oracle(torch_cuda, case):
    """The patched-window protocol for the agents with interaction rules: sweep + every window in one launch
    (SGW_STEP_NO_MOVE), then per agent sgw_act = TagAgent.act / CleanupAgent.act + repair of the later agents' windows (the
    mover's cells, the tagger's and its victim's, the beam cells).  What each agent's policy would read, rewards, grid,
    positions, totals, the agents' types / types at observation time / facings: all against the C oracle."""
    torch = torch_cuda
    import dataclasses

    kw = {}
    grid0 = pos0 = None
    if case.startswith("tag"):
        d, spec = H.load_golden("tag_11x11_default")
        ws = H.world_spec(spec)
        h, w, a, E = {"tag_11x11": (11, 11, 5, 90), "tag_crowded_12x9": (12, 9, 30, 25), "tag_big_70x66": (70, 66, 40, 6)}[case]
        ws = dataclasses.replace(ws, height=h, width=w, num_agents=a, agent_type=[ws.agent_type[0]] * a,
                                 vision_radius=min(ws.vision_radius, (min(h, w) - 1) // 2))
        T = 12
    else:
        name = "cleanup_21x31_default" if case == "cleanup_21x31" else "cleanup_15x16"
        d, spec = H.load_golden(name)
        ws = H.world_spec(spec)
        E, T = 23, 14
        if case == "cleanup_big_40x48":
            h, w, a = 40, 48, 10
            ws = dataclasses.replace(ws, height=h, width=w, num_agents=a, agent_type=[ws.agent_type[0]] * a, beam_radius=7)
            g = np.zeros((3, h, w), np.uint8)
            g[:, 0, :] = g[:, -1, :] = 2
            g[:, :, 0] = g[:, :, -1] = 2
            g[0, 1:12, 1:-1] = 3
            g[0, 12:28, 1:-1] = 1
            g[0, 28:39, 1:-1] = 5
            pos = np.array([[14 + (i // 5) * 6, 4 + (i % 5) * 9] for i in range(a)], np.uint8)
            for (y, x) in pos:
                g[1, y, x] = 11
            grid0, pos0 = g, pos
            E = 7
        else:
            grid0, pos0 = d["grid0"][0], d["pos0"][0]
        if case == "cleanup_u8":
            kw["obs_dtype"] = torch.uint8
    A = ws.num_agents
    eng, co = make_engine(ws, E, first=21, **kw), H.COracle(ws, E, first_env_id=21)
    if grid0 is not None:
        eng.grid.copy_(torch.from_numpy(np.broadcast_to(grid0, (E,) + grid0.shape).copy()))
        eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos0, (E,) + pos0.shape).copy()))
        eng.total_reward.zero_()
        co.grid[...] = grid0
        co.pos[...] = pos0
        co.total[...] = 0
    else:
        eng.reset(0)
        co.reset(0)
    if eng.agent_state is not None:
        assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state)
    rows = eng.window_rows(None)
    for t in range(1, T + 1):
        assert co.step(0, t, random_actions=True) == 0
        eng.actions.copy_(torch.from_numpy(co.actions))
        eng.obs.fill_(77)
        eng.step(eng.actions, sweep=True, no_move=True, turn=t)                 # the sweep and every agent's window, once
        seen = torch.zeros_like(eng.obs)
        rew = torch.zeros_like(eng.rewards)
        pov_types = torch.zeros_like(eng.actions)
        for a in range(A):
            seen[:, a] = eng.obs[:, a]                                           # what agent a's policy reads
            if eng.agent_state is not None:
                pov_types[:, a] = eng.agent_state[:, a]                          # ... and the flag TagAgent.pov appends
            eng.act(a, rows, action=eng.actions[:, a].to(torch.int64).contiguous() if a % 2 else None)
            rew[:, a] = eng.rewards[:, a]
        torch.cuda.synchronize()
        assert np.array_equal(seen.cpu().numpy().astype(np.float32), co.obs), f"{case} turn {t}: windows"
        assert np.array_equal(rew.cpu().numpy(), co.rewards), f"{case} turn {t}: rewards"
        assert np.array_equal(eng.grid.cpu().numpy(), co.grid), f"{case} turn {t}: grid"
        assert np.array_equal(eng.agent_pos.cpu().numpy(), co.pos), f"{case} turn {t}: positions"
        assert np.array_equal(eng.total_reward.cpu().numpy(), co.total), f"{case} turn {t}: total_reward"
        if eng.agent_state is not None:
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), f"{case} turn {t}: agent_state"
            assert np.array_equal(eng.state_at_pov.cpu().numpy(), co.state_at_pov) and np.array_equal(pov_types.cpu().numpy(), co.state_at_pov), t
        if eng.agent_dir is not None:
            assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), f"{case} turn {t}: agent_dir"
    if case.startswith("tag"):
        assert len(np.unique(eng.agent_state.cpu().numpy(), axis=0)) > 1 or E < 10    # the flag really moved around
    if case.startswith("cleanup"):
        assert (eng.grid[:, 2] != eng.grid[0, 2, 1, 1]).any(), "no beam was ever fired"
    assert eng.status() == 0


# ------------------------------------------------------------------ soak: the patched-window protocol on random rule worlds
@pytest.mark.parametrize("case", range(int(os.environ.get("SGW_SOAK", "48"))))
def test_patched_window_protocol_soak_random_rule_worlds(torch_cuda, case):
    """Random layered worlds (2-4 layers, spawners, BECOME_IF tables, timers; plain movers, Tag or Cleanup agents; random
    maps, radii, agent counts, beam radii): every turn is played as sweep + every window once (SGW_STEP_NO_MOVE, or
    sgw_observe_rows where an instance exists) and one sgw_act per agent, the policy's actions handed over as int64 / int32 /
    uint8 tensors; what every agent's policy would read and every piece of state against the C oracle's take_turn."""
    torch = torch_cuda
    from sorrel_amd import _native as N

    rng = np.random.default_rng(9100 + case)
    ws, g, pos = H.random_rule_world(rng)
    E, T = int(rng.integers(2, 40)), int(rng.integers(3, 10))
    first = int(rng.integers(0, 2**31))
    kw = {}
    onehot = bool(((ws.appearance == 0) | (ws.appearance == 1)).all() and (ws.appearance.sum(axis=1) <= 1).all())
    if onehot and case % 5 == 4:
        kw["obs_dtype"] = torch.uint8
    eng = make_engine(ws, E, first=first, **kw)
    co = H.COracle(ws, E, first_env_id=first)
    eng.grid.copy_(torch.from_numpy(np.broadcast_to(g, (E,) + g.shape).copy()))
    eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos, (E,) + pos.shape).copy()))
    eng.total_reward.zero_()
    co.grid[...] = g
    co.pos[...] = pos
    co.total[...] = 0
    epoch = int(rng.integers(0, 9))
    eng.epoch = epoch
    A = ws.num_agents
    caps = eng.capabilities()
    assert caps & N.CAP_ACT
    per_env = int(np.prod(ws.obs_shape[1:]))
    kinds = (torch.int64, torch.int32, torch.uint8)
    for t in range(1, T + 1):
        assert co.step(epoch, t, random_actions=True) == 0
        acts = torch.from_numpy(co.actions.copy()).cuda()
        own_rows = bool(caps & N.CAP_OBSERVE_ROWS) and t % 2 == 0
        dests = [torch.full((E, per_env), -3.0, device="cuda:0") for _ in range(A)] if own_rows else None
        rows = eng.window_rows(dests)
        if own_rows:
            eng.step(acts, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t, advance_turn=False)
            eng.observe_rows(rows)
        else:
            eng.step(acts, sweep=True, no_move=True, turn=t, advance_turn=False)
        eng.actions.fill_(99)                                  # the acts below read the policy's tensors, not this table
        seen = torch.zeros_like(eng.obs)
        rew = torch.zeros_like(eng.rewards)
        for a in range(A):
            seen[:, a] = dests[a].view(E, *ws.obs_shape[1:]) if own_rows else eng.obs[:, a]
            eng.act(a, rows, action=acts[:, a].to(kinds[(a + case) % 3]).contiguous())
            rew[:, a] = eng.rewards[:, a]
        torch.cuda.synchronize()
        ctx = f"case {case} turn {t} (rule {ws.agent_rule}, {ws.layers} layers, {A} agents, r {ws.vision_radius})"
        assert np.array_equal(seen.cpu().numpy().astype(np.float32), co.obs), ctx + ": windows"
        assert np.array_equal(rew.cpu().numpy(), co.rewards), ctx + ": rewards"
        assert np.array_equal(eng.actions.cpu().numpy(), co.actions), ctx + ": recorded actions"
        assert np.array_equal(eng.grid.cpu().numpy(), co.grid), ctx + ": grid"
        assert np.array_equal(eng.agent_pos.cpu().numpy(), co.pos), ctx + ": positions"
        assert np.array_equal(eng.total_reward.cpu().numpy(), co.total), ctx + ": total_reward"
        if eng.agent_dir is not None:
            assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), ctx + ": agent_dir"
        if eng.agent_state is not None:
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), ctx + ": agent_state"
            assert np.array_equal(eng.state_at_pov.cpu().numpy(), co.state_at_pov), ctx + ": state_at_pov"
    assert eng.status() == 0


# ------------------------------------------------------------------ layered rule sets between 4 and 8 KiB per env on the wave-per-env RULES kernel
@pytest.mark.parametrize("case", ["cleanup_40x48", "cleanup_48x48_wide_beam", "cleanup_ragged_45x43", "become_if_movers_50x52", "phased_40x48", "rollout_40x48",
                                  "cleanup_56x64_11k", "become_if_movers_ragged_59x61_11k"])
def test_layered_rule_worlds_up_to_8k_per_env_vs_oracle(torch_cuda, monkeypatch, case):
    """Cleanup / BECOME_IF worlds above 4 KiB (up to 8 KiB) per env stay on step_fast<..., RULES>: the part of the grid
    beyond the first 4 KiB reaches LDS in a second round.  Fused turns, the 1 + A phased form, sgw_rollout, a ragged world
    (byte count not a multiple of 16) and plain movers under BECOME_IF rules, against the C oracle."""
    torch = torch_cuda
    import dataclasses

    d, spec = H.load_golden("cleanup_15x16")
    ws = H.world_spec(spec)
    h, w, a, R = {"cleanup_40x48": (40, 48, 10, 3), "cleanup_48x48_wide_beam": (48, 48, 12, 9), "cleanup_ragged_45x43": (45, 43, 7, 3),
                  "become_if_movers_50x52": (50, 52, 9, 3), "phased_40x48": (40, 48, 6, 3), "rollout_40x48": (40, 48, 8, 3),
                  "cleanup_56x64_11k": (56, 64, 10, 4), "become_if_movers_ragged_59x61_11k": (59, 61, 9, 3)}[case]
    if case.endswith("_11k"):      # up to 11 KiB per env from 16 384 envs on; the hook takes the path for a batch a test can check
        N.set_option("rules_11k", 1)
    ws = dataclasses.replace(ws, height=h, width=w, num_agents=a, agent_type=[ws.agent_type[0]] * a, beam_radius=R)
    if case.startswith("become_if_movers"):      # the same layered rule tables with MovingAgent.act agents
        ws = dataclasses.replace(ws, agent_rule=0, action_kind=[0] * len(ws.action_dy))
    g = np.zeros((3, h, w), np.uint8)
    g[:, 0, :] = g[:, -1, :] = 2
    g[:, :, 0] = g[:, :, -1] = 2
    g[0, 1:h // 3, 1:-1] = 3
    g[0, h // 3:2 * h // 3, 1:-1] = 1
    g[0, 2 * h // 3:h - 1, 1:-1] = 5
    pos = np.array([[h // 3 + 1 + (i // 5) * 3, 4 + (i % 5) * 7] for i in range(a)], np.uint8)
    for (y, x) in pos:
        g[1, y, x] = 11
    E, T = 9, 10
    eng, co = make_engine(ws, E, first=5), H.COracle(ws, E, first_env_id=5)
    info = eng.launch_info()
    assert "step_fast<" in info and "group=64 " in info, info
    eng.grid.copy_(torch.from_numpy(np.broadcast_to(g, (E,) + g.shape).copy()))
    eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos, (E,) + pos.shape).copy()))
    eng.total_reward.zero_()
    co.grid[...] = g
    co.pos[...] = pos
    co.total[...] = 0

    def same(ctx, what=("grid", "agent_pos", "total_reward", "rewards", "actions", "obs")):
        torch.cuda.synchronize()
        ref = dict(grid=co.grid, agent_pos=co.pos, total_reward=co.total, rewards=co.rewards, obs=co.obs, actions=co.actions)
        for k in what:
            assert np.array_equal(getattr(eng, k).cpu().numpy(), ref[k]), f"{case} {ctx}: {k}"
        if eng.agent_dir is not None:
            assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), f"{case} {ctx}: agent_dir"

    if case == "rollout_40x48":
        for t in range(1, T + 1):
            assert co.step(0, t, random_actions=True) == 0
        eng.rollout(T)
        same("after sgw_rollout")
    else:
        for t in range(1, T + 1):
            assert co.step(0, t, random_actions=True) == 0
            if case == "phased_40x48":
                acts = torch.from_numpy(co.actions.copy()).cuda()
                seen = torch.zeros_like(eng.obs)
                eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
                for i in range(a):
                    seen[:, i] = eng.obs[:, i]
                    eng.step(acts, sweep=False, agent_begin=i, agent_end=i + 1, obs_next=i + 1 < a, write_obs=False, turn=t, advance_turn=False)
                torch.cuda.synchronize()
                assert np.array_equal(seen.cpu().numpy(), co.obs), f"{case} turn {t}: phased windows"
                same(f"turn {t}", ("grid", "agent_pos", "total_reward"))
            else:
                eng.step(random_actions=True, turn=t, advance_turn=False)
                same(f"turn {t}")
    assert (eng.grid[:, 2] != eng.grid[0, 2, 1, 1]).any() or case == "become_if_movers_50x52", "no beam was ever fired"
    assert eng.status() == 0


# ------------------------------------------------------------------ step_big: windows staged in LDS, line-aligned streaming stores
@pytest.mark.parametrize("case", ["config5_shape", "crowded_48x48", "odd_70x90", "u8", "phased", "rollout", "misaligned_obs"])
def test_big_kernel_staged_windows_vs_oracle(torch_cuda, monkeypatch, case):
    """Worlds above 4 KiB whose tables have a compile-time step_big instance stage each window's byte counts in the rendering
    wave's LDS area and write them as 16-byte streaming stores on 128-byte lines (batches of more than ~1.75 rounds of
    workgroups; SGW_BIG_STAGE=1 forces it for the small batches a test can check element by element).  Windows whose first
    element sits anywhere in a line (A * C * V * V odd multiples), the uint8 format, the phased 1 + A form (one window per
    launch), sgw_rollout, and an observation tensor that is not 16-byte aligned (falls back to the direct stores); every
    tensor against the C oracle."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("big_stage", 1)
    kw = {}
    h, w, a, E, T = {"config5_shape": (128, 128, 64, 5, 4), "crowded_48x48": (48, 48, 64, 21, 8), "odd_70x90": (70, 90, 13, 11, 6),
                     "u8": (64, 80, 17, 6, 5), "phased": (80, 64, 7, 5, 5), "rollout": (72, 72, 9, 6, 6),
                     "misaligned_obs": (64, 80, 5, 7, 4)}[case]
    ws = treasurehunt_spec(h, w, a, 5, spawn_prob=0.05, seed=31, dense_prob=0.3 if case == "crowded_48x48" else 0.1)
    if case == "u8":
        kw["obs_dtype"] = torch.uint8
    eng = make_engine(ws, E, first=5, **kw)
    info = eng.launch_info()
    assert "step_big<" in info and "big_stage=0" not in info, info
    co = H.COracle(ws, E, first_env_id=5)
    eng.reset(0)
    co.reset(0)
    if case == "misaligned_obs":      # a view that starts 4 bytes into the allocation: the kernel must not stage (and must still be right)
        flat = torch.full((eng.obs.numel() + 4,), -7.0, device="cuda")
        eng.obs = flat[1:1 + eng.obs.numel()].view(eng.obs.shape)
        assert eng.obs.data_ptr() % 16 == 4
    ref = lambda: dict(grid=co.grid, agent_pos=co.pos, total_reward=co.total, rewards=co.rewards, obs=co.obs, actions=co.actions)
    if case == "rollout":
        obs_t = torch.empty((T,) + tuple(eng.obs.shape), device="cuda")
        eng.rollout(T, obs_out=obs_t)
        torch.cuda.synchronize()
        for t in range(1, T + 1):
            assert co.step(0, t, random_actions=True) == 0
            assert np.array_equal(obs_t[t - 1].cpu().numpy(), co.obs), f"rollout turn {t}: obs"
        for k in ("grid", "agent_pos", "total_reward"):
            assert np.array_equal(getattr(eng, k).cpu().numpy(), ref()[k]), f"rollout: {k}"
        return
    for t in range(1, T + 1):
        assert co.step(0, t, random_actions=True) == 0
        if case == "phased":
            acts = torch.from_numpy(co.actions.copy()).cuda()
            a = ws.num_agents
            seen = torch.zeros_like(eng.obs)
            eng.obs.fill_(-3.0)
            eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
            for i in range(a):
                seen[:, i] = eng.obs[:, i]
                eng.step(acts, sweep=False, agent_begin=i, agent_end=i + 1, obs_next=i + 1 < a, write_obs=False, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            assert np.array_equal(seen.cpu().numpy(), co.obs), f"phased turn {t}: windows"
            what = ("grid", "agent_pos", "total_reward")
        else:
            eng.step(random_actions=True, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            what = ("grid", "agent_pos", "total_reward", "rewards", "actions", "obs")
        for k in what:
            assert np.array_equal(getattr(eng, k).cpu().numpy().astype(ref()[k].dtype), ref()[k]), f"{case} turn {t}: {k}"
    eng.obs.zero_()
    eng.observe()
    co.observe()
    torch.cuda.synchronize()
    assert np.array_equal(eng.obs.cpu().numpy().astype(co.obs.dtype), co.obs), f"{case}: sgw_observe"
    assert eng.status() == 0


# ------------------------------------------------------------------ compile-time window on a run-time map
@pytest.mark.parametrize("shape", [(33, 32, 8, 3), (24, 24, 8, 2), (20, 20, 4, 4), (30, 26, 7, 5), (19, 23, 5, 3)])
def test_static_radius_instances_on_runtime_maps_vs_oracle(torch_cuda, monkeypatch, shape):
    """Treasurehunt-shaped worlds whose MAP the library holds no instance for run on the instance specialised for that map
    (round 3: a compile-time window on a run-time map; round 4: the map too -- whole-env burst where the windows are a multiple
    of four elements, chunked bursts elsewhere; ragged maps included: 19x23x2 = 874 cells).  Every tensor against the C
    oracle, turn by turn, then sgw_observe; group = 64 keeps small worlds off the packed kernel."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("group", 64)
    h, w, a, r = shape
    ws = treasurehunt_spec(h, w, a, r, spawn_prob=0.06, seed=41, dense_prob=0.15)
    E, T = 37, 9
    eng, co = make_engine(ws, E, first=7), H.COracle(ws, E, first_env_id=7)
    assert f"step_fast<true, 2, 6, {r}, {h}, {w}" in eng.launch_info() and "specialised=1" in eng.launch_info(), eng.launch_info()
    eng.reset(0)
    co.reset(0)
    for t in range(1, T + 1):
        eng.step(random_actions=True)
        assert co.step(0, t, random_actions=True) == 0
        torch.cuda.synchronize()
        for k, ref in (("grid", co.grid), ("agent_pos", co.pos), ("total_reward", co.total), ("rewards", co.rewards),
                       ("actions", co.actions), ("obs", co.obs)):
            assert np.array_equal(getattr(eng, k).cpu().numpy(), ref), f"{shape} turn {t}: {k}"
    eng.obs.zero_()
    eng.observe()
    co.observe()
    torch.cuda.synchronize()
    assert np.array_equal(eng.obs.cpu().numpy(), co.obs)
    assert eng.status() == 0


# ------------------------------------------------------------------ line-aligned bursts: chunks smaller than a line, every alignment
@pytest.mark.parametrize("case", [(24, 20, 1, 2, 7, 1, 1), (24, 20, 1, 2, 7, 1, 2), (24, 20, 2, 3, 9, 1, 1), (18, 22, 2, 6, 5, 1, 1),
                                  (16, 16, 1, 1, 6, 2, 1), (20, 24, 2, 5, 11, 2, 3), (32, 33, 2, 6, 8, 3, 3), (21, 31, 3, 9, 10, 5, 3)],
                         ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}_C{c[3]}_A{c[4]}_r{c[5]}_burst{c[6]}")
def test_staged_bursts_carry_partial_lines_vs_oracle(torch_cuda, monkeypatch, case):
    """emit_chunk: a chunk leaves up to its last 128-byte line boundary and the bytes behind it are carried into the next
    chunk.  Forced small bursts (SGW_STAGE_AGENTS) over windows of 18 / 27 / 25 / 54 bytes -- chunks that end inside the
    env's first line and write nothing yet, chunks smaller than a line, 99 envs so that an env's block starts at every
    multiple of 8 / 4 bytes -- in both observation formats, then the ordinary burst sizes; every element against the C oracle."""
    torch = torch_cuda
    h, w, L, C, a, r, burst = case
    N.set_option("group", 64)
    N.set_option("stage_agents", burst)
    ws = _move_world(h, w, L, C, a, r, seed=5)
    for dtype in (torch.float32, torch.uint8):
        E, T = 99, 4
        eng, co = make_engine(ws, E, first=2, obs_dtype=dtype), H.COracle(ws, E, first_env_id=2)
        assert f"stage_agents={burst}" in eng.launch_info(), eng.launch_info()
        eng.reset(0)
        co.reset(0)
        for t in range(1, T + 1):
            eng.obs.fill_(77)
            eng.step(random_actions=True)
            assert co.step(0, t, random_actions=True) == 0
            torch.cuda.synchronize()
            assert np.array_equal(eng.obs.cpu().numpy().astype(co.obs.dtype), co.obs), f"{case} {dtype} turn {t}: obs"
            assert np.array_equal(eng.grid.cpu().numpy(), co.grid) and np.array_equal(eng.total_reward.cpu().numpy(), co.total)
        assert eng.status() == 0


# ------------------------------------------------------------------ plain / Tag worlds of 4-8 KiB on the wave-per-env kernel
@pytest.mark.parametrize("case", ["th_48x48_r5", "th_64x64_A16", "th_ragged_51x47", "th_u8", "th_phased", "th_rollout", "th_policy_protocol",
                                  "tag_72x72", "tag_ragged_89x91_crowded", "generic_tables_60x60", "th_74x75_11k", "tag_105x106_11k"])
def test_mid_size_worlds_on_the_wave_per_env_kernel_vs_oracle(torch_cuda, monkeypatch, case):
    """Worlds between 4 and 8 KiB per env (11 KiB from 16 384 envs on) run a wave per env when the batch is large (>= 4 096 envs; SGW_FAST_8K=1 forces it for the
    batches a test can check element by element): the part of the grid beyond the first 4 KiB reaches LDS in a second round.
    Treasurehunt tables with compile-time windows, ragged maps, uint8 windows, the phased 1 + A form, sgw_rollout, the
    patched-window policy protocol (NO_MOVE + sgw_act), Tag (3-bit counters), another entity set; every tensor vs the C oracle."""
    torch = torch_cuda
    import dataclasses
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("fast_8k", 1)
    kw = {}
    tag = case.startswith("tag")
    if tag:
        d, spec = H.load_golden("tag_11x11_default")
        ws = H.world_spec(spec)
        h, w, a, r = {"tag_72x72": (72, 72, 16, 4), "tag_ragged_89x91_crowded": (89, 91, 40, 3), "tag_105x106_11k": (105, 106, 20, 4)}[case]
        ws = dataclasses.replace(ws, height=h, width=w, num_agents=a, vision_radius=r, agent_type=[ws.agent_type[0]] * a)
    elif case == "generic_tables_60x60":
        ws = _move_world(60, 60, 2, 8, 9, 3, seed=8)
    else:
        h, w, a, r = {"th_48x48_r5": (48, 48, 8, 5), "th_64x64_A16": (64, 64, 16, 3), "th_ragged_51x47": (51, 47, 7, 4), "th_u8": (50, 50, 6, 2),
                      "th_phased": (48, 50, 5, 3), "th_rollout": (56, 56, 6, 3), "th_policy_protocol": (48, 48, 6, 3), "th_74x75_11k": (74, 75, 9, 5)}[case]
        ws = treasurehunt_spec(h, w, a, r, spawn_prob=0.04, seed=51, dense_prob=0.2)
        if case == "th_u8":
            kw["obs_dtype"] = torch.uint8
    E, T = 21, 6
    eng, co = make_engine(ws, E, first=4, **kw), H.COracle(ws, E, first_env_id=4)
    assert "step_fast<" in eng.launch_info() and 4096 < ws.layers * ws.height * ws.width <= 11264, eng.launch_info()
    eng.reset(0)
    co.reset(0)
    ref = lambda: dict(grid=co.grid, agent_pos=co.pos, total_reward=co.total, rewards=co.rewards, obs=co.obs, actions=co.actions)
    if case == "th_rollout":
        obs_t = torch.empty((T,) + tuple(eng.obs.shape), device="cuda")
        eng.rollout(T, obs_out=obs_t)
        torch.cuda.synchronize()
        for t in range(1, T + 1):
            assert co.step(0, t, random_actions=True) == 0
            assert np.array_equal(obs_t[t - 1].cpu().numpy(), co.obs), f"rollout turn {t}: obs"
        for k in ("grid", "agent_pos", "total_reward"):
            assert np.array_equal(getattr(eng, k).cpu().numpy(), ref()[k]), f"rollout: {k}"
        return
    A = ws.num_agents
    for t in range(1, T + 1):
        assert co.step(0, t, random_actions=True) == 0
        acts = torch.from_numpy(co.actions.copy()).cuda()
        if case == "th_phased":
            seen = torch.zeros_like(eng.obs)
            eng.obs.fill_(-3.0)
            eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
            for i in range(A):
                seen[:, i] = eng.obs[:, i]
                eng.step(acts, sweep=False, agent_begin=i, agent_end=i + 1, obs_next=i + 1 < A, write_obs=False, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            assert np.array_equal(seen.cpu().numpy(), co.obs), f"phased turn {t}: windows"
            what = ("grid", "agent_pos", "total_reward")
        elif case == "th_policy_protocol":
            eng.actions.copy_(acts)
            eng.step(eng.actions, sweep=True, no_move=True, turn=t, advance_turn=False)      # sweep + every window, nobody moves
            rows = eng.window_rows(None)
            seen = torch.zeros_like(eng.obs)
            for i in range(A):
                seen[:, i] = eng.obs[:, i]              # what agent i's policy reads: the grid after the acts of agents < i
                eng.act(i, rows)
            torch.cuda.synchronize()
            assert np.array_equal(seen.cpu().numpy(), co.obs), f"policy protocol turn {t}: windows"
            what = ("grid", "agent_pos", "total_reward", "rewards")
        else:
            eng.step(random_actions=True, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            what = ("grid", "agent_pos", "total_reward", "rewards", "actions", "obs")
        for k in what:
            assert np.array_equal(getattr(eng, k).cpu().numpy().astype(ref()[k].dtype), ref()[k]), f"{case} turn {t}: {k}"
        if tag:
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), f"{case} turn {t}: agent_state"
    eng.obs.zero_()
    eng.observe()
    co.observe()
    torch.cuda.synchronize()
    assert np.array_equal(eng.obs.cpu().numpy().astype(co.obs.dtype), co.obs), f"{case}: sgw_observe"
    assert eng.status() == 0


# ------------------------------------------------------------------ step_big with four or eight waves per workgroup
@pytest.mark.parametrize("threads", ["256", "512"])
@pytest.mark.parametrize("case", ["plain_90x90_A16", "config5_shape_A64", "crowded_48x48_A64", "staged_100x100_A8_r5", "walking_96x96_A12",
                                  "tag_128x128_A32", "tag_crowded_80x80_A64", "generic_tables_96x96"])
def test_big_kernel_four_or_eight_waves_vs_oracle(torch_cuda, monkeypatch, case, threads):
    """step_big<..., BT>: 256 threads per workgroup where agents x window cells <= 2 048, else 512 (SGW_BIG_THREADS_RT forces either).
    Both on few and on many agents (more agents than waves x 8, crowded maps), with the staged windows, on the walking
    variant, for Tag and for another entity set; every tensor against the C oracle, then sgw_observe."""
    torch = torch_cuda
    import dataclasses
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("big_threads", int(threads))
    tag = case.startswith("tag")
    if tag:
        d, spec = H.load_golden("tag_11x11_default")
        ws = H.world_spec(spec)
        h, w, a, r = {"tag_128x128_A32": (128, 128, 32, 3), "tag_crowded_80x80_A64": (80, 80, 64, 4)}[case]
        ws = dataclasses.replace(ws, height=h, width=w, num_agents=a, vision_radius=r, agent_type=[ws.agent_type[0]] * a)
    elif case == "generic_tables_96x96":
        ws = _move_world(96, 96, 2, 8, 11, 3, seed=9)
    else:
        h, w, a, r = {"plain_90x90_A16": (90, 90, 16, 3), "config5_shape_A64": (128, 128, 64, 5), "crowded_48x48_A64": (48, 48, 64, 5),
                      "staged_100x100_A8_r5": (100, 100, 8, 5), "walking_96x96_A12": (96, 96, 12, 4)}[case]
        ws = treasurehunt_spec(h, w, a, r, spawn_prob=0.05, seed=61, dense_prob=0.3 if "crowded" in case else 0.1)
    if case.startswith("staged"):
        N.set_option("big_stage", 1)
    if case.startswith("walking"):
        N.set_option("big_walk_blocks", 3)
    N.set_option("fast_8k", 0)      # (the 4-8 KiB cases stay on this kernel)
    E, T = 10, 6
    eng, co = make_engine(ws, E, first=6), H.COracle(ws, E, first_env_id=6)
    info = eng.launch_info()
    assert "step_big<" in info and f"threads={threads}" in info, info
    eng.reset(0)
    co.reset(0)
    for t in range(1, T + 1):
        eng.step(random_actions=True, turn=t, advance_turn=False)
        assert co.step(0, t, random_actions=True) == 0
        torch.cuda.synchronize()
        for k, ref in (("grid", co.grid), ("agent_pos", co.pos), ("total_reward", co.total), ("rewards", co.rewards), ("actions", co.actions), ("obs", co.obs)):
            assert np.array_equal(getattr(eng, k).cpu().numpy(), ref), f"{case} / {threads} threads, turn {t}: {k}"
        if tag:
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), f"{case} turn {t}: agent_state"
    eng.obs.zero_()
    eng.observe()
    co.observe()
    torch.cuda.synchronize()
    assert np.array_equal(eng.obs.cpu().numpy(), co.obs)
    assert eng.status() == 0


# ------------------------------------------------------------------ RGB observation specs (integer colour tables, clip / 255) on the byte-staging pipeline
@pytest.mark.parametrize("case", ["fixture_turns", "th_32x32", "th_ragged_23x29_r4", "three_layers", "tag_rgb_20x24", "bright_colours_clip",
                                  "phased_and_ranges", "float_colours_fall_back", "th_60x60_8k"])
def test_rgb_integer_tables_on_the_byte_staging_pipeline_vs_oracle(torch_cuda, monkeypatch, case):
    """RGBObservationSpec as the reference builds it (uint8 colours summed over the layers, np.clip(.., 0, 255) / 255) runs on the
    wave-per-env kernel's I16 instances: 16-bit counters, the clipped sum staged as a byte, the burst reading (float)(k / 255.0)
    from a 256-entry table.  The reference-generated fixture turn by turn, batches at other shapes, three layers whose colours add
    up past 255 (the clip), Tag, calls the staged kernel cannot serve (agent ranges / OBS_NEXT: the float64 kernel with its own LDS
    layout), and non-integer colours (must stay on the float64 path); every float bit for bit against the C oracle."""
    torch = torch_cuda
    import dataclasses

    N.set_option("group", 64)
    d, spec = H.load_golden("rgb_treasurehunt")
    ws = H.world_spec(spec)
    E, T = 45, 6
    expect_i16 = True
    if case == "fixture_turns":
        E = 3
    elif case == "th_32x32":
        ws = dataclasses.replace(ws, height=32, width=32, num_agents=8, vision_radius=3, agent_type=[ws.agent_type[0]] * 8)
    elif case == "th_ragged_23x29_r4":
        ws = dataclasses.replace(ws, height=23, width=29, num_agents=5, vision_radius=4, agent_type=[ws.agent_type[0]] * 5)
    elif case == "th_60x60_8k":          # 7 200 bytes per env: the second round of the grid copy (large batches; forced here)
        N.set_option("fast_8k", 1)
        ws = dataclasses.replace(ws, height=60, width=60, num_agents=9, vision_radius=3, agent_type=[ws.agent_type[0]] * 9)
        E = 19
    elif case == "three_layers":
        ws = dataclasses.replace(ws, height=20, width=22, layers=3, agent_layer=2, num_agents=6, vision_radius=3, agent_type=[ws.agent_type[0]] * 6,
                                 layer_fill_type=[ws.layer_fill_type[0], ws.layer_fill_type[0], ws.layer_fill_type[1]],
                                 layer_border_type=[ws.layer_border_type[0], ws.layer_border_type[0], ws.layer_border_type[1]])
    elif case == "tag_rgb_20x24":
        d2, spec2 = H.load_golden("tag_11x11_default")
        wt = H.world_spec(spec2)
        app = np.zeros((len(wt.appearance), 3))
        for t in range(len(app)):
            app[t] = [(37 * t) % 256, (91 * t + 5) % 256, (160 * t) % 256]
        ws = dataclasses.replace(wt, height=20, width=24, num_agents=7, vision_radius=3, agent_type=[wt.agent_type[0]] * 7, num_channels=3,
                                 appearance=app, obs_post=1)
    elif case == "bright_colours_clip":       # two layers of bright entities: sums above 255 in every channel
        app = np.asarray(ws.appearance, dtype=np.float64).copy()
        app[app > 0] = 255.0
        app[0] = [200.0, 180.0, 90.0]        # the empty kind glows too: every cell's two layers add up
        ws = dataclasses.replace(ws, height=18, width=18, num_agents=4, vision_radius=2, agent_type=[ws.agent_type[0]] * 4, appearance=app)
    elif case == "phased_and_ranges":
        ws = dataclasses.replace(ws, height=24, width=24, num_agents=5, vision_radius=3, agent_type=[ws.agent_type[0]] * 5)
    elif case == "float_colours_fall_back":
        app = np.asarray(ws.appearance, dtype=np.float64).copy()
        app[2, 0] = 254.5
        ws = dataclasses.replace(ws, height=24, width=24, num_agents=5, vision_radius=3, agent_type=[ws.agent_type[0]] * 5, appearance=app)
        expect_i16 = False
    eng, co = make_engine(ws, E, first=2), H.COracle(ws, E, first_env_id=2)
    name = eng.launch_info().split(" group")[0]
    assert name.startswith("step_fast<"), eng.launch_info()
    assert name.endswith("false, false, true>") == expect_i16, eng.launch_info()       # ..., MULTI = false, P3 = false, I16 = true
    eng.reset(0)
    co.reset(0)
    tag = case.startswith("tag")
    A = ws.num_agents
    for t in range(1, T + 1):
        assert co.step(0, t, random_actions=True) == 0
        if case == "phased_and_ranges" and t % 2 == 0:
            acts = torch.from_numpy(co.actions.copy()).cuda()
            seen = torch.zeros_like(eng.obs)
            eng.obs.fill_(-3.0)
            eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
            for i in range(A):
                seen[:, i] = eng.obs[:, i]
                eng.step(acts, sweep=False, agent_begin=i, agent_end=i + 1, obs_next=i + 1 < A, write_obs=False, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            assert np.array_equal(seen.cpu().numpy(), co.obs), f"phased turn {t}: windows"
            what = ("grid", "agent_pos", "total_reward")
        else:
            eng.step(random_actions=True, turn=t, advance_turn=False)
            torch.cuda.synchronize()
            what = ("grid", "agent_pos", "total_reward", "rewards", "actions", "obs")
        for k, ref in dict(grid=co.grid, agent_pos=co.pos, total_reward=co.total, rewards=co.rewards, obs=co.obs, actions=co.actions).items():
            if k in what:
                assert np.array_equal(getattr(eng, k).cpu().numpy(), ref), f"{case} turn {t}: {k}"
        if tag:
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), f"{case} turn {t}: agent_state"
    if case == "phased_and_ranges":      # a range of agents through sgw_observe: the float64 kernel again
        eng.obs.fill_(7.0)
        eng.observe(agent_begin=1, agent_end=3)
        co.observe()
        torch.cuda.synchronize()
        assert np.array_equal(eng.obs[:, 1:3].cpu().numpy(), co.obs[:, 1:3]) and float(eng.obs[:, 0].min()) == 7.0
    eng.obs.zero_()
    eng.observe()
    co.observe()
    torch.cuda.synchronize()
    assert np.array_equal(eng.obs.cpu().numpy(), co.obs), f"{case}: sgw_observe"
    if case == "bright_colours_clip":
        assert float(eng.obs.max()) == 1.0
    assert eng.status() == 0


# ------------------------------------------------------------------ bench.py: the timed launches as one hipGraph
@pytest.mark.gpu
def test_bench_timed_region_as_a_graph_plays_the_same_rollout(built):
    """bench.py --graph captures its K timed sgw_step launches in one hipGraph (each node with the turn number it carries when the
    region runs) and replays it inside the region.  The same launches with the same turn numbers: the rollout it reports -- float64
    totals summed over the batch -- must be exactly what the plain loop reports, with and without pre-warm / re-warm launches in
    front.  This test makes no GPU call itself."""
    from tests.test_gpu_round2 import _run_bench

    for extra in (["--prewarm-steps", "0"], ["--prewarm-steps", "40", "--rewarm-steps", "7"]):
        common = ["--gpus", "1", "--envs", "2048", "--steps", "9", "--warmup", "3", "--no-cpu-baseline", "--no-side-configs", "--no-self-check"] + extra
        a = _run_bench(common + ["--graph"], 1, {})
        b = _run_bench(common, 1, {})
        assert "hipGraph" in a["timed_region_submission"] and b["timed_region_submission"] == "K sgw_step calls"
        assert a["rollout"]["sum_total_reward"] == b["rollout"]["sum_total_reward"] and a["rollout"]["status"] == b["rollout"]["status"] == 0
        assert a["steps"] == b["steps"] == 9 and a["roofline"]["kernel_ms"] > 0

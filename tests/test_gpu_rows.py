"""``sgw_sweep_observe_rows`` -- the entity sweep and EVERY agent's window into its own row, one launch -- on the kernels that got it in round 6:
the workgroup-per-env kernel (``step_big``: worlds above 4 KiB, plain and Tag movers, any appearance table) and the chunk-staging wave-per-env
instances (``step_fast_rowsx``: layered rule sets -- Cleanup --, Tag, run-time maps), row tails included.  Checked against the two-launch form
(sweep alone + ``sgw_observe_rows``) and the C oracle.  Reference: ``Agent.transition`` / ``pov``, ``sorrel/agents/agent.py:155-173``; the tails:
``sorrel/examples/tag/agents.py:57-65``, ``sorrel/examples/cleanup/agents.py:52-60``.  (The cases of the compile-time-shape instance of round 5,
the emit modes of ``sgw_observe_rows``, agent-major windows and the row tails of rounds 3-5 follow at the end of this file.)"""
import ctypes

import numpy as np
import pytest

from sorrel_amd import _native as N
from tests import helpers as H

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


def _th(h, w, a, r, **kw):
    from sorrel_amd.spec import treasurehunt_spec

    return treasurehunt_spec(h, w, a, r, spawn_prob=0.04, seed=21, dense_prob=0.15, **kw)


def _float_table(ws):
    """The same world behind a general (not one-hot) appearance table: the float64 path of the kernels."""
    app = np.asarray(ws.appearance, dtype=np.float64).copy()
    app[2] *= 0.5
    app[3, 1] = 0.25
    ws.appearance = app
    return ws


def _tag(h, w, a, r):
    from tests.gpu_common import _tag_spec

    return _tag_spec(h, w, a, r)


def _cleanup():
    from tests.gpu_common import _cleanup_spec

    return _cleanup_spec()[0]


def _big_cleanup():
    ws = _cleanup()
    ws.height, ws.width = 40, 44
    return ws


def _rule_world(seed, mode):
    rng = np.random.default_rng(seed)
    while True:
        ws, g, pos = H.random_rule_world(rng)
        kind = "cleanup" if ws.agent_rule == 2 else ("tag" if ws.agent_rule == 1 else "move")
        if kind == mode and ws.layers * ws.height * ws.width >= 8:
            ws.appearance = (np.asarray(ws.appearance) != 0).astype(np.float64)      # one-hot tables: the row kernels' domain
            return ws, g, pos


# (name, spec factory, envs, options, substring of the instance that must run the fused launch, tail: None / "it" / table length)
CASES = [
    ("c5_shape_direct", lambda: _th(128, 128, 64, 5), 6, {"big_stage": 0, "big_walk": 0}, "step_big<true, 2, 6, 5", None),
    ("c5_shape_staged", lambda: _th(128, 128, 64, 5), 6, {"big_stage": 1, "big_walk": 0}, "step_big<true, 2, 6, 5", None),
    ("c5_shape_walk_window", lambda: _th(128, 128, 64, 5), 23, {"big_walk_blocks": 5}, "step_big<true, 2, 6, 5", None),      # (a batch the step itself would walk: the rows launch does not)
    ("c5_shape_prebuilt_has_none", lambda: _th(128, 128, 64, 5), 5, {"jit": 0}, None, None),      # (the ROWS instances are specialised only)
    ("big_72x80_r3_tail7", lambda: _th(72, 80, 6, 3), 9, {"big_stage": 1}, "step_big<", 7),
    ("big_70x66_r4_256_threads", lambda: _th(70, 66, 10, 4), 11, {}, "step_big<", None),
    ("big_float_table", lambda: _float_table(_th(72, 80, 6, 3)), 7, {}, "step_big<false", None),
    ("big_tag_70x80_it", lambda: _tag(70, 80, 9, 4), 8, {}, "step_big<", "it"),
    ("cleanup_15x16_rules_table12", _cleanup, 19, {}, "step_fast_rowsx<", 12),
    ("cleanup_15x16_rules_no_tail", _cleanup, 19, {}, "step_fast_rowsx<", None),
    ("tag_11x11_it_whole_env", lambda: _tag(11, 11, 5, 4), 37, {"group": 64}, "step_fast_rows<1, 4, 4, 11, 11, true, true>", "it"),
    ("tag_70x60_it_chunked", lambda: _tag(70, 60, 7, 4), 14, {"fast_8k": 1}, "step_fast_rowsx<", "it"),
    ("tag_13x12_prebuilt_has_none", lambda: _tag(13, 12, 4, 2), 9, {"group": 64, "jit": 0}, None, "it"),
    ("runtime_map_33x35_r3", lambda: _th(33, 35, 7, 3), 21, {"burst": 2}, "step_fast_rowsx<", None),
    ("packed_tag_11x11_it", lambda: _tag(11, 11, 5, 4), 70, {"group": 32}, "step_kernel<32, true, 1, 4, 1, 4, 11, 11, false, 64, true>", "it"),      # two envs to a wave
    ("packed_th_10x10_tail4", lambda: _th(10, 10, 2, 2), 133, {"group": 16}, "step_kernel<16, ", 4),                                              # four envs to a wave
    ("generic_wave_per_env_18x14", lambda: _th(18, 14, 4, 3), 21, {"force_generic": 1}, "step_kernel<64, ", None),
    ("ticket_kernel_80_agents_tail5", lambda: _th(40, 42, 80, 2), 9, {}, "step_kernel<256, ", 5),                                              # > 64 agents: 128-entry arrays
    ("ticket_kernel_big_cleanup", lambda: _big_cleanup(), 7, {"fast_rules": 0}, "step_kernel<256, ", 12),
    ("chunked_20x20_r1_odd", lambda: _th(20, 20, 3, 1), 26, {}, "step_fast_rowsx<", 3),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sweep_and_rows_in_one_launch_on_the_big_and_the_chunk_staging_kernels(torch_cuda, case):
    torch = torch_cuda
    name, mk, E, opts, inst, tail = case
    ws = mk()
    A = ws.num_agents
    for k, v in opts.items():
        N.set_option(k, v)
    a, b = make_engine(ws, E, first=3), make_engine(ws, E, first=3)
    N.reset_options()
    if inst is None:           # (the prebuilt run-time-shape instances have no fused twin: the capability bit says so, the call is refused)
        assert not (a.capabilities() & N.CAP_SWEEP_ROWS), a.launch_info()
        rows = a.window_rows([torch.zeros((E, int(np.prod(ws.obs_shape[1:]))), device="cuda:0") for _ in range(A)])
        with pytest.raises(ValueError):
            a.sweep_observe_rows(rows)
        return
    if not (a.capabilities() & N.CAP_SWEEP_ROWS):
        pytest.fail(f"{name}: no fused instance: {a.launch_info()}")
    co = H.COracle(ws, E, first_env_id=3)
    for e in (a, b):
        e.reset(0)
    co.reset(0)
    if a.agent_state is not None:
        co.agent_state[...] = a.agent_state.cpu().numpy()
    table = None
    if tail == "it":
        for e in (a, b):
            e.bind_row_tail(N.TAIL_AGENT_IS_IT)
    elif tail:
        g = torch.Generator().manual_seed(3)
        table = torch.randn((ws.height, ws.width, tail), generator=g).cuda()
        for e in (a, b):
            e.bind_row_tail(N.TAIL_POSITION_TABLE, table)
    assert a.capabilities() & N.CAP_SWEEP_ROWS, (name, "with the tail bound")
    Nw = int(np.prod(ws.obs_shape[1:]))
    Nr = Nw + a.row_tail
    guard = 5
    bufs_a = [torch.full((guard + E * Nr + guard,), -9.0, device="cuda:0") for _ in range(A)]      # (4-byte aligned starts, nothing more)
    dest_a = [buf[guard:guard + E * Nr].view(E, Nr) for buf in bufs_a]
    dest_b = [torch.full((E, Nr), -9.0, device="cuda:0") for _ in range(A)]
    arr = (ctypes.c_void_p * A)()
    for i, t in enumerate(dest_a):
        arr[i] = t.data_ptr()
    rows_a = (arr, Nr, dest_a)
    rows_b = b.window_rows(dest_b)
    two = bool(b.capabilities() & N.CAP_OBSERVE_ROWS)
    gen = np.random.default_rng(5)
    nact = len(ws.action_dy)
    for t in range(1, 5):
        a.sweep_observe_rows(rows_a, sweep=t != 3, turn=t)
        b.step(sweep=t != 3, agent_begin=0, agent_end=0, write_obs=False, turn=t)
        if two:
            b.observe_rows(rows_b)
        assert co.step(0, t, sweep=t != 3, write_obs=False, a0=0, a1=0) == 0
        co.observe()
        torch.cuda.synchronize()
        if t == 1:
            info = a.launch_info()
            srows = info.split("sweep_rows=")[1]
            assert (inst in srows) if inst.startswith("step_fast_rows") else (srows.startswith(inst) and srows.rstrip().endswith(", true>")), (name, info)
        assert np.array_equal(a.grid.cpu().numpy(), co.grid) and torch.equal(a.grid, b.grid), (name, t, "grid after the sweep")
        pos = a.agent_pos.cpu().numpy()
        for k in range(A):
            mine = dest_a[k].cpu().numpy()
            assert np.array_equal(mine[:, :Nw], co.obs[:, k].reshape(E, Nw)), (name, t, k, "window vs the oracle")
            if two:
                assert torch.equal(dest_a[k], dest_b[k]), (name, t, k, "vs sweep + observe_rows")
            if tail == "it":
                assert np.array_equal(mine[:, Nw] != 0, co.agent_state[:, k] == ws.tag_it_type), (name, t, k, "it flag")
            elif tail:
                want = table.cpu().numpy()[pos[:, k, 0], pos[:, k, 1]]
                assert np.array_equal(mine[:, Nw:], want), (name, t, k, "positional code")
            assert bool((bufs_a[k][:guard] == -9.0).all()) and bool((bufs_a[k][-guard:] == -9.0).all()), (name, t, k, "guards")
        acts = gen.integers(0, nact, (E, A)).astype(np.uint8)          # the agents act (a whole-turn step without a sweep), then the next turn
        ta = torch.from_numpy(acts).cuda()
        for e in (a, b):
            e.step(ta, sweep=False, write_obs=False, turn=t)
        assert co.step(0, t, actions=acts, sweep=False, write_obs=False) == 0
    assert a.status() == 0 and b.status() == 0


@pytest.mark.parametrize("mode", ["cleanup", "tag", "move"])
@pytest.mark.parametrize("seed", range(max(6, int(__import__("os").environ.get("SGW_SOAK", "0")) // 4)))
def test_fused_rows_soak_random_rule_worlds(torch_cuda, mode, seed):
    """Random layered rule worlds (tests/helpers.random_rule_world: BECOME_IF tables, timers, spawners; Cleanup / Tag / plain agents) from an
    injected grid: the fused launch where the engine offers it == sweep alone + sgw_observe_rows == the oracle."""
    torch = torch_cuda
    ws, g, pos = _rule_world(100 * seed + {"cleanup": 1, "tag": 2, "move": 3}[mode], mode)
    E, A = 13, ws.num_agents
    N.set_option("group", 64)                 # (a wave per env: the packed kernels of small worlds have no fused launch)
    a, b = make_engine(ws, E), make_engine(ws, E)
    N.reset_options()
    if not (a.capabilities() & N.CAP_SWEEP_ROWS) or not (b.capabilities() & N.CAP_OBSERVE_ROWS):
        pytest.skip(f"no fused / row instance for this world: {a.launch_info().split(' group')[0]}")
    co = H.COracle(ws, E)
    for e in (a, b):
        e.grid.copy_(torch.from_numpy(np.broadcast_to(g, (E,) + g.shape).copy()).cuda())
        e.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos, (E,) + pos.shape).copy()).cuda())
    co.grid[...] = g
    co.pos[...] = pos
    if a.agent_state is not None:
        co.agent_state[...] = a.agent_state.cpu().numpy()
    Nw = int(np.prod(ws.obs_shape[1:]))
    dest_a = [torch.full((E, Nw), -9.0, device="cuda:0") for _ in range(A)]
    dest_b = [torch.full((E, Nw), -9.0, device="cuda:0") for _ in range(A)]
    ra, rb = a.window_rows(dest_a), b.window_rows(dest_b)
    for t in range(1, 4):
        a.sweep_observe_rows(ra, sweep=True, turn=t)
        b.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t)
        b.observe_rows(rb)
        assert co.step(0, t, sweep=True, write_obs=False, a0=0, a1=0) == 0
        co.observe()
        torch.cuda.synchronize()
        assert np.array_equal(a.grid.cpu().numpy(), co.grid), (mode, seed, t)
        for k in range(A):
            assert torch.equal(dest_a[k], dest_b[k]), (mode, seed, t, k)
            assert np.array_equal(dest_a[k].cpu().numpy(), co.obs[:, k].reshape(E, Nw)), (mode, seed, t, k)
        acts = a.random_actions(turn=t).clone()
        for e in (a, b):
            e.step(acts, sweep=False, write_obs=False, turn=t)
        assert co.step(0, t, actions=acts.cpu().numpy(), sweep=False, write_obs=False) == 0
    assert a.status() == 0 and b.status() == 0


# ------------------------------------------------------------------ moved here from the by-round files of rounds 2-5 (bodies unchanged)
import json  # noqa: E402,F401
import os  # noqa: E402,F401
import subprocess  # noqa: E402,F401
import sys  # noqa: E402,F401

from oracle import gridstep_oracle as O  # noqa: E402,F401
from sorrel_amd import _native as N  # noqa: E402,F401
from tests import helpers as H  # noqa: E402,F401
from tests.gpu_common import *  # noqa: E402,F401,F403


def test_policy_turn_writes_windows_straight_into_replay_rows(torch_cuda):
    """Environment.take_turn with policy models: each agent's window is rendered into the row of its replay buffer that
    add_memory fills (no copy), a model shared by all agents and a pov that appends to the window fall back to the
    observation tensor -- and all of it stores exactly what the copy path stores."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from sorrel_amd.environment import Environment
    from tests.test_gpu_api import make_env

    E, T = 19, 7

    def factory(shared):
        made = []

        class Policy(BaseModel):
            def __init__(self, input_size, action_space):
                super().__init__(input_size, action_space, memory_size=5, num_envs=E, device="cuda:0")
                self.seen = []

            def take_action(self, state):
                self.seen.append(state.data_ptr())
                return (state.reshape(state.shape[0], -1).sum(dim=1).long() * 5 + 1) % 4

        def make(input_size, action_space):
            if shared and made:
                return made[0]
            made.append(Policy(input_size, action_space))
            return made[-1]

        return make

    for shared in (False, True):
        ref = None
        for patch in (True, False):     # round 3's protocol (all windows once + sgw_act repairs) and the 1 + A protocol
            runs = []
            for direct in (True, False):
                env = make_env(15, 17, 3, 2, E, p=0.05, model_factory=factory(shared))
                env.write_obs_into_replay = direct
                env.patch_windows = patch
                for _ in range(T):
                    env.take_turn()
                torch.cuda.synchronize()
                runs.append(env)
            d, c = runs
            ref = ref or c
            for other in (d, c):
                for ad, ac in zip(other.agents, ref.agents):
                    md, mc = ad.model.memory, ac.model.memory
                    assert md.idx == mc.idx and md.size == mc.size
                    assert torch.equal(md.states, mc.states) and torch.equal(md.actions, mc.actions) and torch.equal(md.rewards, mc.rewards)
                assert torch.equal(other.world.grid, ref.world.grid) and torch.equal(other.world.total_reward, ref.world.total_reward)
            rows = {d.agents[0].model.memory.states[i].data_ptr() for i in range(5)}
            in_rows = [p in rows for p in d.agents[0].model.seen]
            if shared and not patch:      # 1 + A protocol: only agent 0's window (rendered by the sweep launch, nobody adds in between) can go straight in
                assert any(in_rows) and not all(in_rows)
            else:                         # own buffers; or all windows rendered up front into consecutive rows of the shared one
                assert all(in_rows), "every state the policy saw was already sitting in its replay row"
        assert not any(p in {c.agents[0].model.memory.states[i].data_ptr() for i in range(5)} for p in c.agents[0].model.seen)


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


def test_agent_major_windows_in_one_launch_and_gather_rows(torch_cuda):
    """SGW_STEP_OBS_AGENT_MAJOR (workgroup-per-env kernels): the sweep and every agent's pre-move window into [A][E][C*V*V] rows in ONE
    launch = the sweep alone + sgw_observe_rows; a whole turn with moves into such rows = the [E][A][...] tensor transposed -- on the
    walking, the staged and the direct-store variant.  sgw_gather_rows = index_select."""
    torch = torch_cuda
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec
    from tests.test_gpu_parity import make_engine

    ws = treasurehunt_spec(72, 80, 20, 4, spawn_prob=0.05, seed=12, dense_prob=0.2)
    for opts in ({}, {"big_walk_blocks": 3}, {"big_walk": 0, "big_stage": 1}, {"big_walk": 0, "big_stage": 0}):
        for k, v in opts.items():
            N.set_option(k, v)
        E = 23
        a, b = make_engine(ws, E), make_engine(ws, E)
        assert a.capabilities() & N.CAP_OBS_AGENT_MAJOR
        for e in (a, b):
            e.reset(0)
        rows_a = torch.full((ws.num_agents, E, int(np.prod(ws.obs_shape[1:]))), -5.0, device="cuda:0")
        a.step(a.actions, sweep=True, no_move=True, turn=1, obs_out=rows_a, agent_major=True)
        b.step(b.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=1)
        rows_b = b.speculation_windows()
        torch.cuda.synchronize()
        assert torch.equal(a.grid, b.grid) and torch.equal(rows_a, rows_b), opts
        a.step(random_actions=True, turn=2, obs_out=rows_a, agent_major=True)           # a whole turn, windows agent-major
        b.step(random_actions=True, turn=2)
        torch.cuda.synchronize()
        assert torch.equal(a.grid, b.grid) and torch.equal(a.rewards, b.rewards) and torch.equal(a.total_reward, b.total_reward)
        assert torch.equal(rows_a, b.obs.reshape(E, ws.num_agents, -1).permute(1, 0, 2).contiguous()), opts
        N.reset_options()
    flat = rows_a.view(-1, rows_a.shape[2])
    idx = torch.randint(0, flat.shape[0], (1000,), device="cuda:0")
    assert torch.equal(a.gather_rows(flat, idx), flat.index_select(0, idx))
    small = make_engine(treasurehunt_spec(16, 16, 4, 2), 8)                              # a wave-per-env world: not offered, and refused
    assert not (small.capabilities() & N.CAP_OBS_AGENT_MAJOR)
    with pytest.raises(ValueError):
        small.step(random_actions=True, obs_out=small.speculation_rows(), agent_major=True)


@pytest.mark.parametrize("case", SWEEP_ROWS_CASES, ids=[c[0] for c in SWEEP_ROWS_CASES])
def test_sweep_and_every_window_into_rows_in_one_launch(torch_cuda, case):
    """sgw_sweep_observe_rows (SGW_CAP_SWEEP_ROWS) = sgw_step(sweep only) + sgw_observe_rows = the C oracle's sweep followed by every agent's
    window: the grid after the sweep and every row, bit for bit, over several turns with acts in between (destinations in separate
    allocations, 8 bytes off a 16-byte boundary for odd envs, guard elements around them untouched)."""
    torch = torch_cuda
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec
    from tests.test_gpu_parity import make_engine

    name, (h, w, A, r), E, opts = case
    for k, v in opts.items():
        N.set_option(k, v)
    ws = treasurehunt_spec(h, w, A, r, spawn_prob=0.04, seed=21, dense_prob=0.15)
    a, b = make_engine(ws, E), make_engine(ws, E)
    N.reset_options()
    co = H.COracle(ws, E, first_env_id=0)
    assert a.capabilities() & N.CAP_SWEEP_ROWS, a.plan() if hasattr(a, "plan") else name
    for e in (a, b):
        e.reset(0)
    co.reset(0)
    Nw = int(np.prod(ws.obs_shape[1:]))
    guard = 7
    bufs_a = [torch.full((guard + E * Nw + guard,), -9.0, device="cuda:0") for _ in range(A)]
    dest_a = [buf[guard:guard + E * Nw].view(E, Nw) for buf in bufs_a]
    dest_b = [torch.full((E, Nw), -9.0, device="cuda:0") for _ in range(A)]
    rows_a = (N_ptr_array(dest_a), Nw, dest_a)
    rows_b = b.window_rows(dest_b)
    two = bool(b.capabilities() & N.CAP_OBSERVE_ROWS)               # (radius 0 has no row-load instance: the oracle alone checks that case)
    gen = np.random.default_rng(5)
    for t in range(1, 6):
        a.sweep_observe_rows(rows_a, sweep=t != 3, turn=t)
        b.step(sweep=t != 3, agent_begin=0, agent_end=0, write_obs=False, turn=t)
        if two:
            b.observe_rows(rows_b)
        assert co.step(0, t, sweep=t != 3, write_obs=False, a0=0, a1=0) == 0
        co.observe()
        torch.cuda.synchronize()
        assert np.array_equal(a.grid.cpu().numpy(), co.grid) and torch.equal(a.grid, b.grid), (name, t, "grid after the sweep")
        for k in range(A):
            assert not two or torch.equal(dest_a[k], dest_b[k]), (name, t, k)
            assert np.array_equal(dest_a[k].cpu().numpy(), co.obs[:, k].reshape(E, Nw)), (name, t, k, "oracle")
            assert bool((bufs_a[k][:guard] == -9.0).all()) and bool((bufs_a[k][-guard:] == -9.0).all()), (name, t, k, "guards")
        acts = gen.integers(0, 4, (E, A)).astype(np.uint8)          # the agents act (one whole-turn step without a sweep), then the next turn
        ta = torch.from_numpy(acts).cuda()
        for e in (a, b):
            e.step(ta, sweep=False, write_obs=False, turn=t)
        assert co.step(0, t, actions=acts, sweep=False, write_obs=False) == 0
    assert a.status() == 0 and b.status() == 0
    with pytest.raises(ValueError):                                   # rows of another size are refused, not written
        a.sweep_observe_rows((rows_a[0], Nw + 2, None))


def test_engines_without_the_fused_instance_say_so(torch_cuda):
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec
    from tests.test_gpu_parity import make_engine

    big = make_engine(treasurehunt_spec(72, 80, 6, 3), 5)                       # workgroup per env
    odd = make_engine(treasurehunt_spec(20, 20, 3, 1), 5)                       # 6 * 9 = 54 elements per window: even, offered; 3 agents * 54 % 4 != 0: no whole-env burst
    N.set_option("jit", 0)                                                      # (round 6: specialised chunk-staging instances have a fused twin; the prebuilt ones do not)
    plain = make_engine(treasurehunt_spec(20, 20, 3, 1), 5)
    N.reset_options()
    for eng in (big, odd, plain):
        if eng.capabilities() & N.CAP_SWEEP_ROWS:
            continue
        rows = eng.window_rows([torch_cuda.zeros((5, int(np.prod(eng.spec.obs_shape[1:]))), device="cuda:0") for _ in range(eng.spec.num_agents)])
        with pytest.raises(ValueError):
            eng.sweep_observe_rows(rows)
    assert big.capabilities() & N.CAP_SWEEP_ROWS                                # (round 6: step_big renders into per-agent rows itself)
    assert not (plain.capabilities() & N.CAP_SWEEP_ROWS)

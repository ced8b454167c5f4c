"""``sgw_sweep_observe_rows`` -- the entity sweep and EVERY agent's window into its own row, one launch -- on the kernels that got it in round 6:
the workgroup-per-env kernel (``step_big``: worlds above 4 KiB, plain and Tag movers, any appearance table) and the chunk-staging wave-per-env
instances (``step_fast_rowsx``: layered rule sets -- Cleanup --, Tag, run-time maps), row tails included.  Checked against the two-launch form
(sweep alone + ``sgw_observe_rows``) and the C oracle.  Reference: ``Agent.transition`` / ``pov``, ``sorrel/agents/agent.py:155-173``; the tails:
``sorrel/examples/tag/agents.py:57-65``, ``sorrel/examples/cleanup/agents.py:52-60``.  (The compile-time-shape instance of round 5 is covered in
``test_gpu_round5.py``.)"""
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
    from tests.test_gpu_round2 import _tag_spec

    return _tag_spec(h, w, a, r)


def _cleanup():
    from tests.test_gpu_round2 import _cleanup_spec

    return _cleanup_spec()[0]


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
    ("c5_shape_walking", lambda: _th(128, 128, 64, 5), 23, {"big_walk_blocks": 5}, "step_big<true, 2, 6, 5, false, true", None),
    ("c5_shape_prebuilt", lambda: _th(128, 128, 64, 5), 5, {"jit": 0}, "step_big<true, 2, 6, 5", None),
    ("big_72x80_r3_tail7", lambda: _th(72, 80, 6, 3), 9, {"big_stage": 1}, "step_big<", 7),
    ("big_70x66_r4_256_threads", lambda: _th(70, 66, 10, 4), 11, {}, "step_big<", None),
    ("big_float_table", lambda: _float_table(_th(72, 80, 6, 3)), 7, {}, "step_big<false", None),
    ("big_tag_70x80_it", lambda: _tag(70, 80, 9, 4), 8, {}, "step_big<", "it"),
    ("cleanup_15x16_rules_table12", _cleanup, 19, {}, "step_fast_rowsx<", 12),
    ("cleanup_15x16_rules_no_tail", _cleanup, 19, {}, "step_fast_rowsx<", None),
    ("tag_11x11_it_whole_env", lambda: _tag(11, 11, 5, 4), 37, {"group": 64}, "step_fast_rows<1, 4, 4, 11, 11, true>", "it"),
    ("tag_70x60_it_chunked", lambda: _tag(70, 60, 7, 4), 14, {"fast_8k": 1}, "step_fast_rowsx<", "it"),
    ("tag_13x12_prebuilt_has_none", lambda: _tag(13, 12, 4, 2), 9, {"group": 64, "jit": 0}, None, "it"),
    ("runtime_map_33x35_r3", lambda: _th(33, 35, 7, 3), 21, {"burst": 2}, "step_fast_rowsx<", None),
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
            assert (inst in info.split("sweep_rows=")[1]) if inst.startswith("step_fast_rows") else (inst in info and "sweep_rows=the-step-kernel" in info), (name, info)
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
@pytest.mark.parametrize("seed", range(6))
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

"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the
C ABI, against (1) the golden vectors produced by the reference itself, (2) the C oracle
on the same seeded inputs at sizes it finishes in seconds, and (3) size-independent
properties at BASELINE.json's full sizes.  Integer / byte / index work: bit-exact.
Observations and rewards are float32 holding exactly representable values and are
compared bit-exactly as well (np.array_equal)."""
import os

import numpy as np
import pytest

from tests import helpers as H
from sorrel_amd import _native as N

pytestmark = pytest.mark.gpu

COUNTER_FIXTURES = [n for n in H.golden_names() if n != "stock_np_random"]


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no silent CPU fallback)")
    return torch


def make_engine(ws, E, first=0, **kw):
    from sorrel_amd.engine import GridEngine

    return GridEngine(ws, E, device="cuda:0", first_env_id=first, **kw)


def assert_same(eng, co, what=("grid", "pos", "actions", "obs", "rewards", "total"), ctx=""):
    import torch

    torch.cuda.synchronize()
    pairs = dict(grid=(eng.grid, co.grid), pos=(eng.agent_pos, co.pos), actions=(eng.actions, co.actions),
                 obs=(eng.obs, co.obs), rewards=(eng.rewards, co.rewards), total=(eng.total_reward, co.total))
    for k in what:
        mine, ref = pairs[k]
        mine = mine.cpu().numpy()
        if not np.array_equal(mine, ref):
            bad = np.argwhere(mine != ref)
            raise AssertionError(f"{ctx}: {k} differs from the oracle at {bad[:5].tolist()} ({len(bad)} elements)")


# ------------------------------------------------------------------ (1) golden vectors from the reference
@pytest.mark.parametrize("name", COUNTER_FIXTURES)
def test_hip_matches_reference_golden(torch_cuda, name):
    torch = torch_cuda
    d, spec = H.load_golden(name)
    ws = H.world_spec(spec)
    turns = d["obs"].shape[0]
    epoch = int(d["epoch"]) if "epoch" in d else 0
    for n, env_id in enumerate(int(e) for e in d["env_ids"]):
        eng = make_engine(ws, 1, first=env_id)
        if name in H.INJECTED_FIXTURES:      # worlds populated by host code: start from the stored state
            eng.grid.copy_(torch.from_numpy(d["grid0"][n][None]))
            eng.agent_pos.copy_(torch.from_numpy(d["pos0"][n][None]))
            eng.total_reward.zero_()
            eng.epoch = epoch
        else:
            eng.reset(epoch=epoch)
            torch.cuda.synchronize()
            assert np.array_equal(eng.grid.cpu().numpy()[0], d["grid0"][n]), f"{name}: reset grid"
            assert np.array_equal(eng.agent_pos.cpu().numpy()[0], d["pos0"][n]), f"{name}: reset positions"
        for t in range(turns):
            if "scripted" in d:
                eng.step(torch.from_numpy(d["scripted"][t, n][None].astype(np.uint8)).cuda())
            else:
                eng.step(random_actions=True)
            torch.cuda.synchronize()
            assert np.array_equal(eng.obs.cpu().numpy()[0], d["obs"][t, n]), f"{name}: obs turn {t}"
            assert np.array_equal(eng.actions.cpu().numpy()[0], d["actions"][t, n]), f"{name}: actions turn {t}"
            assert np.array_equal(eng.rewards.cpu().numpy()[0], d["rewards"][t, n]), f"{name}: rewards turn {t}"
            assert eng.total_reward.cpu().numpy()[0] == d["total_reward"][t, n], f"{name}: total_reward turn {t}"
            assert np.array_equal(eng.grid.cpu().numpy()[0], d["grid"][t, n]), f"{name}: grid turn {t}"
            assert np.array_equal(eng.agent_pos.cpu().numpy()[0], d["pos"][t, n]), f"{name}: pos turn {t}"
            if "agent_state" in d:      # interaction rules (Tag): who is "it" now / was when observing
                assert np.array_equal(eng.agent_state.cpu().numpy()[0], d["agent_state"][t, n]), f"{name}: agent_state turn {t}"
                assert np.array_equal(eng.state_at_pov.cpu().numpy()[0], d["state_at_pov"][t, n]), f"{name}: state_at_pov turn {t}"
            if "agent_dir" in d:        # Cleanup: the facing that aims the beams
                assert np.array_equal(eng.agent_dir.cpu().numpy()[0], d["agent_dir"][t, n]), f"{name}: agent_dir turn {t}"
        assert eng.status() == 0


def test_hip_from_injected_state_matches_golden(torch_cuda):
    """State set from the fixture (not by sgw_reset), then stepped: add()/populate plumbing path."""
    torch = torch_cuda
    d, spec = H.load_golden("crowded_6x6")
    ws = H.world_spec(spec)
    ids = [int(e) for e in d["env_ids"]]
    for n, env_id in enumerate(ids):
        eng = make_engine(ws, 1, first=env_id)
        eng.grid.copy_(torch.from_numpy(d["grid0"][n][None]))
        eng.agent_pos.copy_(torch.from_numpy(d["pos0"][n][None]))
        eng.total_reward.zero_()
        for t in range(d["obs"].shape[0]):
            eng.step(random_actions=True)
            assert np.array_equal(eng.obs.cpu().numpy()[0], d["obs"][t, n])
            assert np.array_equal(eng.grid.cpu().numpy()[0], d["grid"][t, n])


# ------------------------------------------------------------------ (2) HIP vs C oracle, seeded, bigger
def assert_instance(eng, jit):
    """With specialised instances on, what launches is the instance compiled for THIS engine (its template-id carries the
    engine's own layers / channels / radius / map) unless the library holds exactly that instance; with them off, a prebuilt one."""
    info = eng.launch_info()
    if not jit:
        assert "specialised=0" in info, info
        return
    plan = N.plan(eng.config)
    name = info.split(" group=")[0]
    # (a refused compile would launch the prebuilt twin, whose template-id differs from the one the plan asks for)
    assert plan["specialised"] == 1 and name in (plan["kernel"], plan["kernel_walk"]), (info, plan["kernel"], plan["kernel_walk"])


def rollout_vs_oracle(ws, E, T, first=0, epoch=0, check_every=1, jit=None):
    eng = make_engine(ws, E, first=first)
    if jit is not None:
        assert_instance(eng, jit)
    co = H.COracle(ws, E, first_env_id=first)
    eng.reset(epoch=epoch)
    co.reset(epoch)
    assert_same(eng, co, ("grid", "pos", "total"), ctx="reset")
    for t in range(1, T + 1):
        eng.step(random_actions=True)
        st = co.step(epoch, t, random_actions=True)
        assert st == 0
        if t % check_every == 0 or t == T:
            assert_same(eng, co, ctx=f"turn {t}")
    assert eng.status() == 0
    return eng, co


def test_config2_full_batch_vs_oracle(torch_cuda):
    """BASELINE config 2 at full size: 16x16, 4 agents, 5x5 window, 4096 envs."""
    from sorrel_amd.spec import treasurehunt_spec

    rollout_vs_oracle(treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.005, seed=0), 4096, 12)


def test_config3_shape_vs_oracle(torch_cuda):
    """Headline shape 32x32, 8 agents, 7x7 window on 8192 envs (oracle finishes in seconds)."""
    from sorrel_amd.spec import treasurehunt_spec

    rollout_vs_oracle(treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=1), 8192, 8, first=60000, check_every=2)


def test_config5_shape_vs_oracle(torch_cuda):
    """LDS-tile stress shape: 128x128, 64 agents, 11x11 window, dense entities (workgroup per env)."""
    from sorrel_amd.spec import treasurehunt_spec

    rollout_vs_oracle(treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=2, dense_prob=0.25), 48, 4, first=16000)


def test_cleanup_rule_batch_vs_oracle(torch_cuda):
    """Cleanup (layered world, ordered sweep with BECOME_IF rules, beams, facing) on a few hundred
    envs started from the fixture's hand-built river/orchard map; every env draws its own spawns
    and actions, so the trajectories diverge."""
    import torch
    d, spec = H.load_golden("cleanup_15x16")
    ws = H.world_spec(spec)
    E, T = 257, 40
    eng = make_engine(ws, E, first=9)
    co = H.COracle(ws, E, first_env_id=9)
    g0, p0 = d["grid0"][0], d["pos0"][0]
    eng.grid.copy_(torch.from_numpy(np.broadcast_to(g0, (E,) + g0.shape).copy()))
    eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(p0, (E,) + p0.shape).copy()))
    eng.total_reward.zero_()
    co.grid[...] = g0
    co.pos[...] = p0
    co.total[...] = 0
    for t in range(1, T + 1):
        eng.step(random_actions=True)
        assert co.step(0, t, random_actions=True) == 0
        assert_same(eng, co, ctx=f"cleanup turn {t}")
        assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), f"cleanup turn {t}: agent_dir"
    assert eng.status() == 0
    assert len(np.unique(eng.total_reward.cpu().numpy())) > 3      # the envs really diverged


def test_tag_rule_batch_vs_oracle(torch_cuda):
    """Tag (agent <-> agent interaction) on a few hundred envs, incl. a reset in the middle: the
    "it" flag survives resets (agents are not re-created), positions do not."""
    import torch
    d, spec = H.load_golden("tag_9x9")
    ws = H.world_spec(spec)
    E = 300
    eng = make_engine(ws, E, first=50)
    co = H.COracle(ws, E, first_env_id=50)
    assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state)          # same initial "it" draw
    assert ((co.agent_state == ws.tag_it_type).sum(axis=1) == 1).all()
    for epoch in (0, 1):
        eng.reset(epoch=epoch)
        co.reset(epoch)
        assert_same(eng, co, ("grid", "pos"), ctx=f"reset {epoch}")
        for t in range(1, 21):
            eng.step(random_actions=True)
            co.step(epoch, t, random_actions=True)
            assert_same(eng, co, ctx=f"epoch {epoch} turn {t}")
            torch.cuda.synchronize()
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state)
            assert np.array_equal(eng.state_at_pov.cpu().numpy(), co.state_at_pov)
        assert ((co.agent_state == ws.tag_it_type).sum(axis=1) == 1).all()       # exactly one "it" per env, always


@pytest.mark.parametrize("name", ["c2_treasurehunt_16x16", "crowded_6x6", "tag_9x9", "tag_crowded_6x7", "rgb_treasurehunt",
                                  "c5_small_dense", "basic_doublewall", "cleanup_15x16", "tag_11x11_default", "cleanup_21x31_default"])
def test_generic_kernel_matches_reference_golden(torch_cuda, name, monkeypatch):
    """The fallback kernel (any shape, every rule) on fixtures the specialised kernels would take."""
    N.set_option("force_generic", 1)
    test_hip_matches_reference_golden(torch_cuda, name)


def test_generic_kernel_batches_vs_oracle(torch_cuda, monkeypatch):
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("force_generic", 1)
    rollout_vs_oracle(treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.01, seed=3), 1000, 5)                       # wave per env
    rollout_vs_oracle(treasurehunt_spec(96, 80, 40, 4, spawn_prob=0.05, seed=4, dense_prob=0.2), 30, 4)        # workgroup per env
    d, spec = H.load_golden("tag_9x9")
    ws = H.world_spec(spec)
    eng, co = make_engine(ws, 200, first=9), H.COracle(ws, 200, first_env_id=9)
    eng.reset(0)
    co.reset(0)
    for t in range(1, 15):
        eng.step(random_actions=True)
        co.step(0, t, random_actions=True)
        assert_same(eng, co, ctx=f"generic tag turn {t}")
        assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state)


@pytest.mark.parametrize("case", ["tag_70x80", "tag_crowded_66x64", "cleanup_40x48", "cleanup_wide_beam", "move_phased", "move_wide_window",
                                  "observe_only", "u8_obs"])
def test_workgroup_per_env_generic_kernel_ticket_order(torch_cuda, case, monkeypatch):
    """Worlds above 4 KiB on the generic kernel (step_kernel<256>: what Tag / Cleanup worlds of that size run on): the
    four waves of a workgroup take the agents in turn behind an LDS ticket, window bytes captured before the act and
    stored after it.  Tag (flags handed from wave to wave, crowded so that many tags happen), Cleanup (beams, ordered
    sweep; the widest beam the library takes, 3R = 63 cells), plain moves stepped agent by agent with OBS_NEXT, a 13x13
    window (more cells than the registers hold: the rest is rendered before the act), sgw_observe, uint8 observations."""
    torch = torch_cuda
    import dataclasses
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("force_generic", 1)
    kw = {}
    grid0 = pos0 = None
    if case.startswith("tag"):
        d, spec = H.load_golden("tag_11x11_default")
        ws = H.world_spec(spec)
        h, w, a = (70, 80, 12) if case == "tag_70x80" else (66, 64, 64)
        ws = dataclasses.replace(ws, height=h, width=w, num_agents=a, agent_type=[ws.agent_type[0]] * a)
        E, T = (13, 25) if case == "tag_70x80" else (7, 12)
    elif case.startswith("cleanup"):
        d, spec = H.load_golden("cleanup_15x16")
        ws = H.world_spec(spec)
        h, w, a = 40, 48, 10
        ws = dataclasses.replace(ws, height=h, width=w, num_agents=a, agent_type=[ws.agent_type[0]] * a,
                                 beam_radius=21 if case == "cleanup_wide_beam" else ws.beam_radius)
        # the reference's map at this size (type ids as in oracle/make_golden.py: 1 sand, 2 wall, 3 river, 5 apple tree, 11 agent)
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
        E, T = 9, 14
    else:
        r = 6 if case == "move_wide_window" else 4
        ws = treasurehunt_spec(72, 64, 23, r, spawn_prob=0.05, seed=41, dense_prob=0.25)
        E, T = 11, 5
        if case == "u8_obs":
            kw["obs_dtype"] = torch.uint8
    eng = make_engine(ws, E, first=17, **kw)
    assert "step_kernel<" in eng.launch_info() and "group=256 " in eng.launch_info()
    co = H.COracle(ws, E, first_env_id=17)
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
    what = ("grid", "pos", "actions", "rewards", "total") if case == "u8_obs" else ("grid", "pos", "actions", "obs", "rewards", "total")
    for t in range(1, T + 1):
        if case == "move_phased":       # 1 + A launches, each rendering the NEXT agent's window after its own agent's move
            assert co.step(0, t, random_actions=True) == 0
            acts = torch.from_numpy(co.actions.copy()).cuda()
            eng.obs.fill_(-3.0)
            eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
            for a in range(ws.num_agents):
                eng.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, obs_next=a + 1 < ws.num_agents, write_obs=False, turn=t, advance_turn=False)
        else:
            eng.step(random_actions=True, turn=t, advance_turn=False)
            assert co.step(0, t, random_actions=True) == 0
        assert_same(eng, co, what, ctx=f"{case} turn {t}")
        if case == "u8_obs":
            torch.cuda.synchronize()
            assert np.array_equal(eng.obs.cpu().numpy().astype(np.float32), co.obs), f"{case} turn {t}: uint8 observations"
        if case.startswith("tag"):
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state) and np.array_equal(eng.state_at_pov.cpu().numpy(), co.state_at_pov)
        if case.startswith("cleanup"):
            assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir)
    if case == "observe_only":
        eng.obs.zero_()
        eng.observe()
        co.observe()
        assert_same(eng, co, ("obs",), ctx="observe")
        eng.observe(5, 9)
        co.observe(5, 9)
        assert_same(eng, co, ("obs",), ctx="observe range")
    if case.startswith("tag"):
        assert len(np.unique(eng.agent_state.cpu().numpy(), axis=0)) > 1        # the flag really moved around
    assert eng.status() == 0


@pytest.mark.parametrize("group", ["16", "32"])
@pytest.mark.parametrize("name", ["c1_treasurehunt_10x10", "c2_treasurehunt_16x16", "crowded_6x6", "tag_9x9", "tag_crowded_6x7",
                                  "rgb_treasurehunt", "basic_doublewall", "float_appearance_3layer", "ragged_9x13_rmax",
                                  "scripted_noop", "cleanup_15x16", "basic_1layer", "c3_treasurehunt_32x32", "tag_11x11_default",
                                  "cleanup_13x12_r2", "cleanup_21x31_default"])
def test_packed_kernels_match_reference_golden(torch_cuda, name, group, monkeypatch):
    """Two / four envs per wave (step_kernel<32> / <16>): every fixture small enough, bit for bit."""
    N.set_option("group", int(group))
    test_hip_matches_reference_golden(torch_cuda, name)


@pytest.mark.parametrize("group", ["16", "32"])
def test_packed_kernels_batches_vs_oracle(torch_cuda, group, monkeypatch):
    """Batches that do not fill the last wave / workgroup, ragged byte counts, windows wider than the group (several
    render passes), the uint8 format, phased stepping with OBS_NEXT, Tag -- on the packed kernels."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("group", int(group))
    for (h, w, a, r, E) in ((21, 21, 2, 2, 1003), (10, 10, 2, 2, 517), (16, 16, 4, 2, 64), (13, 9, 5, 4, 77), (32, 32, 8, 3, 130)):
        eng, co = rollout_vs_oracle(treasurehunt_spec(h, w, a, r, spawn_prob=0.05, seed=31, dense_prob=0.1), E, 5, first=11)
        assert "step_kernel<" in eng.launch_info() and f"group={group} " in eng.launch_info()
    for fixture in ("tag_9x9", "tag_11x11_default"):      # the second is the shape with its own static instance (G = 32)
        d, spec = H.load_golden(fixture)
        ws = H.world_spec(spec)
        eng, co = make_engine(ws, 333, first=9), H.COracle(ws, 333, first_env_id=9)
        eng.reset(0)
        co.reset(0)
        for t in range(1, 12):
            eng.step(random_actions=True)
            co.step(0, t, random_actions=True)
            assert_same(eng, co, ctx=f"packed {fixture} turn {t}")
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state)
    # compact uint8 observations
    ws = treasurehunt_spec(21, 21, 3, 2, spawn_prob=0.05, seed=5)
    e8, e32 = make_engine(ws, 200, obs_dtype=torch.uint8), make_engine(ws, 200)
    for e in (e8, e32):
        e.reset(0)
    for _ in range(4):
        e8.step(random_actions=True)
        e32.step(random_actions=True)
    torch.cuda.synchronize()
    assert torch.equal(e8.obs.float(), e32.obs) and torch.equal(e8.grid, e32.grid)


def test_dispatch_rule_packs_small_worlds_of_large_batches(torch_cuda):
    """The automatic rule (sgw_create): 65 536 envs of the reference's Treasurehunt default (21x21, 2 agents, r = 2) run
    four to a wave, a small batch of the same world and the BASELINE config-3 shape stay on the wave-per-env kernel;
    a strided sample of the big packed batch is checked against the oracle."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(21, 21, 2, 2, spawn_prob=0.02, seed=77)
    big = make_engine(ws, 65536, first=100)
    assert "step_kernel<16, true, 2, 6, 0, 2, 21, 21>" in big.launch_info() and "group=16 " in big.launch_info()   # specialised for this map and window
    with N.options(jit=0):      # the prebuilt instances: the example's static 5x5 window; another radius: run-time shape
        assert "step_kernel<16, true, 2, 6, SGW_AGENT_RULE_MOVE, 2> group=16 " in make_engine(ws, 65536).launch_info()
        assert "step_kernel<G, true, 2, 6> group=16 " in make_engine(treasurehunt_spec(21, 21, 2, 1), 65536).launch_info()
    assert "step_fast" in make_engine(ws, 512).launch_info()
    assert "step_fast<true, 2, 6, 3, 32, 32>" in make_engine(treasurehunt_spec(32, 32, 8, 3), 65536).launch_info()
    big.reset(0)
    ids = list(range(0, 65536, 4099)) + [65535]
    cos = []
    for i in ids:
        co = H.COracle(ws, 1, first_env_id=100 + i)
        co.reset(0)
        cos.append(co)
    for t in range(1, 7):
        big.step(random_actions=True)
        torch.cuda.synchronize()
        for i, co in zip(ids, cos):
            co.step(0, t, random_actions=True)
            assert np.array_equal(big.obs[i].cpu().numpy(), co.obs[0]), (t, i)
            assert np.array_equal(big.grid[i].cpu().numpy(), co.grid[0]) and big.total_reward[i].item() == co.total[0]
    assert big.status() == 0


@pytest.mark.parametrize("shape", [(32, 32, 8, 3, 300), (21, 21, 2, 2, 100), (128, 128, 64, 5, 6)])
def test_compact_uint8_observations(torch_cuda, shape, monkeypatch):
    """SGW_OBS_U8: same layout, the float32 counts as bytes (fast, big and generic kernels)."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    h, w, a, r, E = shape
    ws = treasurehunt_spec(h, w, a, r, spawn_prob=0.05, seed=77, dense_prob=0.2)
    for force_generic in ("0", "1"):
        N.set_option("force_generic", int(force_generic))
        f32, u8 = make_engine(ws, E), make_engine(ws, E, obs_dtype=torch.uint8)
        f32.reset(0)
        u8.reset(0)
        for _ in range(4):
            f32.step(random_actions=True)
            u8.step(random_actions=True)
            assert u8.obs.dtype == torch.uint8 and torch.equal(u8.obs.float(), f32.obs)
            assert torch.equal(u8.grid, f32.grid) and torch.equal(u8.rewards, f32.rewards)
        u8.obs.zero_()
        u8.observe()
        f32.observe()
        assert torch.equal(u8.obs.float(), f32.obs)


def test_compact_uint8_needs_a_one_hot_table(torch_cuda):
    torch = torch_cuda
    d, spec = H.load_golden("float_appearance_3layer")
    with pytest.raises(ValueError):
        make_engine(H.world_spec(spec), 4, obs_dtype=torch.uint8)


def test_largest_worlds_vs_oracle(torch_cuda):
    """More than 64 KiB of LDS per env (needs the raised dynamic-LDS limit), up to the 256x256 coordinate limit."""
    from sorrel_amd.spec import treasurehunt_spec

    rollout_vs_oracle(treasurehunt_spec(200, 200, 48, 5, spawn_prob=0.01, seed=15, dense_prob=0.1), 6, 3, first=2)     # 80 KB
    rollout_vs_oracle(treasurehunt_spec(256, 256, 64, 5, spawn_prob=0.01, seed=16, dense_prob=0.05), 4, 2, first=1)    # 128 KiB


def test_big_kernel_crowded_vs_oracle(torch_cuda):
    """Workgroup-per-env kernel under heavy contention: 64 agents on a 46x46 interior with 11x11
    windows, dense items -- many moves touch many windows (journal undo), many agents compete for
    cells and follow each other into vacated cells (register move resolution)."""
    from sorrel_amd.spec import treasurehunt_spec

    rollout_vs_oracle(treasurehunt_spec(48, 48, 64, 5, spawn_prob=0.1, seed=12, dense_prob=0.3), 40, 12, first=7)
    rollout_vs_oracle(treasurehunt_spec(80, 64, 64, 2, spawn_prob=0.02, seed=13), 24, 6, first=3)     # 10 KiB env, one window pass


@pytest.mark.parametrize("blocks", ["1", "5", "16"])
@pytest.mark.parametrize("variant", ["config5_shape", "general_tables", "float_appearance", "scripted_actions", "no_sweep"])
def test_big_kernel_walking_workgroups_vs_oracle(torch_cuda, monkeypatch, blocks, variant):
    """step_big<..., WALK>: fewer workgroups than envs, each walking envs b, b + blocks, ... with the next env's grid,
    positions, actions and total loaded one env ahead (SGW_BIG_WALK_BLOCKS forces the workgroup count; the full-size
    config-5 tests take this path on their own).  Uneven walks (23 envs over 5 or 16 workgroups), one workgroup doing
    everything, the run-time-shape and the float-appearance instantiations, actions given by the caller, no sweep."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("big_walk_blocks", int(blocks))
    if variant == "config5_shape":
        ws = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=21, dense_prob=0.25)
    elif variant == "float_appearance":
        ws = treasurehunt_spec(64, 80, 24, 4, spawn_prob=0.05, seed=23, dense_prob=0.2)
        app = np.asarray(ws.appearance, dtype=np.float64).copy()
        app[app != 0] = 0.375
        ws.appearance = app
    else:
        ws = treasurehunt_spec(80, 64, 40, 3, spawn_prob=0.0 if variant == "no_sweep" else 0.04, seed=22, dense_prob=0.3)
    E, T = 23, 5
    eng = make_engine(ws, E, first=3)
    assert "step_big" in eng.launch_info()
    co = H.COracle(ws, E, first_env_id=3)
    eng.reset(epoch=1)
    co.reset(1)
    rng = np.random.default_rng(5)
    for t in range(1, T + 1):
        if variant == "scripted_actions":
            acts = rng.integers(0, len(ws.action_dy), size=(E, ws.num_agents), dtype=np.uint8)
            eng.step(torch.from_numpy(acts).cuda())
            assert co.step(1, t, actions=acts) == 0
        else:
            eng.step(random_actions=True)
            assert co.step(1, t, random_actions=True) == 0
        assert_same(eng, co, ctx=f"{variant} blocks={blocks} turn {t}")
    # the other calls that reach the same kernel: sgw_observe, a sweep-only launch, an agent range with SGW_STEP_OBS_NEXT
    eng.obs.zero_()
    eng.observe()
    co.observe()
    assert_same(eng, co, ("obs",), ctx=f"{variant} blocks={blocks} observe")
    a = ws.num_agents // 2
    eng.step(random_actions=True, agent_begin=0, agent_end=a, write_obs=False, obs_next=True, advance_turn=False, turn=T + 1)
    assert co.step(1, T + 1, random_actions=True, write_obs=False, a0=0, a1=a) == 0
    co.observe(a, a + 1)
    torch.cuda.synchronize()
    assert np.array_equal(eng.obs[:, a].cpu().numpy(), co.obs[:, a]), f"{variant} blocks={blocks}: OBS_NEXT window"
    assert_same(eng, co, ("grid", "pos", "total"), ctx=f"{variant} blocks={blocks} agent range")
    # the first launch of a policy-driven turn: sweep only, plus agent 0's window, into a one-window-per-env tensor
    row = torch.zeros((E,) + tuple(ws.obs_shape[1:]), dtype=eng.obs_dtype, device="cuda:0")
    eng.step(sweep=True, agent_begin=0, agent_end=0, obs_next=True, obs_next_out=row, advance_turn=False, turn=T + 2)
    assert co.step(1, T + 2, sweep=True, write_obs=False, a0=0, a1=0) == 0
    co.observe(0, 1)
    torch.cuda.synchronize()
    assert np.array_equal(row.cpu().numpy(), co.obs[:, 0]), f"{variant} blocks={blocks}: sweep-only launch, packed window"
    assert_same(eng, co, ("grid", "pos", "total"), ctx=f"{variant} blocks={blocks} sweep only")
    assert eng.status() == 0


@pytest.mark.parametrize("jit", [1, 0], ids=["specialised", "prebuilt"])
@pytest.mark.parametrize("shape", [(9, 13, 3, 4), (7, 7, 5, 3), (33, 21, 9, 2), (64, 64, 16, 4), (66, 70, 7, 6), (5, 5, 2, 2)])
def test_ragged_shapes_vs_oracle(torch_cuda, shape, jit):
    """Grid byte counts that are not multiples of 16 / 4, odd sizes, maximum radius -- on the instance specialised for each
    shape, and on the library's prebuilt run-time-shape instances."""
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("jit", jit)
    h, w, a, r = shape
    rollout_vs_oracle(treasurehunt_spec(h, w, a, r, spawn_prob=0.08, seed=sum(shape), dense_prob=0.2), 37, 6, first=11, epoch=2, jit=jit)


def _random_world(rng):
    """A random but valid world: shape, layers, agents, radius, spawn/dense rates, values, actions."""
    from sorrel_amd.spec import treasurehunt_spec

    big = rng.random() < 0.25
    h = int(rng.integers(48, 90)) if big else int(rng.integers(5, 41))
    w = int(rng.integers(48, 90)) if big else int(rng.integers(5, 41))
    if rng.random() < 0.5:          # make the byte count a multiple of 16 half of the time (specialised kernels)
        w = max(8, (w // 8) * 8)
        h = max(6, (h // 2) * 2)
    rmax = (min(h, w) - 1) // 2
    r = int(rng.integers(0, min(rmax, 5) + 1))
    a = int(rng.integers(1, min(64, (h - 2) * (w - 2)) + 1))
    a = min(a, 64 if big else 24)
    ws = treasurehunt_spec(h, w, a, r, spawn_prob=float(rng.choice([0.0, 0.003, 0.05, 0.4])), seed=int(rng.integers(0, 2**40)),
                           gem_value=int(rng.integers(1, 20)), food_value=float(rng.choice([5, 0.5, 2.25])),
                           bone_value=-int(rng.integers(1, 20)), dense_prob=float(rng.choice([0.0, 0.1, 0.5])))
    if rng.random() < 0.3:          # a fifth, non-move action name: the agent stays and is paid its own value
        ws.action_dy, ws.action_dx = ws.action_dy + [0], ws.action_dx + [0]
    if rng.random() < 0.3:          # a third, inert layer on top
        ws.layers = 3
        ws.layer_fill_type = ws.layer_fill_type + [0]
        ws.layer_border_type = ws.layer_border_type + [255]
    return ws


@pytest.mark.parametrize("jit", [1, 0], ids=["specialised", "prebuilt"])
@pytest.mark.parametrize("case", range(int(os.environ.get("SGW_SOAK", "48"))))
def test_random_worlds_vs_oracle(torch_cuda, case, jit):
    """Soak: random shapes / agent counts / radii / rates through whichever kernel the dispatcher
    picks (step_fast, step_big, generic) -- the instance specialised for each world, and the prebuilt ones -- a few dozen envs,
    every tensor compared every turn."""
    N.set_option("jit", jit)
    rng = np.random.default_rng(1000 + case)
    ws = _random_world(rng)
    if case % 3 and ws.num_agents <= 16 and ws.layers * ws.height * ws.width <= 4096:     # two thirds of the small cases: packed kernels
        N.set_option("group", 16 if case % 3 == 1 else 32)
    if case % 2:          # half of the cases: worlds that reach step_big take its walking-workgroups instance, 1 / 2 / 5 workgroups
        N.set_option("big_walk_blocks", (1, 2, 5)[case % 3])
    rollout_vs_oracle(ws, int(rng.integers(3, 40)), int(rng.integers(2, 7)), first=int(rng.integers(0, 2**31)),
                      epoch=int(rng.integers(0, 50)), jit=jit)


@pytest.mark.parametrize("jit", [1, 0], ids=["specialised", "prebuilt"])
@pytest.mark.parametrize("case", range(int(os.environ.get("SGW_SOAK", "64"))))
def test_random_rule_worlds_vs_oracle(torch_cuda, case, jit):
    """Soak for the widened rule set (ordered BECOME_IF sweep across layers, timers, several spawners, Cleanup
    beams / facing / all-layer reward) from random maps, on the specialised and on the prebuilt instances: every tensor
    against the C oracle every turn."""
    import torch
    N.set_option("jit", jit)
    if case % 4 == 3:       # a quarter of the cases on the generic kernel (the wave-per-env RULES variant takes the rest)
        N.set_option("fast_rules", 0)
    rng = np.random.default_rng(7000 + case)
    ws, g, pos = H.random_rule_world(rng)
    if ws.agent_rule == 0 and case % 4 == 1:  # plain movers: the policy-driven phases on the byte-gather phase_kernel; the other
        N.set_option("phase_rows", 0)        # cases take the row-load phase kernel wherever an instance exists
        N.set_option("phase_kernel", 1)
    E, T = int(rng.integers(2, 30)), int(rng.integers(3, 12))
    first = int(rng.integers(0, 2**31))
    eng = make_engine(ws, E, first=first)
    assert_instance(eng, jit)
    co = H.COracle(ws, E, first_env_id=first)
    eng.grid.copy_(torch.from_numpy(np.broadcast_to(g, (E,) + g.shape).copy()))
    eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos, (E,) + pos.shape).copy()))
    eng.total_reward.zero_()
    co.grid[...] = g
    co.pos[...] = pos
    co.total[...] = 0
    epoch = int(rng.integers(0, 9))
    eng.epoch = epoch
    phased = case % 3 == 1          # a third of the cases: sweep-only call, then one call per agent (policy path)
    phased_next = case % 6 == 2     # a sixth: the 1 + A launch form (SGW_STEP_OBS_NEXT: each launch renders the NEXT agent)
    for t in range(1, T + 1):
        assert co.step(epoch, t, random_actions=True) == 0
        if phased:
            acts = torch.from_numpy(co.actions.copy())
            eng.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t, advance_turn=False)
            for a in range(ws.num_agents):
                eng.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, turn=t, advance_turn=False)
        elif phased_next:
            acts = torch.from_numpy(co.actions.copy())
            eng.obs.fill_(-5.0)
            eng.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
            for a in range(ws.num_agents):
                eng.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, obs_next=a + 1 < ws.num_agents, write_obs=False, turn=t,
                         advance_turn=False)
        else:
            eng.step(random_actions=True, turn=t, advance_turn=False)
        assert_same(eng, co, ctx=f"case {case} turn {t}{' (phased)' if phased else ' (obs_next)' if phased_next else ''}")
        if eng.agent_dir is not None:
            assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), f"case {case} turn {t}: agent_dir"
        if eng.agent_state is not None:
            assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), f"case {case} turn {t}: agent_state"
            assert np.array_equal(eng.state_at_pov.cpu().numpy(), co.state_at_pov), f"case {case} turn {t}: state_at_pov"
    assert eng.status() == 0


def test_high_spawn_prob_and_certain_spawn(torch_cuda):
    from sorrel_amd.spec import treasurehunt_spec

    rollout_vs_oracle(treasurehunt_spec(12, 12, 4, 2, spawn_prob=1.0, seed=5), 16, 3)
    rollout_vs_oracle(treasurehunt_spec(12, 12, 4, 2, spawn_prob=0.0, seed=5), 16, 3)
    rollout_vs_oracle(treasurehunt_spec(12, 12, 4, 2, spawn_prob=0.5, seed=5), 16, 5)


def test_general_appearance_path_vs_oracle(torch_cuda):
    """Non one-hot entity_map (override_entity_map): float64 layer sum on device."""
    d, spec = H.load_golden("float_appearance_3layer")
    ws = H.world_spec(spec)
    rollout_vs_oracle(ws, 33, 8)


def test_many_channels_hi_nibbles(torch_cuda):
    """More than 8 one-hot channels exercises the second nibble word."""
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(14, 14, 3, 3, spawn_prob=0.1, seed=3)
    ws.num_channels = 13
    app = np.zeros((7, 13))
    for t, ch in enumerate([0, 0, 12, 9, 3, 8, 11]):
        if t >= 2:
            app[t, ch] = 1.0
    ws.appearance = app
    rollout_vs_oracle(ws, 21, 6)


def test_observe_kernel_vs_oracle(torch_cuda):
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(20, 24, 6, 3, spawn_prob=0.1, seed=4, dense_prob=0.3)
    eng, co = rollout_vs_oracle(ws, 100, 3)
    eng.obs.fill_(-1.0)
    co.obs.fill(-1.0)
    eng.observe()
    co.observe()
    assert_same(eng, co, ("obs", "grid", "pos"), ctx="observe all")
    eng.obs.fill_(-1.0)
    co.obs.fill(-1.0)
    eng.observe(2, 4)
    co.observe(2, 4)
    assert_same(eng, co, ("obs",), ctx="observe [2,4)")


@pytest.mark.parametrize("shape", [(16, 16, 4, 2, 200), (32, 32, 8, 3, 70), (21, 21, 3, 2, 33), (13, 9, 5, 4, 50), (64, 64, 9, 5, 5)])
@pytest.mark.parametrize("phase_kernel", ["1", "0", "rows"])
def test_policy_phase_stepping_equals_fused(torch_cuda, shape, phase_kernel, monkeypatch):
    """sweep once, then per agent sgw_observe + sgw_step (round 1's 1 + 2A launches) == one fused take_turn; and the plain
    per-agent step that writes the mover's own pre-move observation.  On the row-load phase kernel (the default where an
    instance exists), the byte-gather phase kernel and the staging kernels."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    if phase_kernel == "rows":
        N.set_option("phase_rows", 1)
    else:
        N.set_option("phase_rows", 0)
        N.set_option("phase_kernel", 1 if phase_kernel == "1" else 0)
    h, w, a_, r_, E = shape
    ws = treasurehunt_spec(h, w, a_, r_, spawn_prob=0.05, seed=6, dense_prob=0.1)
    fused, phased = make_engine(ws, E), make_engine(ws, E)
    fused.reset(0)
    phased.reset(0)
    for t in range(1, 8):
        acts = fused.random_actions(turn=t).clone()
        fused.step(acts)
        phased.turn += 1
        phased.step(acts, sweep=True, agent_begin=0, agent_end=0, turn=t)      # sweep only
        rew = torch.zeros_like(phased.rewards)
        for a in range(ws.num_agents):
            phased.observe(a, a + 1)                                            # pov of agent a
            phased.step(acts, sweep=False, write_obs=False, agent_begin=a, agent_end=a + 1, turn=t)
            rew[:, a] = phased.rewards[:, a]
        torch.cuda.synchronize()
        assert torch.equal(fused.grid, phased.grid)
        assert torch.equal(fused.obs, phased.obs)
        assert torch.equal(fused.rewards, rew)
        assert torch.equal(fused.total_reward, phased.total_reward)
        assert torch.equal(fused.agent_pos, phased.agent_pos)
    # the plain per-agent step: one call that writes the mover's own (pre-move) observation and moves it
    third = make_engine(ws, E)
    third.reset(0)
    fused.reset(0)
    for t in range(1, 5):
        acts = fused.random_actions(turn=t).clone()
        fused.step(acts, turn=t)
        third.obs.fill_(-3.0)
        third.step(acts, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t)
        for a in range(ws.num_agents):
            third.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, turn=t)
        torch.cuda.synchronize()
        assert torch.equal(fused.obs, third.obs) and torch.equal(fused.grid, third.grid)
        assert torch.equal(fused.total_reward, third.total_reward) and torch.equal(fused.agent_pos, third.agent_pos)
    assert fused.status() == 0 and phased.status() == 0 and third.status() == 0


def test_resharding_is_bit_exact(torch_cuda):
    """Env results depend on the GLOBAL env id only: two half shards == one full batch."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.01, seed=7)
    E = 512
    full = make_engine(ws, E, first=1000)
    lo, hi = make_engine(ws, E // 2, first=1000), make_engine(ws, E // 2, first=1000 + E // 2)
    for e in (full, lo, hi):
        e.reset(1)
    for _ in range(5):
        for e in (full, lo, hi):
            e.step(random_actions=True)
    torch.cuda.synchronize()
    for name in ("grid", "agent_pos", "obs", "rewards", "total_reward", "actions"):
        whole = getattr(full, name)
        parts = torch.cat([getattr(lo, name), getattr(hi, name)], dim=0)
        assert torch.equal(whole, parts), name


def test_metric_reduction_vs_oracle(torch_cuda):
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.05, seed=8)
    eng, co = rollout_vs_oracle(ws, 3000, 10, check_every=10)
    m = eng.reduce_metrics().cpu().numpy()
    ref = co.metrics()
    assert m[0] == ref[0] and m[1] == ref[1] and m[2] == 3000.0   # integer-valued sums: exact in any order
    m2 = eng.reduce_metrics().cpu().numpy()
    assert np.array_equal(m, m2)                                  # fixed order: bitwise reproducible


def test_status_flags_bad_action_and_unwalled_border(torch_cuda):
    torch = torch_cuda
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(8, 8, 2, 2, spawn_prob=0.0, seed=9)
    eng = make_engine(ws, 4)
    eng.reset(0)
    eng.step(torch.full((4, 2), 9, dtype=torch.uint8, device="cuda"))
    assert eng.status() & N.STATUS_BAD_ACTION
    # knock a hole in the wall above agent 0 of env 0 and walk it off the grid
    eng.reset(0)
    y, x = [int(v) for v in eng.agent_pos[0, 0].cpu()]
    eng.grid[0, 1, :, x] = 1
    for _ in range(8):
        eng.step(torch.zeros((4, 2), dtype=torch.uint8, device="cuda"))   # "up"
    with pytest.raises(IndexError):
        eng.raise_on_status()


@pytest.mark.parametrize("shape", [(32, 32, 8, 3, 64), (12, 9, 3, 2, 7), (128, 128, 64, 5, 3)])
def test_garbage_positions_are_flagged_and_memory_safe(torch_cuda, shape, monkeypatch):
    """Positions outside the grid (an uninitialised agent_pos tensor): every kernel treats them as (0, 0), stays
    inside its env's LDS slice, raises SGW_STATUS_BAD_POS, and leaves the other envs bit-exact."""
    torch = torch_cuda
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec

    h, w, a, r, E = shape
    ws = treasurehunt_spec(h, w, a, r, spawn_prob=0.01, seed=4)
    for force_generic in (False, True):
        if force_generic:
            N.set_option("force_generic", 1)
        eng = make_engine(ws, E)
        co = H.COracle(ws, E)
        eng.reset(0)
        co.reset(0)
        bad_env = E // 2
        eng.agent_pos[bad_env] = 250                      # (250, 250): far outside
        eng.step(random_actions=True)
        co.step(0, 1, random_actions=True)
        torch.cuda.synchronize()
        assert eng.status() & N.STATUS_BAD_POS
        keep = np.arange(E) != bad_env
        assert np.array_equal(eng.grid.cpu().numpy()[keep], co.grid[keep])
        assert np.array_equal(eng.obs.cpu().numpy()[keep], co.obs[keep])
        assert np.array_equal(eng.agent_pos.cpu().numpy()[keep], co.pos[keep])
        eng.agent_pos[bad_env, 0, 1] = 255                # (the step wrote clamped positions back)
        eng.observe()                                     # K1 flags it too
        assert eng.status() & N.STATUS_BAD_POS
        with pytest.raises(IndexError):
            eng.agent_pos[bad_env] = 251
            eng.step(random_actions=True)
            eng.raise_on_status()


def test_create_rejects_invalid_configs(torch_cuda):
    from sorrel_amd.spec import treasurehunt_spec

    with pytest.raises(ValueError):
        make_engine(treasurehunt_spec(10, 10, 2, 5), 4)          # r > (min-1)//2
    with pytest.raises(ValueError):
        make_engine(treasurehunt_spec(4, 4, 5, 1), 4)            # more agents than interior cells
    ws = treasurehunt_spec(10, 10, 2, 2)
    ws.type_rule[3] = 7
    with pytest.raises(ValueError):
        make_engine(ws, 4)                                       # unsupported transition rule


# ------------------------------------------------------------------ (3) full BASELINE sizes: properties
def check_properties(torch, eng, ws, T):
    """Size-independent invariants of take_turn."""
    E, A = eng.num_envs, ws.num_agents
    agent_t, zA = ws.agent_type[0], ws.agent_layer
    cum = torch.zeros(E, dtype=torch.float64, device="cuda")
    for t in range(T):
        eng.step(random_actions=True)
        cum += eng.rewards.double().sum(dim=1)
    torch.cuda.synchronize()
    # total_reward is the running sum of the per-agent rewards (integers here: exact)
    assert torch.equal(cum, eng.total_reward)
    # exactly A agent cells per env, at the recorded positions, on the agent layer
    g = eng.grid
    assert torch.equal((g == agent_t).sum(dim=(1, 2, 3)), torch.full((E,), A, device="cuda"))
    pos = eng.agent_pos.long()
    e_idx = torch.arange(E, device="cuda")[:, None].expand(E, A)
    assert bool((g[e_idx, zA, pos[..., 0], pos[..., 1]] == agent_t).all())
    # walls and the inert layer never change
    assert bool((g[:, zA, 0, :] == 2).all() and (g[:, zA, -1, :] == 2).all()
                and (g[:, zA, :, 0] == 2).all() and (g[:, zA, :, -1] == 2).all())
    assert bool((g[:, 0] == 0).all())
    # observations: every value is a small count; channel 0 (EmptyEntity) is all-zero;
    # stateless re-observation of the final state shows each agent at its window centre
    assert bool((eng.obs[:, :, 0] == 0).all())
    assert float(eng.obs.max()) <= ws.layers
    eng.observe()
    r = ws.vision_radius
    assert bool((eng.obs[:, :, 5, r, r] == 1.0).all())
    assert eng.status() == 0


def test_config3_full_size_properties(torch_cuda):
    """BASELINE headline: 32x32, 8 agents, 7x7 window, 65 536 envs."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=0)
    eng = make_engine(ws, 65536)
    eng.reset(0)
    check_properties(torch, eng, ws, 20)
    # determinism: an identical second run gives identical tensors
    eng2 = make_engine(ws, 65536, allocate_obs=False)
    eng2.reset(0)
    for _ in range(20):
        eng2.step(random_actions=True, write_obs=False)
    torch.cuda.synchronize()
    assert torch.equal(eng.grid, eng2.grid) and torch.equal(eng.total_reward, eng2.total_reward)
    # a strided sample of envs against the oracle, env by env (global ids)
    for env_id in (0, 1, 777, 32768, 65535):
        co = H.COracle(ws, 1, first_env_id=env_id, threads=1)
        co.reset(0)
        for t in range(1, 21):
            co.step(0, t, random_actions=True)
        assert np.array_equal(eng.grid[env_id].cpu().numpy(), co.grid[0])
        assert eng.total_reward[env_id].item() == co.total[0]


def test_config5_per_gpu_size_properties(torch_cuda):
    """Config 5 per-GPU share: 128x128, 64 agents, 11x11 window, dense, 2048 envs."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    ws = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=0, dense_prob=0.25)
    eng = make_engine(ws, 2048)
    eng.reset(0)
    check_properties(torch, eng, ws, 3)


@pytest.mark.parametrize("which", ["config3_rank0", "config4_last_rank", "config5_rank0", "config5_last_rank"])
def test_full_size_batches_full_tensor_vs_oracle(torch_cuda, which):
    """BASELINE configs 3 / 4 / 5 at their FULL per-GPU sizes, every element of every tensor against the C oracle (its
    OpenMP rollout over all host cores does a 65 536-env turn in well under a second): observations (617 MB per turn at
    config 3), actions, rewards, grid, positions, total_reward.  "config4_last_rank" / "config5_last_rank" are the shards
    the eighth GPU of configs 4 / 5 owns (global env ids 458 752 ... and 14 336 ...): the RNG is keyed by the GLOBAL env
    id, so this is what the 8-rank run computes there."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    if which.startswith("config5"):
        ws, E, T = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=0, dense_prob=0.25), 2048, 3
    else:
        ws, E, T = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=0), 65536, 3
    first = 7 * E if which.endswith("last_rank") else 0
    eng = make_engine(ws, E, first=first)
    co = H.COracle(ws, E, first_env_id=first, threads=0)
    eng.reset(0)
    co.reset(0)
    assert_same(eng, co, ("grid", "pos", "total"), ctx=f"{which} reset")
    for t in range(1, T + 1):
        eng.step(random_actions=True)
        assert co.step(0, t, random_actions=True) == 0
        assert_same(eng, co, ctx=f"{which} turn {t}")
    # and the same turns once more through sgw_rollout on a second engine: identical final state and last-turn tensors
    ro = make_engine(ws, E, first=first)
    ro.reset(0)
    ro.rollout(T)
    assert_same(ro, co, ctx=f"{which} rollout")
    assert eng.status() == 0 and ro.status() == 0


@pytest.mark.parametrize("group", ["16", "32"])
def test_packed_kernels_edge_shapes(torch_cuda, group, monkeypatch):
    """Packed kernels at their limits: as many agents as lanes per env, the widest window a small world allows (many
    render passes per agent), 13 channels (second counter word), five layers, a world of exactly 4 096 bytes, batches of
    one env and of 64 / G + 1 envs (one group of the last wave), general float appearance."""
    from sorrel_amd.spec import treasurehunt_spec

    N.set_option("group", int(group))
    G = int(group)
    rollout_vs_oracle(treasurehunt_spec(12, 12, G, 2, spawn_prob=0.1, seed=1, dense_prob=0.2), 37, 5)          # A == G
    rollout_vs_oracle(treasurehunt_spec(27, 23, 3, 11, spawn_prob=0.05, seed=2), 19, 4)                          # 23x23 window: 529 cells
    rollout_vs_oracle(treasurehunt_spec(64, 32, 5, 4, spawn_prob=0.02, seed=3), 9, 4)                            # 4 096 bytes
    rollout_vs_oracle(treasurehunt_spec(9, 9, 2, 3, spawn_prob=0.2, seed=4), 1, 6)                               # one env
    rollout_vs_oracle(treasurehunt_spec(9, 9, 2, 3, spawn_prob=0.2, seed=4), 64 // G + 1, 6, first=123456789)
    ws = treasurehunt_spec(14, 14, 3, 3, spawn_prob=0.1, seed=5)
    ws.num_channels = 13
    app = np.zeros((7, 13))
    for t, ch in enumerate([0, 0, 12, 9, 3, 8, 11]):
        if t >= 2:
            app[t, ch] = 1.0
    ws.appearance = app
    rollout_vs_oracle(ws, 21, 5)
    ws = treasurehunt_spec(10, 11, 4, 2, spawn_prob=0.1, seed=6)
    ws.layers = 5
    ws.layer_fill_type = ws.layer_fill_type + [0, 0, 0]
    ws.layer_border_type = ws.layer_border_type + [255, 2, 255]
    rollout_vs_oracle(ws, 30, 5)
    ws = treasurehunt_spec(11, 13, 3, 2, spawn_prob=0.1, seed=7)
    ws.appearance = np.asarray(ws.appearance) * np.array([[0.5], [1.0], [2.0], [1.5], [3.0], [0.25], [1.0]])     # not one-hot: float64 path
    rollout_vs_oracle(ws, 25, 5)

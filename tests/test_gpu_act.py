"""sgw_act: one agent acts, the later agents' windows are repaired; action values and exploration in the act launch; MovingAgent.act / TagAgent.act / CleanupAgent.act, sorrel/agents/agent.py:215-225, examples/tag/agents.py:76-106, examples/cleanup/agents.py:93-177.
(Round 6: regrouped by component from the by-round files of rounds 2-5; no test body changed.)"""
import json  # noqa: F401
import os  # noqa: F401
import subprocess  # noqa: F401
import sys  # noqa: F401

import numpy as np  # noqa: F401
import pytest

from oracle import gridstep_oracle as O  # noqa: F401
from sorrel_amd import _native as N  # noqa: F401
from tests import helpers as H  # noqa: F401
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


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

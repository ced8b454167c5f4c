"""Policy-driven PHASES on the step kernels themselves: one agent acts per launch and the next agent's window is rendered (SGW_STEP_OBS_NEXT), the row-load phase kernel (phase_rows); Agent.transition, sorrel/agents/agent.py:155-173.
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


@pytest.mark.parametrize("phase_kernel", ["phase_rows", "phase_kernel", "staging_kernels"])
@pytest.mark.parametrize("case", KERNEL_CASES, ids=[c[0] for c in KERNEL_CASES])
def test_obs_next_phased_turn_equals_fused(torch_cuda, case, phase_kernel, monkeypatch):
    """Sweep + obs of agent 0 in one launch, then ONE launch per agent that moves it and renders the next agent's
    observation: every observation, reward and the final state equal the fused take_turn (which equals the oracle),
    in 1 + A launches."""
    torch = torch_cuda
    _, mk, env, E = case
    for k, v in env.items():
        N.set_option(k, v)
    if phase_kernel != "phase_rows":           # (round 3's row-load phase kernel is the default where an instance exists)
        N.set_option("phase_rows", 0)
    if phase_kernel == "staging_kernels":      # the phases on the step kernels themselves (what Tag / Cleanup phases always use)
        N.set_option("phase_kernel", 0)
    ws = mk()
    A = ws.num_agents
    fused, phased = make_engine(ws, E, first=5), make_engine(ws, E, first=5)
    co = H.COracle(ws, E, first_env_id=5)
    for e in (fused, phased):
        e.reset(0)
    co.reset(0)
    if fused.agent_state is not None:
        co.agent_state[...] = fused.agent_state.cpu().numpy()
    for t in range(1, 6):
        acts = fused.random_actions(turn=t).clone()
        fused.step(acts, turn=t)
        co.step(0, t, actions=acts.cpu().numpy())
        phased.set_timing(True)
        seen = torch.zeros_like(fused.obs)
        rew = torch.zeros_like(phased.rewards)
        phased.obs.fill_(-7.0)
        phased.step(acts, sweep=True, agent_begin=0, agent_end=0, turn=t, obs_next=True)     # sweep + pov of agent 0
        for a in range(A):
            seen[:, a] = phased.obs[:, a]                                                      # what agent a's policy would read
            phased.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, turn=t, obs_next=a + 1 < A, write_obs=False)
            rew[:, a] = phased.rewards[:, a]
        ms, launches = phased.step_time_ms()
        phased.set_timing(False)
        assert launches == 1 + A
        torch.cuda.synchronize()
        assert torch.equal(seen, fused.obs), f"turn {t}: phased observations differ"
        assert np.array_equal(fused.obs.cpu().numpy(), co.obs), f"turn {t}: fused observations differ from the oracle"
        assert torch.equal(fused.grid, phased.grid) and torch.equal(fused.agent_pos, phased.agent_pos)
        assert torch.equal(fused.rewards, rew) and torch.equal(fused.total_reward, phased.total_reward)
        if fused.agent_state is not None:
            assert torch.equal(fused.agent_state, phased.agent_state)
    assert fused.status() == 0 and phased.status() == 0


def test_obs_next_on_the_rules_kernel(torch_cuda):
    """Cleanup (the RULES variant of the wave-per-env kernel, facing + beams): phased with OBS_NEXT == fused."""
    torch = torch_cuda
    ws, d = _cleanup_spec()
    E, A = 6, ws.num_agents
    fused, phased = make_engine(ws, E), make_engine(ws, E)
    g0 = torch.from_numpy(np.broadcast_to(d["grid0"][0], (E,) + d["grid0"][0].shape).copy())
    p0 = torch.from_numpy(np.broadcast_to(d["pos0"][0], (E,) + d["pos0"][0].shape).copy())
    for e in (fused, phased):
        e.grid.copy_(g0)
        e.agent_pos.copy_(p0)
        e.total_reward.zero_()
    for t in range(1, 9):
        acts = fused.random_actions(turn=t).clone()
        fused.step(acts, turn=t)
        seen = torch.zeros_like(fused.obs)
        phased.step(acts, sweep=True, agent_begin=0, agent_end=0, turn=t, obs_next=True)
        for a in range(A):
            seen[:, a] = phased.obs[:, a]
            phased.step(acts, sweep=False, agent_begin=a, agent_end=a + 1, turn=t, obs_next=a + 1 < A, write_obs=False)
        torch.cuda.synchronize()
        assert torch.equal(seen, fused.obs), t
        assert torch.equal(fused.grid, phased.grid) and torch.equal(fused.agent_dir, phased.agent_dir)
        assert torch.equal(fused.total_reward, phased.total_reward)


# ------------------------------------------------------------------ SGW_STEP_OBS_NEXT_PACKED: the next agent's window, one per env
@pytest.mark.parametrize("case", ["fast_32x32", "rows_32x32", "packed_21x21", "rows_128", "phase_kernel_128", "step_big_128", "generic_256_128",
                                  "rules_cleanup", "u8"])
def test_obs_next_packed_destination_equals_the_tensor_slot(torch_cuda, case, monkeypatch):
    """``obs_next_out`` (one window per env, e.g. a replay row) receives exactly what slot ``agent_end`` of the observation
    tensor receives without it, on every kernel family that serves policy-driven phases; nothing else is written."""
    torch = torch_cuda
    from sorrel_amd.spec import treasurehunt_spec

    kw = {}
    if not case.startswith("rows_"):           # the older phase paths; rows_*: the row-load phase kernel (the default)
        N.set_option("phase_rows", 0)
    if case in ("fast_32x32", "rows_32x32"):
        ws, E = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.05, seed=3, dense_prob=0.2), 70
    elif case == "packed_21x21":
        N.set_option("group", 16)
        ws, E = treasurehunt_spec(21, 21, 3, 2, spawn_prob=0.05, seed=4, dense_prob=0.2), 203
    elif case == "rules_cleanup":
        d, spec = H.load_golden("cleanup_15x16")
        ws, E = H.world_spec(spec), 33
    elif case == "u8":
        ws, E = treasurehunt_spec(20, 24, 5, 2, spawn_prob=0.05, seed=6, dense_prob=0.2), 41
        kw["obs_dtype"] = torch.uint8
    else:
        if case == "step_big_128":
            N.set_option("phase_kernel", 0)
        if case == "generic_256_128":
            N.set_option("force_generic", 1)
        ws, E = treasurehunt_spec(128, 128, 24, 5, spawn_prob=0.05, seed=5, dense_prob=0.25), 9
    A = ws.num_agents
    a, b = make_engine(ws, E, first=5, **kw), make_engine(ws, E, first=5, **kw)
    if case == "rules_cleanup":
        for e in (a, b):
            e.grid.copy_(torch.from_numpy(np.broadcast_to(d["grid0"][0], (E,) + d["grid0"][0].shape).copy()))
            e.agent_pos.copy_(torch.from_numpy(np.broadcast_to(d["pos0"][0], (E,) + d["pos0"][0].shape).copy()))
            e.total_reward.zero_()
    else:
        a.reset(0)
        b.reset(0)
    per_env = int(np.prod(ws.obs_shape[1:]))
    rng = np.random.default_rng(2)
    for t in range(1, 4):
        acts = torch.from_numpy(rng.integers(0, len(ws.action_dy), size=(E, A), dtype=np.uint8)).cuda()
        rows = [torch.full((E, per_env), 7, dtype=a.obs_dtype, device="cuda:0") for _ in range(A)]
        a.obs.fill_(9)
        b.obs.fill_(9)
        a.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, turn=t, advance_turn=False)
        b.step(acts, sweep=True, agent_begin=0, agent_end=0, obs_next=True, obs_next_out=rows[0], turn=t, advance_turn=False)
        for i in range(A):
            nxt = i + 1 < A
            a.step(acts, sweep=False, write_obs=False, agent_begin=i, agent_end=i + 1, obs_next=nxt, turn=t, advance_turn=False)
            b.step(acts, sweep=False, write_obs=False, agent_begin=i, agent_end=i + 1, obs_next=nxt,
                   obs_next_out=rows[i + 1] if nxt else None, turn=t, advance_turn=False)
        torch.cuda.synchronize()
        for i in range(A):
            assert torch.equal(rows[i].view(E, *ws.obs_shape[1:]), a.obs[:, i]), f"{case} turn {t}: window of agent {i}"
        assert bool((b.obs == 9).all()), f"{case}: the packed calls must not touch the observation tensor"
        assert torch.equal(a.grid, b.grid) and torch.equal(a.agent_pos, b.agent_pos) and torch.equal(a.total_reward, b.total_reward)
    assert a.status() == 0 and b.status() == 0
    with pytest.raises(ValueError):
        b.step(acts, obs_next=False, obs_next_out=rows[0])
    with pytest.raises(ValueError):
        b.step(acts, agent_begin=0, agent_end=1, obs_next=True, obs_next_out=torch.zeros((E, per_env + 1), dtype=a.obs_dtype, device="cuda:0"))


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

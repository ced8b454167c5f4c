"""More than 64 agents (round 6; SGW_MAX_AGENTS = 128).  The reference steps any ``self.agents`` list (``sorrel/environment.py:92-93``); the engine
keeps an agent per lane on its wave- and workgroup-per-env kernels, so 65..128 agents run on the ticket-ordered generic kernel (``step_kernel<256>``),
whatever the world's size.  Pinned by two fixtures the reference's own ``take_turn`` produced (80 Treasurehunt agents, 70 Tag agents:
``oracle/make_golden.py round6``; they also run in ``test_gpu_parity.py::test_hip_matches_reference_golden``), and checked against the C oracle
at batch sizes -- reset (the wave-parallel placement's second register set), fused and phased turns, ``sgw_observe_rows`` + ``sgw_act``, rollouts."""
import numpy as np
import pytest

from sorrel_amd import _native as N
from tests import helpers as H
from tests.test_gpu_parity import assert_same, make_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no silent CPU fallback)")
    return torch


def _th(h, w, a, r, **kw):
    from sorrel_amd.spec import treasurehunt_spec

    return treasurehunt_spec(h, w, a, r, spawn_prob=0.04, seed=31, dense_prob=0.2, **kw)


def _tag(h, w, a, r):
    from tests.gpu_common import _tag_spec

    return _tag_spec(h, w, a, r)


CASES = [
    ("th_128x128_A128_r5", lambda: _th(128, 128, 128, 5), 9),          # the review's target: 128 agents on config 5's map
    ("th_128x128_A65_r5", lambda: _th(128, 128, 65, 5), 6),            # one more than a wave holds
    ("th_24x26_A80_r2", lambda: _th(24, 26, 80, 2), 70),               # a small, crowded world (1.2 KB per env) on the workgroup-per-env kernel
    ("th_14x14_A100_r1", lambda: _th(14, 14, 100, 1), 33),             # 100 agents on 144 interior cells: nearly every move is contested
    ("tag_40x42_A96_r4", lambda: _tag(40, 42, 96, 4), 12),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_more_than_64_agents_vs_the_c_oracle(torch_cuda, case):
    torch = torch_cuda
    name, mk, E = case
    ws = mk()
    A = ws.num_agents
    eng, co = make_engine(ws, E, first=11), H.COracle(ws, E, first_env_id=11)
    assert "step_kernel<256" in eng.launch_info(), eng.launch_info()
    assert not (eng.capabilities() & N.CAP_RESOLVE)                    # (the speculative resolve keeps an agent per lane)
    eng.reset(epoch=2)
    co.reset(2)
    if eng.agent_state is not None:
        co.agent_state[...] = eng.agent_state.cpu().numpy()
    torch.cuda.synchronize()
    assert np.array_equal(eng.grid.cpu().numpy(), co.grid) and np.array_equal(eng.agent_pos.cpu().numpy(), co.pos), f"{name}: reset"
    pos = co.pos.astype(np.int64)
    assert all(len({(int(y), int(x)) for y, x in pos[e]}) == A for e in range(E)), "agents on distinct cells"
    for t in range(1, 5):                                               # fused turns, actions drawn on the device
        eng.step(random_actions=True, turn=t)
        assert co.step(2, t, random_actions=True) == 0
        assert_same(eng, co, ctx=f"{name} fused turn {t}")
    gen = np.random.default_rng(3)
    nact = len(ws.action_dy)
    for t in range(5, 8):                                               # given actions, agent after agent (1 + A launches), windows at pov time
        acts = gen.integers(0, nact, (E, A)).astype(np.uint8)
        ta = torch.from_numpy(acts).cuda()
        assert co.step(2, t, actions=acts) == 0
        seen = torch.zeros_like(eng.obs)
        eng.step(ta, sweep=True, agent_begin=0, agent_end=0, turn=t, obs_next=True)
        for a in range(A):
            seen[:, a] = eng.obs[:, a]
            eng.step(ta, sweep=False, agent_begin=a, agent_end=a + 1, turn=t, obs_next=a + 1 < A, write_obs=False)
        torch.cuda.synchronize()
        assert np.array_equal(seen.cpu().numpy(), co.obs), f"{name} phased turn {t}: windows"
        assert_same(eng, co, what=("grid", "pos", "rewards", "total"), ctx=f"{name} phased turn {t}")
    if eng.capabilities() & N.CAP_OBSERVE_ROWS:                         # windows once + sgw_act per agent (act_patch<64, 2>: two agents per lane)
        Nw = int(np.prod(ws.obs_shape[1:]))
        dests = [torch.full((E, Nw), -5.0, device="cuda:0") for _ in range(A)]
        rows = eng.window_rows(dests)
        for t in range(8, 11):
            acts = gen.integers(0, nact, (E, A)).astype(np.uint8)
            ta = torch.from_numpy(acts).cuda()
            assert co.step(2, t, actions=acts) == 0
            eng.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t)
            eng.observe_rows(rows)
            seen = []
            for a in range(A):
                seen.append(dests[a].clone())
                eng.act(a, rows, action=ta[:, a].to(torch.int64).contiguous())
            torch.cuda.synchronize()
            got = torch.stack(seen, dim=1).view(E, A, *ws.obs_shape[1:]).cpu().numpy()
            assert np.array_equal(got, co.obs), f"{name} sgw_act turn {t}: windows at pov time"
            assert_same(eng, co, what=("grid", "pos", "rewards", "total"), ctx=f"{name} sgw_act turn {t}")
    eng.rollout(4)                                                      # sgw_rollout continues from here
    for t in range(eng.turn - 3, eng.turn + 1):
        assert co.step(2, t, random_actions=True) == 0
    assert_same(eng, co, ctx=f"{name} rollout")
    assert eng.status() == 0


def test_cleanup_rules_with_more_than_64_agents_vs_the_c_oracle(torch_cuda):
    """The layered rule set (BECOME_IF sweeps, beams on the layer above, facing, all-layer rewards: examples/cleanup) with 72 agents on a 30x34x3
    map -- the generic kernel's Cleanup instance with 128-entry per-agent arrays --, fused turns and sgw_act turns against the C oracle."""
    torch = torch_cuda
    from tests.gpu_common import _cleanup_spec

    ws, _d = _cleanup_spec()
    A = 72
    ws.height, ws.width, ws.num_agents, ws.agent_type = 30, 34, A, [ws.agent_type[0]] * A
    E = 10
    eng, co = make_engine(ws, E, first=2), H.COracle(ws, E, first_env_id=2)
    assert "step_kernel<256" in eng.launch_info() and "128" in eng.launch_info().split(" group")[0], eng.launch_info()
    eng.reset(epoch=1)
    co.reset(1)
    co.agent_dir[...] = eng.agent_dir.cpu().numpy()
    nact = len(ws.action_dy)
    for t in range(1, 6):
        eng.step(random_actions=True, turn=t)
        assert co.step(1, t, random_actions=True) == 0
        assert_same(eng, co, ctx=f"cleanup 72 agents, fused turn {t}")
        assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), t
    gen = np.random.default_rng(8)
    rows = eng.window_rows(None)
    for t in range(6, 9):                                               # windows once (NO_MOVE) + sgw_act per agent: beams repaired in later windows
        acts = gen.integers(0, nact, (E, A)).astype(np.uint8)
        ta = torch.from_numpy(acts).cuda()
        assert co.step(1, t, actions=acts) == 0
        eng.step(ta, sweep=True, no_move=True, turn=t)
        seen = torch.zeros_like(eng.obs)
        for a in range(A):
            seen[:, a] = eng.obs[:, a]
            eng.act(a, rows, action=ta[:, a].to(torch.int64).contiguous())
        torch.cuda.synchronize()
        assert np.array_equal(seen.cpu().numpy(), co.obs), f"sgw_act turn {t}: windows at pov time"
        assert_same(eng, co, what=("grid", "pos", "rewards", "total"), ctx=f"cleanup 72 agents, sgw_act turn {t}")
        assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), t
    assert eng.status() == 0


def test_agent_limit_is_128_and_said_so(torch_cuda):
    from sorrel_amd.spec import treasurehunt_spec

    assert N.MAX_AGENTS == 128
    with pytest.raises(ValueError):
        treasurehunt_spec(64, 64, 129, 2).to_config(4, 0)
    ok = make_engine(treasurehunt_spec(64, 64, 128, 2), 3)
    ok.reset(0)
    ok.step(random_actions=True)
    assert ok.status() == 0


def test_environment_with_96_policy_driven_agents(torch_cuda):
    """The Python API: 96 agents with a shared linear policy and a shared replay ring through Environment.take_turn (the eager loops), against
    the C oracle stepping the actions taken."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

    E, A = 17, 96
    one = []

    class Shared(BaseModel):
        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=0, num_envs=E, device="cuda:0")
            self.memory = Buffer(capacity=2 * A, obs_shape=tuple(input_size), num_envs=E, device="cuda:0")
            self.w = torch.randn(int(np.prod(input_size)), action_space, generator=torch.Generator().manual_seed(4)).cuda()

        def take_action(self, state):
            return (state.reshape(state.shape[0], -1) @ self.w).argmax(dim=1)

    def factory(input_size, action_space):
        if not one:
            one.append(Shared(input_size, action_space))
        return one[0]

    env = make_env(30, 31, A, 2, E, p=0.05, seed=9, model_factory=factory)
    env.speculate_turns = "always"                                     # (the resolve kernel keeps an agent per lane: the generic speculative turn plays, sgw_verify_rows)
    co = H.COracle(env.compile_spec(), E)
    co.grid[...] = env.world.grid.cpu().numpy()
    co.pos[...] = env.world.agent_pos.cpu().numpy()
    for t in range(1, 5):
        env.take_turn()
        torch.cuda.synchronize()
        assert env.turn_plan()["loop"] == ("speculative" if t < 4 else "fast"), env.turn_plan()
        if t == 3:
            assert env._spec_generic is True
            env.speculate_turns = False                                # ... and the eager loop for the last turn
        assert co.step(0, t, actions=env.actions.cpu().numpy()) == 0
        assert np.array_equal(env.world.grid.cpu().numpy(), co.grid) and np.array_equal(env.rewards.cpu().numpy(), co.rewards), t
        assert np.array_equal(env.world.total_reward.cpu().numpy(), co.total), t
        mem = one[0].memory
        rows = [(mem.idx - A + a) % mem.capacity for a in range(A)]
        got = torch.stack([mem.states[r] for r in rows], dim=1).cpu().numpy().reshape(co.obs.shape)
        assert np.array_equal(got, co.obs), (t, "the windows the agents acted on")
    env.raise_on_status()


@pytest.mark.parametrize("layout", ["rows", "tensor"])
def test_recorded_turn_with_72_agents_equals_the_eager_loop(torch_cuda, layout):
    """``Environment.capture_turn()`` beyond 64 agents (the device-side turn state holds 128 rings): 72 agents with a linear policy and a replay ring
    each, one graph replay per turn against the eager loop and the C oracle -- state, step outputs, every ring, across a reset."""
    torch = torch_cuda
    from tests.gpu_common import _policy_env

    E, shape = 11, (24, 25, 72, 2)
    a, b = _policy_env(E, shape=shape, memory=4), _policy_env(E, shape=shape, memory=4)
    b.capture_layout = layout
    cap = b.capture_turn(warmup=2, force=True)
    assert cap is not None, getattr(b, "capture_error", None)
    for _ in range(2):
        a.take_turn()
    co = H.COracle(a._engine.spec, E, first_env_id=0)

    def sync_oracle():
        co.grid[...] = a.world.grid.cpu().numpy()
        co.pos[...] = a.world.agent_pos.cpu().numpy()
        co.total[...] = a.world.total_reward.cpu().numpy()

    sync_oracle()
    for t in range(9):
        if t == 5:
            a.reset()
            b.reset()
            sync_oracle()
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        assert b.turn_plan()["loop"] == "recorded"
        assert co.step(a.epoch, a.turn, actions=a.actions.cpu().numpy()) == 0
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert torch.equal(a.rewards, b.rewards) and torch.equal(a.actions, b.actions), t
        assert np.array_equal(b.world.grid.cpu().numpy(), co.grid) and np.array_equal(b.rewards.cpu().numpy(), co.rewards), t
        for k, (x, y) in enumerate(zip(a.agents, b.agents)):
            mx, my = x.model.memory, y.model.memory
            assert (mx.idx, mx.size) == (my.idx, my.size), (t, k)
            for name in ("states", "actions", "rewards", "dones"):
                assert torch.equal(getattr(mx, name), getattr(my, name)), (t, k, name)
    a.raise_on_status()
    b.raise_on_status()


@pytest.mark.parametrize("case", range(max(6, int(__import__("os").environ.get("SGW_SOAK", "0")) // 8)))
def test_many_agents_soak_random_worlds(torch_cuda, case):
    """Random worlds with 65 ... 128 agents -- Treasurehunt, Tag or Cleanup rules, map, radius, batch and global env ids drawn per case -- on the generic
    workgroup-per-env kernel: reset, fused turns with device-drawn actions, a turn of sgw_act launches (windows at pov time), against the C oracle."""
    torch = torch_cuda
    from tests.gpu_common import _cleanup_spec

    rng = np.random.default_rng(9000 + case)
    kind = ("move", "tag", "cleanup")[case % 3]
    A = int(rng.integers(65, 129))
    side = int(np.ceil(np.sqrt(A * rng.uniform(1.3, 6.0)))) + 2
    h, w = side + int(rng.integers(0, 9)), side + int(rng.integers(0, 9))
    r = int(rng.integers(1, min(5, (min(h, w) - 1) // 2) + 1))
    E, first = int(rng.integers(3, 14)), int(rng.integers(0, 5000))
    if kind == "move":
        ws = _th(h, w, A, r)
    elif kind == "tag":
        ws = _tag(h, w, A, r)
    else:
        ws, _d = _cleanup_spec()
        ws.height, ws.width, ws.num_agents, ws.vision_radius, ws.agent_type = h, w, A, r, [ws.agent_type[0]] * A
    eng, co = make_engine(ws, E, first=first), H.COracle(ws, E, first_env_id=first)
    assert "step_kernel<256" in eng.launch_info(), eng.launch_info()
    epoch = int(rng.integers(0, 4))
    eng.reset(epoch=epoch)
    co.reset(epoch)
    if eng.agent_state is not None:
        co.agent_state[...] = eng.agent_state.cpu().numpy()
    if eng.agent_dir is not None:
        co.agent_dir[...] = eng.agent_dir.cpu().numpy()
    ctx = f"case {case}: {kind} {h}x{w} A={A} r={r} E={E}"
    assert np.array_equal(eng.grid.cpu().numpy(), co.grid) and np.array_equal(eng.agent_pos.cpu().numpy(), co.pos), ctx + " reset"
    for t in range(1, 4):
        eng.step(random_actions=True, turn=t)
        assert co.step(epoch, t, random_actions=True) == 0
        assert_same(eng, co, ctx=ctx + f" fused turn {t}")
    nact = len(ws.action_dy)
    acts = rng.integers(0, nact, (E, A)).astype(np.uint8)
    ta = torch.from_numpy(acts).cuda()
    assert co.step(epoch, 4, actions=acts) == 0
    rows = eng.window_rows(None)
    eng.step(ta, sweep=True, no_move=True, turn=4)
    seen = torch.zeros_like(eng.obs)
    for a in range(A):
        seen[:, a] = eng.obs[:, a]
        eng.act(a, rows, action=ta[:, a].to(torch.int64).contiguous())
    torch.cuda.synchronize()
    assert np.array_equal(seen.cpu().numpy(), co.obs), ctx + " sgw_act turn: windows at pov time"
    assert_same(eng, co, what=("grid", "pos", "rewards", "total"), ctx=ctx + " sgw_act turn")
    if eng.agent_state is not None:
        assert np.array_equal(eng.agent_state.cpu().numpy(), co.agent_state), ctx
    if eng.agent_dir is not None:
        assert np.array_equal(eng.agent_dir.cpu().numpy(), co.agent_dir), ctx
    assert eng.status() == 0

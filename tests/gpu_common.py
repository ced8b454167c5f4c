"""What the GPU test files share (round 6: the tests of rounds 2-5 regrouped by component; no test body changed): the ``torch_cuda`` fixture, engine / world /
environment builders, case tables and comparison helpers.  Not collected by pytest (no ``test_`` prefix)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import gridstep_oracle as O
from sorrel_amd import _native as N
from tests import helpers as H
from tests.mixed_env import make_mixed_env, to_fixture_ids  # noqa: F401
from tests.test_gpu_parity import assert_same  # noqa: F401

__all__ = ['ROOT', 'torch_cuda', 'make_engine', '_tag_spec', '_cleanup_spec', 'KERNEL_CASES', 'make_env', '_run_bench', 'ROLLOUT_CASES', '_move_world', 'ROWS_CASES', 'PATCH_CASES', 'O_full', '_policy_env', '_long_horizon', '_values_and_expected', '_compare_turn', '_rollout_vs_oracle', '_float_world', 'SPEC_CASES', '_speculative_vs_oracle', '_LOOPS_SEEN', 'SWEEP_ROWS_CASES', 'N_ptr_array'] + ["assert_same", "make_mixed_env", "to_fixture_ids"]


ROOT = H.ROOT


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no silent CPU fallback)")
    return torch


def make_engine(ws, E, first=0, **kw):
    from sorrel_amd.engine import GridEngine

    return GridEngine(ws, E, device="cuda:0", first_env_id=first, **kw)


def _tag_spec(h, w, a, r):
    d, spec = H.load_golden("tag_9x9")
    ws = H.world_spec(spec)
    ws.height, ws.width, ws.num_agents, ws.vision_radius, ws.agent_type = h, w, a, r, [ws.agent_type[0]] * a
    return ws


def _cleanup_spec():
    d, spec = H.load_golden("cleanup_15x16")
    return H.world_spec(spec), d


KERNEL_CASES = [
    ("fast_static_c3", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.05, seed=11), {}, 37),
    ("fast_runtime_21", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(21, 21, 3, 2, spawn_prob=0.05, seed=12), {}, 50),
    ("fast_tag", lambda: _tag_spec(11, 11, 5, 4), {}, 40),
    ("big_64", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(64, 64, 12, 4, spawn_prob=0.05, seed=13, dense_prob=0.2), {}, 9),
    ("generic", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(18, 14, 4, 3, spawn_prob=0.05, seed=14), {"force_generic": 1}, 21),
]


def make_env(h, w, a, r, E, p=0.02, seed=5, model_factory=None, max_turns=100, extra_model=None):
    from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
    from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
    from sorrel_amd.examples.treasurehunt.main import make_config
    from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld

    cfg = make_config(h, w, a, r, spawn_prob=p, max_turns=max_turns)
    cfg["model"].update(extra_model or {})
    world = TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=seed)
    return TreasurehuntEnv(world, cfg, model_factory=model_factory)


def _run_bench(args, nproc, env_extra, plain=False):
    env = dict(os.environ, **env_extra)
    env.pop("SGW_FORCE_GENERIC", None)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if nproc == 1 or plain:         # plain: `python bench.py --gpus N` starts its own ranks
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", "29613", os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


ROLLOUT_CASES = [
    ("fast_static_c3", (32, 32, 8, 3, 70), {}),
    ("fast_static_c2", (16, 16, 4, 2, 33), {}),
    ("fast_runtime_stage", (24, 24, 4, 3, 50), {}),
    ("fast_runtime_plain", (24, 24, 4, 3, 50), {"stage": 0}),
    ("packed_16", (21, 21, 2, 2, 133), {"group": 16}),
    ("packed_32", (13, 9, 5, 4, 90), {"group": 32}),
    ("generic_64", (18, 14, 4, 3, 21), {"force_generic": 1}),
    ("big", (64, 64, 10, 4, 7), {}),
    ("generic_256", (64, 64, 10, 4, 5), {"force_generic": 1}),
]


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


# ------------------------------------------------------------------ sgw_observe_rows + sgw_act: windows rendered once, repaired by the movers
PATCH_CASES = ROWS_CASES + [
    (14, 18, 2, 5, 6, 3, 37, "float"),     # a non one-hot appearance table: sgw_observe renders, sgw_act repairs in float64
    (20, 16, 2, 6, 7, 2, 45, "u8"),        # compact uint8 windows
    (26, 22, 2, 6, 40, 2, 11, "plain"),    # 40 agents: a wave per env in sgw_act
    (12, 30, 1, 4, 20, 3, 14, "plain"),    # 20 agents: 32 lanes per env
]


def O_full(spec, grid):
    from oracle import gridstep_oracle as O

    return O.full_view(spec, grid)


def _policy_env(E, shape=(13, 15, 5, 2), memory=6, seed=5):
    import torch
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

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


# ------------------------------------------------------------------ the specialiser: every instance of a plan resolved at sgw_create
def _rollout_vs_oracle(torch, ws, eng, E, d, T=6):
    """sgw_rollout of T turns from the fixture's start against the C oracle turn by turn (final state + last turn's outputs)."""
    from tests.test_gpu_parity import assert_same

    co = H.COracle(ws, E, first_env_id=0)
    g0, p0 = d["grid0"][0], d["pos0"][0]
    eng.grid.copy_(torch.from_numpy(np.broadcast_to(g0, (E,) + g0.shape).copy()))
    eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(p0, (E,) + p0.shape).copy()))
    eng.total_reward.zero_()
    co.grid[...], co.pos[...], co.total[...] = g0, p0, 0
    eng.epoch, eng.turn = 0, 0
    eng.rollout(T)
    for t in range(1, T + 1):
        assert co.step(0, t, random_actions=True) == 0
    assert_same(eng, co, ctx=eng.launch_info().split(" group")[0])


# ------------------------------------------------------------------ speculative policy turns (sgw_turn_resolve)
def _float_world():
    d, spec = H.load_golden("float_appearance_3layer")
    return H.world_spec(spec)


SPEC_CASES = [
    ("c3_shape", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.02, seed=3, dense_prob=0.2), 96, 6),
    ("c3_shape_8200_envs", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.02, seed=3, dense_prob=0.2), 8200, 3),   # (from 8 192 envs on the dirty list is laid out by a scan, not by atomics)
    ("c5_shape", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=4, dense_prob=0.25), 10, 4),
    ("crowded_6x6", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(6, 6, 6, 2, spawn_prob=0.2, seed=5), 64, 8),
    ("ragged_9x13_rmax", lambda: __import__("sorrel_amd.spec", fromlist=["x"]).treasurehunt_spec(9, 13, 5, 4, spawn_prob=0.1, seed=6, dense_prob=0.3), 33, 6),
    ("float_tables_3layer", _float_world, 21, 6),
]


def _speculative_vs_oracle(torch, name, ws, E, T, first=0, epoch=0, seed=11):
    """The body of the two tests below; returns the largest number of passes a turn needed."""
    from sorrel_amd import _native as N
    from tests.test_gpu_parity import make_engine

    A, nact = ws.num_agents, len(ws.action_dy)
    eng = make_engine(ws, E, first=first)
    assert eng.capabilities() & N.CAP_RESOLVE
    co = H.COracle(ws, E, first_env_id=first)
    if name == "float_tables_3layer":           # (a world populated by the fixture: every env starts from its grid)
        d, _ = H.load_golden("float_appearance_3layer")
        g0, p0 = d["grid0"][0], d["pos0"][0]
        eng.grid.copy_(torch.from_numpy(np.broadcast_to(g0, (E,) + g0.shape).copy()))
        eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(p0, (E,) + p0.shape).copy()))
        eng.total_reward.zero_()
        co.grid[...], co.pos[...], co.total[...] = g0, p0, 0
    else:
        eng.reset(epoch)
        co.reset(epoch)
    eng.epoch = epoch
    rows = eng.speculation_rows()
    Nw = rows.shape[2]
    gen = torch.Generator().manual_seed(seed)
    Wt = torch.randn((A, Nw, nact), generator=gen).cuda()

    def policy(x, agents):                      # a linear layer per agent + argmax: a pure function of the window
        return torch.einsum("bn,bnk->bk", x, Wt[agents]).argmax(dim=1)

    agent_of_row = torch.arange(A, device="cuda:0").repeat_interleave(E)
    most = 0
    for t in range(1, T + 1):
        eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=t)
        eng.speculation_windows()                   # (sgw_observe_rows, or -- float tables -- the resolve kernel's render mode)
        flat = rows.view(A * E, Nw)
        taken = policy(flat, agent_of_row).view(A, E).clone()
        fresh, passes = taken.view(-1).contiguous(), 1
        while True:
            eng.turn_resolve(passes, None, fresh)
            lst = eng.spec_dirty(passes)
            dirty = eng._spec_state[2]                       # the same set as bytes [E][A]
            want = torch.nonzero(dirty.t().reshape(-1)).squeeze(1)
            assert torch.equal(torch.sort(lst).values, want), f"{name} turn {t} pass {passes}: the dirty list and the dirty bytes agree"
            if lst.numel() == 0:
                break
            passes += 1
            assert passes <= A + 1
            fresh = policy(flat[lst], lst // E)
            taken.view(-1)[lst] = fresh
        most = max(most, passes)
        torch.cuda.synchronize()
        assert bool(eng._spec_state[0].view(-1)[:E].all()), "every env committed"
        assert torch.equal(eng.actions, taken.t().to(torch.uint8)), "the actions tensor holds what the policies ended on"
        acts = taken.t().contiguous().cpu().numpy().astype(np.uint8)
        assert co.step(epoch, t, actions=acts) == 0
        for key, mine, ref in (("grid", eng.grid, co.grid), ("pos", eng.agent_pos, co.pos), ("rewards", eng.rewards, co.rewards),
                               ("total", eng.total_reward, co.total)):
            assert np.array_equal(mine.cpu().numpy(), ref), f"{name} turn {t}: {key}"
        seen = torch.from_numpy(co.obs.reshape(E, A, Nw)).cuda().permute(1, 0, 2).contiguous()       # what each agent saw when its turn came
        assert torch.equal(seen, rows), f"{name} turn {t}: the rows are the windows at pov time"
        assert torch.equal(policy(seen.view(A * E, Nw), agent_of_row).view(A, E), taken), f"{name} turn {t}: every action is the policy of that window"
    assert eng.status() == 0
    return most


_LOOPS_SEEN = {"cases": 0, "speculative": 0, "recorded": 0, "fast": 0}


SWEEP_ROWS_CASES = [
    ("c3_prebuilt_instance", (32, 32, 8, 3), 67, {"jit": 0}),
    ("c3_specialised", (32, 32, 8, 3), 67, {}),
    ("c2_shape", (16, 16, 4, 2), 33, {}),
    ("ragged_26x23_r2", (26, 23, 6, 2), 41, {}),
    ("30x30_r4_two_agents", (30, 30, 2, 4), 29, {}),
    ("24x40_r3_twelve_agents", (24, 40, 12, 3), 130, {}),
    ("radius_0_20x34", (20, 34, 6, 0), 27, {}),
]


def N_ptr_array(tensors):
    import ctypes

    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr

"""Round 5, GPU: agents that differ in their observation / action specs (one engine handle per distinct pair over the same world
tensors) against the reference-generated fixture and the oracle; step_big at eight waves per SIMD."""
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.mixed_env import make_mixed_env, to_fixture_ids
from oracle import gridstep_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no silent CPU fallback)")
    return torch


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


def test_agents_with_different_specs_vs_the_reference_fixture_device_random(torch_cuda):
    """Radius 2 / radius 4 / full_view / another entity list and fill kind / a float map, three action lists, RandomModel on every
    agent: sweep + per agent (window, act) on the handle of its own specs.  Every turn of the reference's run, envs 0 / 3 / 11."""
    torch = torch_cuda
    env, (d, base, views, full, defs) = make_mixed_env(12, "cuda:0")
    ids = [int(e) for e in d["env_ids"]]
    shapes = [d[f"obs_a{a}"].shape[2:] for a in range(len(defs))]
    eng = env._ensure_engine()
    assert env._mixed and len(env._group_engines) == 5
    torch.cuda.synchronize()
    for n, e in enumerate(ids):
        assert np.array_equal(to_fixture_ids(env, env.world.grid[e].cpu().numpy()), d["grid0"][n]) and np.array_equal(env.world.agent_pos[e].cpu().numpy(), d["pos0"][n])
    for t in range(d["grid"].shape[0]):
        env.take_turn()
        _compare_turn(torch, env, d, t, ids, shapes)
    env.raise_on_status()
    assert env.capture_turn() is None and "different" in str(env.capture_error)
    assert isinstance(env.obs, list) and len(env.obs) == 5


def test_agents_with_different_specs_policy_driven_with_replay_memories(torch_cuda):
    """The same world with a policy on every agent (pov -> get_action -> act -> add_memory, agent after agent): the policies replay the
    oracle's actions for all 12 envs; windows, rewards and the replay memories against the oracle's mixed rollout, the three fixture
    envs against the reference's own arrays."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel

    E = 12
    d, base, views, full, defs = H.load_mixed()
    T = d["grid"].shape[0]
    want = O.rollout_mixed(views, full, list(range(E)), T)
    clock = {"t": 0}

    class Replay(BaseModel):
        def __init__(self, input_size, action_space, slot):
            super().__init__(input_size, action_space, memory_size=T + 2, num_envs=E, device="cuda:0")
            self.slot = slot
            self.seen = []

        def take_action(self, state):
            self.seen.append(state.clone())
            return torch.from_numpy(want["actions"][clock["t"], :, self.slot].astype(np.int64)).to("cuda:0")

    env, _ = make_mixed_env(E, "cuda:0", model_factory=Replay)
    ids = [int(e) for e in d["env_ids"]]
    shapes = [d[f"obs_a{a}"].shape[2:] for a in range(len(defs))]
    for t in range(T):
        clock["t"] = t
        env.take_turn()
        _compare_turn(torch, env, d, t, ids, shapes)
        torch.cuda.synchronize()
        assert np.array_equal(env.rewards.cpu().numpy(), want["rewards"][t]) and np.array_equal(to_fixture_ids(env, env.world.grid.cpu().numpy()), want["grid"][t])
        for a, agent in enumerate(env.agents):
            assert np.array_equal(agent.model.seen[t].cpu().numpy().reshape((E,) + shapes[a]), want[f"obs_a{a}"][t]), (t, a)
    for a, agent in enumerate(env.agents):          # what add_memory stored: float32 windows, int64 actions, float32 rewards, done 0
        mem = agent.model.memory
        assert mem.size == T
        assert np.array_equal(mem.states[:T].cpu().numpy().reshape((T, E) + shapes[a]), want[f"obs_a{a}"])
        assert np.array_equal(mem.actions[:T].cpu().numpy().reshape(T, E), want["actions"][:, :, a].astype(np.int64))
        assert np.array_equal(mem.rewards[:T].cpu().numpy().reshape(T, E), want["rewards"][:, :, a])
        assert float(mem.dones.sum()) == 0.0
    env.raise_on_status()


def test_agents_with_different_specs_given_actions_and_a_bad_index(torch_cuda):
    """``take_turn(actions)``: indices into each agent's OWN list (agent 4 has three actions: index 3 is a KeyError there and only
    there), and two agents that share their specs share one handle."""
    torch = torch_cuda
    d, base, views, full, defs = H.load_mixed()
    defs = [defs[0], defs[4], defs[0], defs[2]]
    env, _ = make_mixed_env(6, "cuda:0", defs=defs)
    eng = env._ensure_engine()
    assert len(env._group_engines) == 3 and env._agent_engine[0] is env._agent_engine[2]
    v = [views[0], views[4], views[0], views[2]]
    f = [False, False, False, True]
    import dataclasses
    v = [dataclasses.replace(x, num_agents=4, agent_type=[6] * 4) for x in v]
    states = [O.reset_env(v[0], e, 0) for e in range(6)]
    rng = np.random.default_rng(5)
    for t in range(1, 9):
        acts = np.stack([rng.integers(0, [4, 3, 4, 5]) for _ in range(6)]).astype(np.uint8)
        env.take_turn(torch.from_numpy(acts).to("cuda:0"))
        torch.cuda.synchronize()
        for e in range(6):
            o, a_, r = O.step_env_mixed(v, f, states[e], e, 0, t, actions=acts[e])
            for a in range(4):
                assert np.array_equal(env.obs_of(a)[e].cpu().numpy(), o[a]), (t, e, a)
            assert np.array_equal(env.rewards[e].cpu().numpy(), r)
        assert np.array_equal(to_fixture_ids(env, env.world.grid.cpu().numpy()), np.stack([s.grid for s in states]))
    env.raise_on_status()
    bad = torch.zeros((6, 4), dtype=torch.uint8, device="cuda:0")
    bad[:, 1] = 3                                    # agent 1's list has three names
    env.take_turn(bad)
    with pytest.raises(KeyError):
        env.raise_on_status()


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


def test_a_refused_side_kernel_replans_the_whole_engine_for_the_prebuilt_instances(torch_cuda, tmp_path):
    """Round-4 advisor: only the whole-turn kernel used to be compiled at sgw_create; when the rollout instance (or any other) was refused
    later, its prebuilt run-time-shape twin ran under a plan laid out for compile-time shapes -- for a world with C % 4 != 0 (Cleanup:
    nine channels) the twin's grouped plane writes then ran past the staging area.  Now every instance is resolved at create and ANY
    refusal re-plans with jit = 0.  The Cleanup fixture's world: rollout instance refused -> prebuilt plan, results = the oracle."""
    torch = torch_cuda
    from sorrel_amd import _native as N
    from tests.test_gpu_parity import make_engine

    d, spec = H.load_golden("cleanup_15x16")
    ws = H.world_spec(spec)
    E = 24
    N.set_option("jit_cache_dir", str(tmp_path))
    plan = N.plan(ws.to_config(E, 0))
    assert plan["specialised"] == 1 and ws.num_channels % 4 != 0 and plan["kernel_rollout"] != plan["kernel"]
    good = make_engine(ws, E)
    assert "specialised=1" in good.launch_info()
    _rollout_vs_oracle(torch, ws, good, E, d)
    N.set_option("jit_refuse", plan["kernel_rollout"])
    eng = make_engine(ws, E)
    assert "specialised=0" in eng.launch_info(), eng.launch_info()          # the WHOLE engine runs the prebuilt plan
    with N.options(jit=0):
        assert eng.launch_info().split(" group")[0] == N.plan(ws.to_config(E, 0))["kernel"]
    _rollout_vs_oracle(torch, ws, eng, E, d)
    N.set_option("jit_refuse", None)
    # a row kernel that exists only specialised: refused -> the capability is not advertised (it used to be, and the first call failed)
    from sorrel_amd.spec import treasurehunt_spec

    own = treasurehunt_spec(20, 22, 3, 4, spawn_prob=0.05, seed=2)
    p2 = N.plan(own.to_config(E, 0))
    assert p2["kernel_observe_rows"].startswith("observe_rows<")
    N.set_option("jit_refuse", "observe_rows<")
    e2 = make_engine(own, E)
    assert "specialised=0" in e2.launch_info() and not (e2.capabilities() & N.CAP_OBSERVE_ROWS)
    N.set_option("jit_refuse", None)
    e3 = make_engine(own, E)
    assert "specialised=1" in e3.launch_info() and (e3.capabilities() & N.CAP_OBSERVE_ROWS)


def test_a_damaged_cache_file_is_recompiled_not_remembered_as_a_failure(torch_cuda, tmp_path):
    """A cached code object that does not load (truncated, bit-flipped, someone else's) is dropped and the instance compiled again on the
    spot -- the engine still runs its specialised plan -- and a file with a wrong checksum is never handed to hipModuleLoadData."""
    torch = torch_cuda
    from sorrel_amd import _native as N
    from sorrel_amd.spec import treasurehunt_spec
    from tests.test_gpu_parity import make_engine, assert_same

    N.set_option("jit_cache_dir", str(tmp_path))
    ws = treasurehunt_spec(17, 27, 4, 3, spawn_prob=0.05, seed=9)      # a shape no other test uses: nothing of it is loaded yet
    inst = N.plan(ws.to_config(16, 0))["kernel"]
    path = N.jit_compile(inst)
    blob = bytearray(open(path, "rb").read())
    blob[len(blob) // 2] ^= 0x5A                                       # flip a byte inside the code object: the checksum no longer matches
    open(path, "wb").write(bytes(blob))
    s0 = N.jit_stats()
    eng = make_engine(ws, 16)
    s1 = N.jit_stats()
    # (the damaged whole-turn instance is compiled again; the engine's other instances -- rollout, ... -- are new to this cache as well)
    assert "specialised=1" in eng.launch_info() and s1["compiled"] >= s0["compiled"] + 1 and s1["failed"] == s0["failed"]
    lowered, code = N.jit_code_object(path)                            # rewritten whole
    assert code[:4] == b"\x7fELF"
    co = H.COracle(ws, 16, first_env_id=0)
    eng.reset(0)
    co.reset(0)
    for t in range(1, 4):
        assert co.step(0, t, random_actions=True) == 0
        eng.step(random_actions=True)
        assert_same(eng, co, ctx=f"turn {t}")


def test_a_capture_that_fails_half_way_leaves_the_replay_rings_as_the_eager_loop_left_them(torch_cuda):
    """Round-4 advisor: when the LAST agent's forward pass does something a capture forbids (a host synchronisation), the agents before it
    have already counted a deferred add_memory inside the failed capture; capture_turn() must hand the rings back exactly as the warm-up
    turns left them, and the eager loop carries on as if nothing had been tried."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E = 9

    class Policy(BaseModel):
        made = [0]

        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=8, num_envs=E, device="cuda:0")
            self.slot = Policy.made[0] % 4
            Policy.made[0] += 1
            self.capturing_ok = True

        def take_action(self, state):
            s = state.reshape(state.shape[0], -1).sum(dim=1)
            if self.slot == 3 and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("this forward pass cannot be recorded")     # (what a host synchronisation under capture ends in)
            return (s.long() + self.slot) % 4

    def fresh():
        Policy.made[0] = 0
        return make_env(12, 14, 4, 2, E, p=0.05, seed=4, model_factory=Policy)

    a, b = fresh(), fresh()
    assert b.capture_turn(warmup=2) is None and b.capture_error is not None
    for _ in range(2):
        a.take_turn()
    for ag_a, ag_b in zip(a.agents, b.agents):
        ma, mb = ag_a.model.memory, ag_b.model.memory
        assert (mb.idx, mb.size) == (ma.idx, ma.size) == (2, 2) and not mb._deferred and mb._deferred_adds == 0
    for _ in range(5):                          # ... and on: the eager loops agree, every ring row included
        a.take_turn()
        b.take_turn()
    torch.cuda.synchronize()
    assert torch.equal(a.world.grid, b.world.grid) and torch.equal(a.total_reward, b.total_reward)
    for ag_a, ag_b in zip(a.agents, b.agents):
        ma, mb = ag_a.model.memory, ag_b.model.memory
        assert (mb.idx, mb.size) == (ma.idx, ma.size)
        assert torch.equal(ma.states, mb.states) and torch.equal(ma.actions, mb.actions) and torch.equal(ma.rewards, mb.rewards)


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


@pytest.mark.parametrize("case", SPEC_CASES, ids=[c[0] for c in SPEC_CASES])
def test_speculative_turn_reaches_the_sequential_turn(torch_cuda, case):
    """sgw_turn_resolve through the C ABI: sweep, every pre-move window, one batched policy evaluation, then resolve / re-evaluate the
    dirty rows until nobody is dirty.  Against the C oracle's agent-after-agent turn: (1) stepping the actions the speculation ended
    on gives the engine's grid, positions, rewards and totals; (2) the window each agent had when ITS turn came (the oracle's
    observation) is the row its action was computed on, and the policy of that window is that action -- i.e. the fixed point IS the
    sequential policy-driven turn.  Also: committed envs are skipped, passes stay far below A, every env ends done."""
    name, mk, E, T = case
    most = _speculative_vs_oracle(torch_cuda, name, mk(), E, T)
    assert most <= (5 if name != "crowded_6x6" else 7), most


@pytest.mark.parametrize("case", range(int(os.environ.get("SGW_SOAK", "24"))))
def test_speculative_turn_soak_random_worlds(torch_cuda, case):
    """The same check on random Treasurehunt-like worlds: maps from 5x5 to 90x90 (wave-per-env and workgroup-per-env step kernels; the
    resolve kernel with one wave and with four per env from 16 agents on), 1 ... 64 agents, radii 1 ... 6, sparse to crowded, random
    batch sizes, global env ids and epochs; every eighth case has more than 8 192 envs (dirty list by scan)."""
    from sorrel_amd.spec import treasurehunt_spec

    rng = np.random.default_rng(77000 + case)
    h, w = int(rng.integers(5, 91)), int(rng.integers(5, 91))
    if case % 8 == 7:
        h, w = int(rng.integers(5, 20)), int(rng.integers(5, 20))
    free = (h - 2) * (w - 2)
    a = int(min(rng.integers(1, 65), max(1, free // 3)))
    r = min(int(rng.integers(1, 5 if case % 8 == 7 else 7)), (min(h, w) - 1) // 2)      # (visual_field's own limit)
    ws = treasurehunt_spec(h, w, a, r, spawn_prob=float(rng.choice([0.0, 0.01, 0.1, 0.5])), seed=int(rng.integers(0, 2**31)),
                           dense_prob=float(rng.choice([0.0, 0.2, 0.6])), gem_value=int(rng.integers(1, 20)), bone_value=-int(rng.integers(1, 20)))
    E = int(rng.integers(8193, 9000)) if case % 8 == 7 else int(rng.integers(1, 70))
    T = 2 if case % 8 == 7 else int(rng.integers(2, 7))
    _speculative_vs_oracle(torch_cuda, f"soak {case} ({h}x{w}, {a} agents, r {r}, {E} envs)", ws, E, T, first=int(rng.integers(0, 2**31)),
                           epoch=int(rng.integers(0, 9)), seed=case)


def test_environment_speculative_turns_equal_the_eager_loop(torch_cuda):
    """Environment.speculate_turns: agents that share one model (one batched forward pass per pass, one shared replay ring filled in
    agent order) and agents with a model each -- grids, totals, step outputs and every replay row equal the eager agent-after-agent
    loop's after 12 turns and a reset."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E, A = 40, 6

    class Linear(BaseModel):
        def __init__(self, input_size, action_space, memory=0):
            super().__init__(input_size, action_space, memory_size=memory, num_envs=E, device="cuda:0")
            g = torch.Generator().manual_seed(99)
            self.weight = torch.randn((int(np.prod(input_size)), action_space), generator=g).cuda()
            self.calls = 0

        def take_action(self, state):
            self.calls += 1
            return (state.reshape(state.shape[0], -1) @ self.weight).argmax(dim=1)

    for shared, cap in ((1, 4 * A), (1, 4 * A + 1), (2, 3 * A)):             # (4 A: the turn's rows of the shared ring are contiguous -- the windows are
        envs = []                                                             # rendered straight into them; 4 A + 1: they are not, add_batch copies;
        for speculate in (False, True):                                       # 2: two models of three agents each -- two batches per pass)
            made = []

            def factory(input_size, action_space):
                k = len(made) * shared // A
                made.append(k)
                if k >= len(models):
                    models.append(Linear(input_size, action_space, memory=cap))
                return models[k]

            models = []

            env = make_env(14, 17, A, 3, E, p=0.06, seed=7, model_factory=factory)
            env.speculate_turns = "always" if speculate else False      # ("always": also where the cost model would keep the sequential loop)
            envs.append(env)
        eager, spec = envs
        for t in range(12):
            if t == 7:
                eager.reset()
                spec.reset()
            eager.take_turn()
            spec.take_turn()
        torch.cuda.synchronize()
        assert spec.speculation_passes >= 1 and spec._speculation_groups(spec._engine) is not None
        assert len(spec._speculation_groups(spec._engine)) == shared
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(eager.world, name), getattr(spec.world, name)), (shared, name)
        assert torch.equal(eager.rewards, spec.rewards) and torch.equal(eager.actions, spec.actions)
        for a in range(A):
            ma, mb = eager.agents[a].model.memory, spec.agents[a].model.memory
            assert (ma.idx, ma.size) == (mb.idx, mb.size)
            assert torch.equal(ma.states, mb.states) and torch.equal(ma.actions, mb.actions) and torch.equal(ma.rewards, mb.rewards)
            assert torch.equal(ma.dones, mb.dones)
        assert spec.agents[0].model.calls < eager.agents[0].model.calls          # one forward pass per PASS, not per agent
        for a in range(A):          # obs_of: the window each agent acted on (the eager loop's lives in its replay row; the last add is the last turn's)
            mem = eager.agents[a].model.memory
            k = sum(1 for b in range(a + 1, A) if eager.agents[b].model.memory is mem)
            last = (mem.idx - 1 - k) % mem.capacity
            assert torch.equal(spec.obs_of(a).reshape(E, -1), mem.states[last].reshape(E, -1)), (shared, a)
        eager.raise_on_status()
        spec.raise_on_status()


@pytest.mark.parametrize("case", ["own_rings", "shared_ring", "values_and_ints", "model_edits_the_world", "subclass_with_own_pov"])
def test_fast_policy_loop_equals_the_generic_transition_loop(torch_cuda, case):
    """Environment.fast_policy_loop (agents with the standard hooks stepped without the generic hooks in between) against the
    Agent.transition loop it replaces: grids, positions, totals, step outputs and every replay row after 11 turns and a reset --
    for a ring per agent, one shared ring whose rows wrap mid-turn, agents that return action values (in-kernel argmax / exploration)
    or a plain int, a model that edits the world between two agents (windows rendered on demand from there on), and a subclass that
    overrides pov (the fast loop must not take it)."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.examples.treasurehunt.agents import TreasurehuntAgent
    from sorrel_amd.examples.treasurehunt.entities import Wall
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E, A = 37, 5

    class Linear(BaseModel):
        def __init__(self, input_size, action_space, k=0, memory=9):
            super().__init__(input_size, action_space, memory_size=memory, num_envs=E, device="cuda:0")
            g = torch.Generator().manual_seed(50 + k)
            self.weight = torch.randn((int(np.prod(input_size)), action_space), generator=g).cuda()
            self.k, self.env, self.turns = k, None, 0
            self.epsilon = 0.25 if (case == "values_and_ints" and k == 1) else 0.0

        def take_action(self, state):
            q = state.reshape(state.shape[0], -1) @ self.weight
            if case == "values_and_ints" and self.k in (1, 3):
                return q                                   # action values: the act launch chooses
            if case == "values_and_ints" and self.k == 2:
                return 1                                   # a plain int for every env
            if case == "values_and_ints" and self.k == 4:
                return q.argmax(dim=1).to(torch.int32)
            if case == "model_edits_the_world" and self.k == 2:
                self.turns += 1
                if self.turns % 3 == 0:
                    self.env.world.add((1 + self.turns % 5, 2, 0), Wall(), env=None)
            return q.argmax(dim=1)

    envs = []
    for fast in (False, True):
        made = []

        def factory(input_size, action_space):
            if case == "shared_ring":
                if not made:
                    made.append(Linear(input_size, action_space, 0, memory=0))
                    made[0].memory = Buffer(capacity=2 * A + 3, obs_shape=tuple(input_size), num_envs=E, device="cuda:0")
                return made[0]
            made.append(Linear(input_size, action_space, len(made)))
            return made[-1]

        env = make_env(13, 16, A, 2, E, p=0.07, seed=11, model_factory=factory)
        for m in made:
            m.env = env
        if case == "subclass_with_own_pov":
            class Dimmed(TreasurehuntAgent):
                def pov(self, world):
                    return super().pov(world) * 0.5
            env.agents[3].__class__ = Dimmed
        env.fast_policy_loop = fast
        envs.append(env)
    generic, quick = envs
    for t in range(11):
        if t == 6:
            generic.reset()
            quick.reset()
        generic.take_turn()
        quick.take_turn()
    torch.cuda.synchronize()
    plan = quick._fast_plan(quick._engine)
    assert (plan is None) == (case == "subclass_with_own_pov") and generic.__dict__.get("_fast_plan_cache") is None
    for name in ("grid", "agent_pos", "total_reward"):
        assert torch.equal(getattr(generic.world, name), getattr(quick.world, name)), name
    assert torch.equal(generic.rewards, quick.rewards) and torch.equal(generic.actions, quick.actions)
    assert float(quick.world.total_reward.abs().sum()) > 0
    for a in range(A):
        ma, mb = generic.agents[a].model.memory, quick.agents[a].model.memory
        assert (ma.idx, ma.size, ma._dones_dirty) == (mb.idx, mb.size, mb._dones_dirty)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(ma, name), getattr(mb, name)), (a, name)
    generic.raise_on_status()
    quick.raise_on_status()


_LOOPS_SEEN = {"cases": 0, "speculative": 0, "recorded": 0, "fast": 0}


@pytest.mark.parametrize("case", range(int(os.environ.get("SGW_SOAK", "16")) // 2))
def test_environment_turn_loops_soak(torch_cuda, case):
    """Random Treasurehunt environments through the Python API, policy-driven, four ways: the generic Agent.transition loop, the fast
    loop, the speculative turn (where the agents share few enough models) and a recorded turn -- same seeds, same policies: every world
    tensor, step output and replay row equal after a few turns, a reset and a few more."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    rng = np.random.default_rng(52000 + case)
    h, w = int(rng.integers(7, 70)), int(rng.integers(7, 70))
    A = int(min(rng.integers(1, 25), max(1, (h - 2) * (w - 2) // 4)))
    r = min(int(rng.integers(1, 6)), (min(h, w) - 1) // 2)
    E = int(rng.integers(1, 50))
    n_models = int(rng.choice([1, 1, 2, A]))                      # agents per model: all share one, two groups, or a model each
    n_models = max(1, min(n_models, A))
    cap = int(rng.integers(A, 4 * A + 3))                          # (a shared ring takes A rows per turn: wraps mid-turn unless a multiple)
    p, seed, T = float(rng.choice([0.0, 0.02, 0.2])), int(rng.integers(0, 2**31)), int(rng.integers(3, 8))

    class Linear(BaseModel):
        def __init__(self, input_size, action_space, k):
            super().__init__(input_size, action_space, memory_size=0, num_envs=E, device="cuda:0")
            self.memory = Buffer(capacity=cap, obs_shape=tuple(input_size), num_envs=E, device="cuda:0")
            g = torch.Generator().manual_seed(1000 * case + k)
            self.weight = torch.randn((int(np.prod(input_size)), action_space), generator=g).cuda()

        def take_action(self, state):
            return (state.reshape(state.shape[0], -1) @ self.weight).argmax(dim=1)

    def build(mode):
        models, made = [], []

        def factory(input_size, action_space):
            k = len(made) * n_models // A
            made.append(k)
            if k >= len(models):
                models.append(Linear(input_size, action_space, k))
            return models[k]

        env = make_env(h, w, A, r, E, p=p, seed=seed % 1000, model_factory=factory)
        env.fast_policy_loop = mode != "generic"
        env.speculate_turns = "always" if mode == "speculative" else False
        if mode == "recorded":
            env.capture_turn(warmup=1)                             # (may decline: the eager loop then plays, which is as good a check)
        return env

    envs = {mode: build(mode) for mode in ("generic", "fast", "speculative", "recorded")}
    envs["generic"].take_turn()
    envs["fast"].take_turn()
    envs["speculative"].take_turn()
    if envs["recorded"]._captured is None:
        envs["recorded"].take_turn()
    for t in range(2 * T):
        for env in envs.values():
            if t == T:
                env.reset()
            env.take_turn()
    torch.cuda.synchronize()
    ref = envs["generic"]
    ctx = f"case {case}: {h}x{w}, {A} agents on {n_models} models, r {r}, {E} envs, ring of {cap}"
    _LOOPS_SEEN["cases"] += 1
    _LOOPS_SEEN["speculative"] += int(getattr(envs["speculative"], "speculation_passes", 0) > 0)
    _LOOPS_SEEN["recorded"] += int(envs["recorded"]._captured is not None and envs["recorded"]._captured.turns_replayed > 0)
    _LOOPS_SEEN["fast"] += int(envs["fast"]._fast_plan(envs["fast"]._engine) is not None)
    for mode, env in envs.items():
        if mode == "generic":
            continue
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(ref.world, name), getattr(env.world, name)), (ctx, mode, name)
        assert torch.equal(ref.rewards, env.rewards) and torch.equal(ref.actions, env.actions), (ctx, mode)
        for a in range(A):
            ma, mb = ref.agents[a].model.memory, env.agents[a].model.memory
            assert (ma.idx, ma.size) == (mb.idx, mb.size), (ctx, mode, a)
            for name in ("states", "actions", "rewards", "dones"):
                assert torch.equal(getattr(ma, name), getattr(mb, name)), (ctx, mode, a, name)
        env.raise_on_status()


def test_environment_turn_loops_soak_reached_every_loop():
    """... and the soak above did run what it names (a case whose agents have a model each keeps the sequential turn; a capture may decline)."""
    seen = _LOOPS_SEEN
    if seen["cases"] < 8:
        pytest.skip("the soak did not run in this session")
    assert seen["fast"] == seen["cases"] and seen["speculative"] >= seen["cases"] // 4 and seen["recorded"] >= seen["cases"] // 2, seen


# ------------------------------------------------------------------ recorded turns at a batch where the kernels change form
@pytest.mark.parametrize("layout", ["rows", "tensor"])
@pytest.mark.parametrize("agents", [8, 12])
def test_recorded_turn_at_16384_envs_vs_the_oracle(torch_cuda, layout, agents):
    """Round-4 review: every recorded-turn test ran at <= 45 envs, although the "rows" layout's double write and sgw_act's 16-lane form (9-16
    agents) switch on the batch size.  16 384 envs of the headline's world, 8 and 12 agents, both layouts: ten replayed turns against the C
    oracle stepping the actions the policies chose -- state, rewards, totals, and the replay row each agent's window went to."""
    torch = torch_cuda
    from tests.test_gpu_round4 import _policy_env

    E = 16384
    b = _policy_env(E, shape=(32, 32, agents, 3), memory=4, seed=9)
    b.capture_layout = layout
    cap = b.capture_turn(warmup=2)
    assert cap is not None, getattr(b, "capture_error", None)
    ws = b._engine.spec
    co = H.COracle(ws, E, first_env_id=0, threads=16)
    torch.cuda.synchronize()
    co.grid[...] = b.world.grid.cpu().numpy()
    co.pos[...] = b.world.agent_pos.cpu().numpy()
    co.total[...] = b.world.total_reward.cpu().numpy()
    for t in range(10):
        b.take_turn()
        torch.cuda.synchronize()
        assert co.step(b.epoch, b.turn, actions=b.actions.cpu().numpy()) == 0
        assert np.array_equal(b.world.grid.cpu().numpy(), co.grid) and np.array_equal(b.world.agent_pos.cpu().numpy(), co.pos), t
        assert np.array_equal(b.rewards.cpu().numpy(), co.rewards) and np.array_equal(b.world.total_reward.cpu().numpy(), co.total), t
        for k, agent in enumerate(b.agents):
            mem = agent.model.memory
            last = (mem.idx - 1) % mem.capacity
            assert np.array_equal(mem.states[last].cpu().numpy().reshape(E, -1), co.obs[:, k].reshape(E, -1)), (t, k)
            assert np.array_equal(mem.actions[last].cpu().numpy(), b.actions[:, k].cpu().numpy().astype(np.int64)), (t, k)
            assert np.array_equal(mem.rewards[last].cpu().numpy(), co.rewards[:, k]), (t, k)
    assert cap.turns_replayed == 10
    b.raise_on_status()


def test_recorded_turn_of_agents_that_share_a_frame_stacking_ring(torch_cuda):
    """Round-4 review, "missing" 5: agents that share ONE ring with ``n_frames = 3`` (sorrel/buffers.py:143-154: agent k's stack is the last
    two rows of the shared ring, i.e. the windows agents k-1 and k-2 acted on this very turn) were refused by capture_turn().  In the "rows"
    layout every window sits in its replay row from the start of the turn, so the gather by the device's row count (sgw_turn_prev_rows of
    the asking agent's own slot) finds them: 26 replayed turns over a 7-row ring (the turn's three rows wrap in most turns), a reset with
    add_empty, equal the eager loop in every replay row."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E, A = 21, 3
    rings = []

    class Stacked(BaseModel):
        n_frames = 3

        def __init__(self, input_size, action_space):
            super().__init__(input_size, action_space, memory_size=0, num_envs=E, device="cuda:0")
            n = int(np.prod(input_size))
            if len(rings) % A == 0:
                rings.append(Buffer(capacity=7, obs_shape=(n,), n_frames=3, num_envs=E, device="cuda:0"))
            else:
                rings.append(rings[-1])
            self.memory = rings[-1]
            self.weight = torch.randn((3 * n, action_space), generator=torch.Generator().manual_seed(len(rings) % A)).cuda()

        def take_action(self, state):
            assert state.shape[1] == self.weight.shape[0]
            return (state @ self.weight).argmax(dim=1)

    a, b = (make_env(12, 13, A, 2, E, p=0.05, seed=4, model_factory=Stacked) for _ in range(2))
    assert a.agents[0].model.memory is a.agents[2].model.memory is not b.agents[0].model.memory
    cap = b.capture_turn(warmup=2)
    assert cap is not None, getattr(b, "capture_error", None)
    for _ in range(2):
        a.take_turn()
    for t in range(26):
        if t == 15:
            for env in (a, b):
                env.reset()
                env.agents[0].model.memory.add_empty()
        a.take_turn()
        b.take_turn()
        torch.cuda.synchronize()
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert torch.equal(a.actions, b.actions) and torch.equal(a.rewards, b.rewards), t
        mx, my = a.agents[0].model.memory, b.agents[0].model.memory
        assert (mx.idx, mx.size) == (my.idx, my.size), t
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(mx, name), getattr(my, name)), (t, name)
    assert cap.turns_replayed == 26 and float(a.world.total_reward.abs().sum()) > 0
    b.raise_on_status()


def test_capture_turn_declines_where_a_replay_would_be_slower(torch_cuda):
    """Few agents, hundreds of MB of windows per turn: the second copy of every window costs more than the host time a replay saves (32x32 /
    8 agents at 65 536 envs: 610 us recorded, 500 eager) -- capture_turn() keeps the eager loop and says why; force=True records."""
    from tests.test_gpu_round4 import _policy_env

    env = _policy_env(4096, shape=(32, 32, 8, 3), memory=2)
    env.capture_max_window_bytes = 16 << 20            # (the same rule at a size a test can afford: 4 096 x 8 x 294 x 4 = 38.5 MB)
    assert env.capture_turn() is None and "twice" in str(env.capture_error)
    env.take_turn()
    assert env.capture_turn(force=True) is not None
    env.take_turn()
    env.raise_on_status()
    # where the eager loop is the fast one the crossover is lower: its own limit applies (and only there)
    env = _policy_env(4096, shape=(32, 32, 8, 3), memory=2)
    env.capture_max_window_bytes_per_agent_fast = 2 << 20          # (4 096 envs x 294 x 4 = 4.8 MB per agent)
    fast = env._fast_plan(env._ensure_engine()) is not None
    assert (env.capture_turn() is None) == fast
    env.fast_policy_loop = False
    assert env.capture_turn() is not None
    env.take_turn()
    env.raise_on_status()


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


SWEEP_ROWS_CASES = [
    ("c3_prebuilt_instance", (32, 32, 8, 3), 67, {"jit": 0}),
    ("c3_specialised", (32, 32, 8, 3), 67, {}),
    ("c2_shape", (16, 16, 4, 2), 33, {}),
    ("ragged_26x23_r2", (26, 23, 6, 2), 41, {}),
    ("30x30_r4_two_agents", (30, 30, 2, 4), 29, {}),
    ("24x40_r3_twelve_agents", (24, 40, 12, 3), 130, {}),
    ("radius_0_20x34", (20, 34, 6, 0), 27, {}),
]


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


def N_ptr_array(tensors):
    import ctypes

    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


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


def test_speculate_turns_true_follows_the_cost_model(torch_cuda):
    """``speculate_turns = True`` speculates only where the measured cost model says it is the faster turn: many agents on one model yes, few
    agents over a large batch no (the sequential loop is device-bound there); "always" speculates wherever it is possible."""
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    def env_of(h, w, A, r, E):
        one = []

        class Shared(BaseModel):
            def __init__(self, input_size, action_space):
                super().__init__(input_size, action_space, memory_size=2 * A, num_envs=E, device="cuda:0")

            def take_action(self, state):
                return state.reshape(state.shape[0], -1).sum(dim=1).long() % 4

        def factory(input_size, action_space):
            if not one:
                one.append(Shared(input_size, action_space))
            return one[0]

        return make_env(h, w, A, r, E, p=0.02, seed=3, model_factory=factory)

    many = env_of(24, 24, 20, 2, 64)
    many.speculate_turns = True
    many.take_turn()
    assert many._speculation_groups(many._engine) is not None and many.speculation_passes >= 1
    few = env_of(32, 32, 8, 3, 16384)                      # 154 MB of windows, eight agents: 406 us speculative against 292 sequential
    few.speculate_turns = True
    few.take_turn()
    assert few._speculation_groups(few._engine) is None and not hasattr(few, "speculation_passes")
    few.speculate_turns = "always"
    few.take_turn()
    assert few.speculation_passes >= 1
    many.raise_on_status()
    few.raise_on_status()


@pytest.mark.parametrize("which", ["tag", "cleanup"])
def test_fast_policy_loop_on_the_tag_and_cleanup_examples(torch_cuda, which):
    """The shipped Tag and Cleanup agents -- pov = the engine's row (window + the "it" flag / the positional code), get_action =
    model.take_action -- go through the fast eager loop too: 30 turns across ring wrap-arounds and a reset leave exactly what the generic
    Agent.transition loop leaves (world, agent state, step outputs, every replay row incl. its tail)."""
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

    def make(fast):
        if which == "tag":
            from sorrel_amd.entities import EmptyEntity
            from sorrel_amd.examples.tag.env import TagEnv
            from sorrel_amd.worlds import Gridworld

            cfg = {"agent": {"num_agents": 6, "vision_radius": 2, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 50}}
            env = TagEnv(Gridworld(8, 9, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=31), cfg, model_factory=Policy)
        else:
            from tests.test_api_host import make_cleanup_env

            env = make_cleanup_env(E=E, seed=7, device="cuda:0", model_factory=Policy)
        env.fast_policy_loop = fast
        return env

    a, b = make(False), make(True)
    for t in range(30):
        if t == 17:
            a.reset(); b.reset()
        a.take_turn()
        b.take_turn()
    torch.cuda.synchronize()
    eng = b._engine
    assert eng.row_tail == (1 if which == "tag" else 12)
    assert b._fast_plan(eng) is not None and a.__dict__.get("_fast_plan_cache") is None
    for name in ("grid", "agent_pos", "total_reward") + (("agent_state",) if which == "tag" else ("agent_dir",)):
        assert torch.equal(getattr(a.world, name), getattr(b.world, name)), name
    assert torch.equal(a.rewards, b.rewards) and torch.equal(a.actions, b.actions)
    assert float(b.world.total_reward.abs().sum()) > 0
    for x, y in zip(a.agents, b.agents):
        mx, my = x.model.memory, y.model.memory
        assert (mx.idx, mx.size) == (my.idx, my.size)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(mx, name), getattr(my, name)), name
    a.raise_on_status()
    b.raise_on_status()


def test_turn_plan_names_the_loop_that_plays(torch_cuda):
    """Environment.turn_plan(): the diagnostic agrees with what take_turn() then does."""
    from sorrel_amd.examples.treasurehunt.agents import TreasurehuntAgent
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env
    from tests.test_gpu_round4 import _policy_env

    rnd = make_env(14, 14, 3, 2, 16)                                   # RandomModel agents
    assert rnd.turn_plan()["loop"] == "fused"
    env = _policy_env(64, shape=(32, 32, 8, 3), memory=4)
    plan = env.turn_plan()
    assert plan["loop"] == "fast" and plan["one_launch_windows"] is True and plan["launches"] == 9, plan
    env.fuse_sweep_and_rows = False
    assert env.turn_plan()["launches"] == 10
    env.fast_policy_loop = False
    assert env.turn_plan()["loop"] == "generic" and "switched off" in env.turn_plan()["fast"]
    env.fast_policy_loop = True

    class Own(TreasurehuntAgent):
        def get_action(self, state):
            return super().get_action(state)

    env.agents[2].__class__ = Own
    env.__dict__.pop("_fast_plan_cache", None)
    assert env.turn_plan()["loop"] == "generic"
    env.agents[2].__class__ = TreasurehuntAgent
    env.__dict__.pop("_fast_plan_cache", None)
    assert env.capture_turn() is not None
    assert env.turn_plan()["loop"] == "recorded"
    env.take_turn()
    mixed = make_mixed_env(9, "cuda:0")[0]
    assert mixed.turn_plan()["loop"] == "per-agent handles" and mixed.turn_plan()["handles"] >= 2

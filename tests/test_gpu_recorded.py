"""A whole policy-driven turn recorded as one graph (Environment.capture_turn, sgw_turn_*): device-side turn and row counters, frame stacks, action values, the examples.
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


def test_captured_turn_with_action_values_follows_a_decaying_epsilon(torch_cuda):
    """Agents whose get_action returns the model's action VALUES: Environment hands them to the act launch (no argmax launch, exploration
    in-kernel at the model's epsilon).  A recorded turn -- one node less per agent -- equals the eager loop and the oracle over 40 turns
    while epsilon decays every few turns and across an epoch reset; the buffers hold the actions TAKEN."""
    torch = torch_cuda
    from oracle import gridstep_oracle as O
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

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


def test_captured_turn_with_frame_stacks(torch_cuda):
    """Memories with ``n_frames = 3`` (the reference's Cleanup / IQN configs stack frames: ``Buffer.current_state``,
    sorrel/buffers.py:143-154, in front of the window): in a recorded turn the previous frames are gathered by the device's own row
    count (sgw_turn_prev_rows) -- 30 turns with wrap-arounds of the 5-row rings, an epoch reset and an ``add_empty`` at its start equal
    the eager loop; the gather itself against the ring for every position of the row counter."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

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
    # agents that SHARE a frame-stacking ring: recorded only where the windows reach the ring as the turn goes (below)
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


def test_a_capture_that_fails_half_way_leaves_the_replay_rings_as_the_eager_loop_left_them(torch_cuda):
    """Round-4 advisor: when the LAST agent's forward pass does something a capture forbids (a host synchronisation), the agents before it
    have already counted a deferred add_memory inside the failed capture; capture_turn() must hand the rings back exactly as the warm-up
    turns left them, and the eager loop carries on as if nothing had been tried."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from tests.gpu_common import make_env

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


# ------------------------------------------------------------------ recorded turns at a batch where the kernels change form
@pytest.mark.parametrize("layout", ["rows", "tensor"])
@pytest.mark.parametrize("agents", [8, 12])
def test_recorded_turn_at_16384_envs_vs_the_oracle(torch_cuda, layout, agents):
    """Round-4 review: every recorded-turn test ran at <= 45 envs, although the "rows" layout's double write and sgw_act's 16-lane form (9-16
    agents) switch on the batch size.  16 384 envs of the headline's world, 8 and 12 agents, both layouts: ten replayed turns against the C
    oracle stepping the actions the policies chose -- state, rewards, totals, and the replay row each agent's window went to."""
    torch = torch_cuda
    from tests.gpu_common import _policy_env

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
    from tests.gpu_common import make_env

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
    from tests.gpu_common import _policy_env

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

"""The turn loops of ``Environment.take_turn`` against each other (round 6 on: grouped by component; the older cases live in
``test_gpu_round{2..5}.py``).  Reference: ``Agent.transition``, ``sorrel/agents/agent.py:155-173``; exploration
``sorrel/models/pytorch/iqn.py:294-309``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch


@pytest.mark.parametrize("shared", [1, 2])
def test_speculative_turns_with_action_values_explore_like_the_sequential_turn(torch_cuda, shared):
    """A policy that returns action VALUES with epsilon > 0 (round-5 advisor finding: the speculative turn took a plain argmax): the eager
    loops hand the values to ``sgw_act`` (``SGW_ACT_QF32``), which explores in-kernel with the engine's keyed draw for (env, turn, agent);
    the speculative turn takes the same choice through ``sgw_choose_actions`` -- state, step outputs and replay rings are equal turn after
    turn, with an epsilon that decays (and differs per model) on the way, and exploration really happens."""
    torch = torch_cuda
    from sorrel_amd.models import BaseModel
    from tests.test_gpu_round2 import make_env

    E, A = 48, 6

    class Values(BaseModel):
        def __init__(self, input_size, action_space, memory, eps):
            super().__init__(input_size, action_space, memory_size=memory, num_envs=E, device="cuda:0")
            g = torch.Generator().manual_seed(5)
            self.weight = torch.randn((int(np.prod(input_size)), action_space), generator=g).cuda()
            self.epsilon = eps

        def take_action(self, state):
            return state.reshape(state.shape[0], -1) @ self.weight          # [n, n_actions]: the engine takes the argmax / explores

    envs = []
    for speculate in (False, True):
        made, models = [], []

        def factory(input_size, action_space):
            k = len(made) * shared // A
            made.append(k)
            if k >= len(models):
                models.append(Values(input_size, action_space, 4 * A, 0.5 if k == 0 else 0.25))
            return models[k]

        env = make_env(14, 17, A, 3, E, p=0.06, seed=11, model_factory=factory)
        env.speculate_turns = "always" if speculate else False
        envs.append((env, models))
    (eager, em), (spec, sm) = envs
    greedy_differs = 0
    for t in range(10):
        if t == 6:
            eager.reset()
            spec.reset()
        for m in em + sm:
            m.epsilon *= 0.9                                                  # a decaying epsilon reaches both loops
        eager.take_turn()
        spec.take_turn()
        assert spec.turn_plan()["loop"] == "speculative"
        torch.cuda.synchronize()
        for name in ("grid", "agent_pos", "total_reward"):
            assert torch.equal(getattr(eager.world, name), getattr(spec.world, name)), (t, name)
        assert torch.equal(eager.rewards, spec.rewards) and torch.equal(eager.actions, spec.actions), t
        # what a greedy choice over the windows the agents acted on would have been: exploration must show
        for a in range(A):
            greedy = (spec.obs_of(a).reshape(E, -1) @ spec.agents[a].model.weight).argmax(dim=1)
            greedy_differs += int((greedy != spec.actions[:, a].to(torch.int64)).sum())
    assert greedy_differs > 0.1 * 10 * A * E * 0.2          # (epsilon 0.2-0.45 x 3/4 of the uniform draws differ from the argmax)
    for a in range(A):
        ma, mb = eager.agents[a].model.memory, spec.agents[a].model.memory
        assert (ma.idx, ma.size) == (mb.idx, mb.size)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(ma, name), getattr(mb, name)), (a, name)
    eager.raise_on_status()
    spec.raise_on_status()


def test_choose_actions_is_the_choice_of_the_act_launch(torch_cuda):
    """``sgw_choose_actions`` row by row against ``sgw_act(SGW_ACT_QF32)``: NaN counts as the maximum, ties take the first index, epsilon 1
    always explores with the action SGW_STEP_RANDOM_ACTIONS would draw, epsilon 0 never, a list of rows picks (agent, env) pairs."""
    torch = torch_cuda
    from sorrel_amd.engine import GridEngine
    from sorrel_amd.spec import treasurehunt_spec

    E, A = 200, 5
    ws = treasurehunt_spec(12, 12, A, 2, spawn_prob=0.02, seed=3)
    eng = GridEngine(ws, E, device="cuda:0")
    eng.reset(epoch=0)
    g = torch.Generator().manual_seed(2)
    q = torch.randn((A * E, ws.num_actions), generator=g).cuda()
    q[5, 1] = float("nan")
    q[6] = 0.25                                                   # a tie: index 0
    q[7, 2] = q[7, 3] = 9.0
    eng.turn_set(2, 6)                                            # epoch 2, six turns completed: the turn in flight is 7
    eng.turn_epsilon(0.0)
    greedy = eng.choose_actions(q, None, 2, 7)
    ref = q.argmax(dim=1)
    ref[5], ref[6], ref[7] = 1, 0, 2
    assert torch.equal(greedy, ref)
    eng.turn_epsilon(1.0)
    eng.epoch, eng.turn = 2, 6
    eng.random_actions()                                          # actions[E][A] <- the draws of turn 7 (SGW_STREAM_ACTION)
    explored = eng.choose_actions(q, None, 2, 7)
    assert torch.equal(explored.view(A, E).t().contiguous(), eng.actions.to(torch.int64))
    eng.turn_epsilon(0.0)
    eng.turn_epsilon(1.0, 3)                                      # only agent 3 explores
    idx = torch.tensor([3 * E + 17, 0 * E + 17, 3 * E + 199, 4 * E + 0], dtype=torch.int64, device="cuda:0")
    picked = eng.choose_actions(q[idx].contiguous(), idx, 2, 7)
    want = torch.stack([eng.actions[17, 3].to(torch.int64), ref[17], eng.actions[199, 3].to(torch.int64), ref[4 * E]])
    assert torch.equal(picked, want)
    with pytest.raises(ValueError):
        eng.choose_actions(q[:, :2], None, 2, 7)
    eng.close()


@pytest.mark.parametrize("values", [False, True], ids=["int_actions", "action_values_eps"])
@pytest.mark.parametrize("which", ["tag", "cleanup", "tag_shared_two_models", "plain_80_agents"])
def test_generic_speculative_turn_equals_the_sequential_turn(torch_cuda, which, values):
    """``Environment.speculate_turns`` for agent rules the resolve kernel does not know (round 6, ``sgw_verify_rows``): Tag ("it" flag in the row,
    tags flip victims' flags and cells), Cleanup (beams on the layer above, facing, all-layer rewards, a 12-element positional tail) and plain
    movers beyond 64 agents -- one model shared by all agents (or two) with a shared replay ring.  World, agent state, step outputs and every
    replay row (tails included) equal the eager agent-after-agent loop's, turn after turn, through ring wrap-arounds and a reset; with action
    values and epsilon > 0 the exploration draws are the sequential turn's."""
    torch = torch_cuda
    from sorrel_amd.buffers import Buffer
    from sorrel_amd.models import BaseModel

    E = 21
    n_models = 2 if which == "tag_shared_two_models" else 1

    def make(speculate):
        made = []

        class Shared(BaseModel):
            def __init__(self, input_size, n_actions, cap):
                n = int(np.prod(input_size))
                super().__init__((n,), n_actions, memory_size=0, num_envs=E, device="cuda:0")
                self.memory = Buffer(capacity=cap, obs_shape=(n,), num_envs=E, device="cuda:0")
                self.weight = torch.randn((n, n_actions), generator=torch.Generator().manual_seed(3 + n + len(made))).cuda()
                self.epsilon = 0.3 if values else 0.0

            def take_action(self, state):
                q = state.reshape(state.shape[0], -1) @ self.weight
                return q if values else q.argmax(dim=1)

        count = [0]

        def factory(input_size, n_actions):
            A = agents_of[which]
            k = count[0] * n_models // A
            count[0] += 1
            while len(made) <= k:
                made.append(Shared(input_size, n_actions, 3 * A + 1))
            return made[k]

        if which.startswith("tag"):
            from sorrel_amd.entities import EmptyEntity
            from sorrel_amd.examples.tag.env import TagEnv
            from sorrel_amd.worlds import Gridworld

            cfg = {"agent": {"num_agents": 6, "vision_radius": 2, "reward_per_turn": 10}, "experiment": {"epochs": 1, "max_turns": 50}}
            env = TagEnv(Gridworld(8, 9, 1, EmptyEntity(), num_envs=E, device="cuda:0", seed=31), cfg, model_factory=factory)
        elif which == "cleanup":
            from tests.test_api_host import make_cleanup_env

            env = make_cleanup_env(E=E, seed=7, device="cuda:0", model_factory=factory)
        else:
            from tests.test_gpu_round2 import make_env

            env = make_env(26, 27, 80, 2, E, p=0.05, seed=9, model_factory=factory)
        env.speculate_turns = "always" if speculate else False
        return env, made

    agents_of = {"tag": 6, "tag_shared_two_models": 6, "cleanup": None, "plain_80_agents": 80}
    if which == "cleanup":
        from tests.test_api_host import make_cleanup_env

        agents_of["cleanup"] = len(make_cleanup_env(E=2, seed=7, device="cuda:0").agents)
    (a, ma), (b, mb) = make(False), make(True)
    A = len(a.agents)
    assert len(ma) == len(mb) == n_models
    passes = []
    for t in range(14):
        if t == 9:
            a.reset(); b.reset()
        for m in ma + mb:
            m.epsilon *= 0.95
        a.take_turn()
        b.take_turn()
        assert b.turn_plan()["loop"] == "speculative", b.turn_plan()
        passes.append(b.speculation_passes)
        torch.cuda.synchronize()
        names = ("grid", "agent_pos", "total_reward") + (("agent_state",) if which.startswith("tag") else ()) + (("agent_dir",) if which == "cleanup" else ())
        for name in names:
            assert torch.equal(getattr(a.world, name), getattr(b.world, name)), (t, name)
        assert torch.equal(a.rewards, b.rewards) and torch.equal(a.actions, b.actions), t
    assert b._spec_generic is True and max(passes) >= 2 and max(passes) <= A + 1
    for x, y in zip(ma, mb):
        assert (x.memory.idx, x.memory.size) == (y.memory.idx, y.memory.size)
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(x.memory, name), getattr(y.memory, name)), name
    assert float(b.world.total_reward.abs().sum()) > 0
    a.raise_on_status()
    b.raise_on_status()

"""TEST INFRASTRUCTURE ONLY -- generate ``tests/golden/*.npz`` from the reference.

Runs only in the build container (needs ``/root/reference``; see
``oracle/ref_loader.py``).  Usage::

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

What runs is the reference's *own* ``Environment.take_turn`` /
``Agent.transition`` / ``Gridworld.move`` / ``visual_field`` / ``shift`` /
``OneHotObservationSpec`` / ``Buffer`` code (sorrel/environment.py:81-93 and
everything under it).  Only the plugin points a Sorrel user is expected to
write are supplied here: an ``Entity.transition`` and a ``BaseModel.take_action``
that draw from the build's counter generator, and ``setup_agents`` /
``populate_environment``.  Each fixture stores inputs (spec, seeds, scripted
actions) and the reference's outputs (initial grid, per-turn float32
observations as the reference's replay ``Buffer`` stored them, actions,
rewards, ``world.total_reward``, grid and agent locations after every turn).

Fixtures are data only; no reference source text is stored.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

from oracle import gridstep_oracle as O
from oracle import ref_loader

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    ref_loader.install()
    import sorrel.action.action_spec as action_spec
    import sorrel.agents as agents
    import sorrel.entities as entities
    import sorrel.environment as environment
    import sorrel.examples.treasurehunt.agents as th_agents
    import sorrel.examples.treasurehunt.entities as th_entities
    import sorrel.examples.treasurehunt.world as th_world
    import sorrel.models.base_model as base_model
    import sorrel.observation.observation_spec as observation_spec
    import sorrel.worlds as worlds

    return dict(action_spec=action_spec, agents=agents, entities=entities, environment=environment,
                th_agents=th_agents, th_entities=th_entities, th_world=th_world,
                base_model=base_model, observation_spec=observation_spec, worlds=worlds)


class Ctx:
    """Harness state the plugin classes read (the reference passes only `world`)."""
    seed = 0
    env = 0
    epoch = 0
    turn = 0
    spec: O.Spec = None  # type: ignore
    scripted = None      # optional [turn][agent] action script


def build_plugins(R):
    """Plugin classes, written against the reference's public plugin API."""
    th = R["th_entities"]
    BaseModel = R["base_model"].BaseModel
    Entity = R["entities"].Entity

    class CounterEmpty(th.EmptyEntity):
        """Treasurehunt EmptyEntity whose two draws come from the counter RNG."""

        def __init__(self):
            super().__init__()
            self.kind = "EmptyEntity"

        def transition(self, world):
            sp = Ctx.spec
            y, x, z = self.location
            idx = int(O.cell_index(sp, y, x, z))
            u = int(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, O.STREAM_SPAWN, idx))
            if u < O.prob_threshold(world.spawn_prob):
                u2 = O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, O.STREAM_SPAWN_KIND, idx)
                k = int(O.categorical(u2, 3))
                ent = [th.Gem(world.values["gem"]), th.Food(world.values["food"]), th.Bone(world.values["bone"])][k]
                world.add(self.location, ent)

    class CounterModel(BaseModel):
        """take_action = counter RNG keyed by the agent's slot (or a script)."""

        def __init__(self, input_size, action_space, memory_size, slot):
            super().__init__(input_size, action_space, memory_size)
            self.slot = slot

        def take_action(self, state):
            if Ctx.scripted is not None:
                return int(Ctx.scripted[Ctx.turn - 1][self.slot])
            u = O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, O.STREAM_ACTION, self.slot)
            return int(O.categorical(u, self.action_space))

    class InertEmpty(Entity):
        """Passable, no transitions, appears as EmptyEntity (like Sand)."""

        def __init__(self):
            super().__init__()
            self.passable = True
            self.kind = "EmptyEntity"

    return CounterEmpty, CounterModel, InertEmpty


# --------------------------------------------------------------------------- #
# harness environments (user-side subclasses of the reference Environment)
# --------------------------------------------------------------------------- #
def make_treasurehunt_env(R, spec: O.Spec, turns: int, actions_names=("up", "down", "left", "right"),
                          entity_map_override=None, rgb=False):
    CounterEmpty, CounterModel, _ = build_plugins(R)
    Environment = R["environment"].Environment
    th, tha, thw = R["th_entities"], R["th_agents"], R["th_world"]
    OneHot = R["observation_spec"].RGBObservationSpec if rgb else R["observation_spec"].OneHotObservationSpec
    ActionSpec = R["action_spec"].ActionSpec
    entity_list = ["EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]

    class HarnessEnv(Environment):
        def setup_agents(self):
            agents = []
            for slot in range(spec.num_agents):
                ospec = OneHot(entity_list, full_view=False, vision_radius=spec.vision_radius)
                if entity_map_override is not None:
                    ospec.override_entity_map(entity_map_override)
                ospec.override_input_size((int(np.prod(ospec.input_size)),))
                aspec = ActionSpec(list(actions_names))
                model = CounterModel(ospec.input_size, aspec.n_actions, memory_size=turns + 1, slot=slot)
                agents.append(tha.TreasurehuntAgent(observation_spec=ospec, action_spec=aspec, model=model))
            self.agents = agents

        def populate_environment(self):
            H, W = self.world.height, self.world.width
            for index in np.ndindex(self.world.map.shape):
                y, x, z = index
                if (y in [0, H - 1] or x in [0, W - 1]) and z == 1:
                    self.world.add(index, th.Wall())
                elif z == 0:
                    self.world.add(index, th.Sand())
            # optional pre-seeding ("dense entities"), same draws as oracle reset_env
            if spec.dense_prob > 0.0:
                for y in range(1, H - 1):
                    for x in range(1, W - 1):
                        idx = int(O.cell_index(spec, y, x, 1))
                        if int(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, 0, O.STREAM_DENSE, idx)) < O.prob_threshold(spec.dense_prob):
                            k = int(O.categorical(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, 0, O.STREAM_DENSE_KIND, idx), 3))
                            v = self.world.values
                            self.world.add((y, x, 1), [th.Gem(v["gem"]), th.Food(v["food"]), th.Bone(v["bone"])][k])
            pos = O.place_agents(spec, Ctx.env, Ctx.epoch)
            for (y, x), agent in zip(pos, self.agents):
                self.world.add((int(y), int(x), 1), agent)

    cfg = {"world": {"height": spec.height, "width": spec.width,
                     "gem_value": spec.type_value[3], "food_value": spec.type_value[5],
                     "bone_value": spec.type_value[4], "spawn_prob": spec.spawn_prob[1]},
           "experiment": {"epochs": 1, "max_turns": turns, "record_period": 1}}
    world = thw.TreasurehuntWorld(config=cfg, default_entity=CounterEmpty())
    return HarnessEnv(world, cfg), CounterEmpty


def type_ids_treasurehunt(R, world, CounterEmpty) -> np.ndarray:
    th = R["th_entities"]
    Agent = R["agents"].Agent
    H, W, L = world.map.shape
    g = np.zeros((L, H, W), dtype=np.uint8)
    for (y, x, z), e in np.ndenumerate(world.map):
        if isinstance(e, Agent):
            t = 6
        elif type(e) is th.Sand:
            t = 0
        elif type(e) is CounterEmpty or type(e) is th.EmptyEntity:
            t = 1
        elif type(e) is th.Wall:
            t = 2
        elif type(e) is th.Gem:
            t = 3
        elif type(e) is th.Bone:
            t = 4
        elif type(e) is th.Food:
            t = 5
        else:
            raise RuntimeError(f"unmapped entity {e!r}")
        assert tuple(e.location) == (y, x, z), "entity.location out of sync with the map"
        g[z, y, x] = t
    return g


def run_reference_treasurehunt(R, spec: O.Spec, env_ids, turns, scripted=None, epoch=0,
                               actions_names=("up", "down", "left", "right"), entity_map_override=None, rgb=False):
    E, A, C, V = len(env_ids), spec.num_agents, spec.num_channels, spec.window
    out = dict(
        grid0=np.zeros((E, spec.layers, spec.height, spec.width), np.uint8),
        pos0=np.zeros((E, A, 2), np.uint8),
        obs=np.zeros((turns, E, A, C, V, V), np.float32),
        actions=np.zeros((turns, E, A), np.uint8),
        rewards=np.zeros((turns, E, A), np.float32),
        dones=np.zeros((turns, E, A), np.float32),
        total_reward=np.zeros((turns, E), np.float64),
        grid=np.zeros((turns, E, spec.layers, spec.height, spec.width), np.uint8),
        pos=np.zeros((turns, E, A, 2), np.uint8),
    )
    for n, env_id in enumerate(env_ids):
        Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, Ctx.spec = spec.seed, int(env_id), epoch, 0, spec
        Ctx.scripted = None if scripted is None else scripted[:, n]
        env, CounterEmpty = make_treasurehunt_env(R, spec, turns, actions_names, entity_map_override, rgb=rgb)
        out["grid0"][n] = type_ids_treasurehunt(R, env.world, CounterEmpty)
        out["pos0"][n] = [a.location[:2] for a in env.agents]
        for t in range(turns):
            Ctx.turn = env.turn + 1                     # Environment.turn after its increment
            env.take_turn()                             # <- the reference's hot path
            assert env.turn == Ctx.turn
            for a, agent in enumerate(env.agents):
                mem = agent.model.memory
                out["obs"][t, n, a] = mem.states[t].reshape(C, V, V)
                out["actions"][t, n, a] = mem.actions[t]
                out["rewards"][t, n, a] = mem.rewards[t]
                out["dones"][t, n, a] = mem.dones[t]
                out["pos"][t, n, a] = agent.location[:2]
                assert agent.location[2] == 1
            out["total_reward"][t, n] = env.world.total_reward
            out["grid"][t, n] = type_ids_treasurehunt(R, env.world, CounterEmpty)
    return out


# --------------------------------------------------------------------------- #
# a second family: basic entities, arbitrary layers, walls on every layer
# --------------------------------------------------------------------------- #
def basic_spec(height, width, layers, num_agents, vision_radius, seed, dense_prob, actions_names,
               appearance=None) -> O.Spec:
    """Types: 0 InertEmpty (kind EmptyEntity), 1 Wall, 2 Gem(3.5), 3 Gem(-2), 4 MovingAgent-subclass 'Walker'.
    Border walls on EVERY layer (the iowa layout, examples/iowa/env.py:105-107)."""
    C = 4  # entity_list = ["EmptyEntity", "Wall", "Gem", "Walker"]
    if appearance is None:
        appearance = np.zeros((5, C))
        appearance[1, 1] = 1.0
        appearance[2, 2] = 1.0
        appearance[3, 2] = 1.0
        appearance[4, 3] = 1.0
    moves = {"up": (-1, 0), "down": (1, 0), "left": (0, -1), "right": (0, 1)}
    return O.Spec(
        height=height, width=width, layers=layers, num_agents=num_agents, vision_radius=vision_radius,
        num_types=5, num_channels=C, agent_layer=layers - 1, default_type=0, fill_type=1,
        action_dy=[moves.get(n, (0, 0))[0] for n in actions_names],
        action_dx=[moves.get(n, (0, 0))[1] for n in actions_names],
        agent_type=[4] * num_agents, type_value=[0, -1, 3.5, -2, 0], type_passable=[1, 0, 1, 1, 0],
        type_rule=[0] * 5, spawn_prob=[0.0] * 5, spawn_choices=[[]] * 5, appearance=appearance, seed=seed,
        layer_fill_type=[0] * layers, layer_border_type=[1] * layers, dense_prob=dense_prob, dense_choices=[2, 3],
    )


def run_reference_basic(R, spec: O.Spec, env_ids, turns, actions_names, entity_map_override=None, epoch=0):
    _, CounterModel, InertEmpty = build_plugins(R)
    Environment = R["environment"].Environment
    Gridworld = R["worlds"].Gridworld
    ent = R["entities"]
    MovingAgent = R["agents"].MovingAgent
    OneHot = R["observation_spec"].OneHotObservationSpec
    ActionSpec = R["action_spec"].ActionSpec
    entity_list = ["EmptyEntity", "Wall", "Gem", "Walker"]
    L, zA = spec.layers, spec.agent_layer

    class Walker(MovingAgent):
        """Concrete MovingAgent: the abstract hooks only (agents/agent.py:57-111)."""

        def reset(self):
            pass

        def pov(self, world):
            return self.observation_spec.observe(world, self.location).reshape(1, -1)

        def get_action(self, state):
            return self.model.take_action(state)

        def is_done(self, world):
            return world.is_done

    class BasicEnv(Environment):
        def setup_agents(self):
            self.agents = []
            for slot in range(spec.num_agents):
                ospec = OneHot(entity_list, full_view=False, vision_radius=spec.vision_radius)
                if entity_map_override is not None:
                    ospec.override_entity_map(entity_map_override)
                ospec.override_input_size((int(np.prod(ospec.input_size)),))
                aspec = ActionSpec(list(actions_names))
                model = CounterModel(ospec.input_size, aspec.n_actions, memory_size=turns + 1, slot=slot)
                self.agents.append(Walker(ospec, aspec, model))

        def populate_environment(self):
            H, W = self.world.height, self.world.width
            for index in np.ndindex(self.world.map.shape):
                y, x, z = index
                if y in [0, H - 1] or x in [0, W - 1]:
                    self.world.add(index, ent.Wall())
            if spec.dense_prob > 0.0:
                for y in range(1, H - 1):
                    for x in range(1, W - 1):
                        idx = int(O.cell_index(spec, y, x, zA))
                        if int(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, 0, O.STREAM_DENSE, idx)) < O.prob_threshold(spec.dense_prob):
                            k = int(O.categorical(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, 0, O.STREAM_DENSE_KIND, idx), 2))
                            self.world.add((y, x, zA), ent.Gem([3.5, -2][k]))
            pos = O.place_agents(spec, Ctx.env, Ctx.epoch)
            for (y, x), agent in zip(pos, self.agents):
                self.world.add((int(y), int(x), zA), agent)

    def type_ids(world):
        H, W, Ls = world.map.shape
        g = np.zeros((Ls, H, W), dtype=np.uint8)
        for (y, x, z), e in np.ndenumerate(world.map):
            if isinstance(e, Walker):
                t = 4
            elif type(e) is InertEmpty:
                t = 0
            elif type(e) is ent.Wall:
                t = 1
            elif type(e) is ent.Gem:
                t = 2 if e.value == 3.5 else 3
            else:
                raise RuntimeError(f"unmapped entity {e!r}")
            g[z, y, x] = t
        return g

    E, A, C, V = len(env_ids), spec.num_agents, spec.num_channels, spec.window
    out = dict(
        grid0=np.zeros((E, L, spec.height, spec.width), np.uint8), pos0=np.zeros((E, A, 2), np.uint8),
        obs=np.zeros((turns, E, A, C, V, V), np.float32), actions=np.zeros((turns, E, A), np.uint8),
        rewards=np.zeros((turns, E, A), np.float32), dones=np.zeros((turns, E, A), np.float32),
        total_reward=np.zeros((turns, E), np.float64),
        grid=np.zeros((turns, E, L, spec.height, spec.width), np.uint8), pos=np.zeros((turns, E, A, 2), np.uint8),
    )
    cfg = {"experiment": {"epochs": 1, "max_turns": turns, "record_period": 1}}
    for n, env_id in enumerate(env_ids):
        Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, Ctx.spec, Ctx.scripted = spec.seed, int(env_id), epoch, 0, spec, None
        env = BasicEnv(Gridworld(spec.height, spec.width, L, InertEmpty()), cfg)
        out["grid0"][n] = type_ids(env.world)
        out["pos0"][n] = [a.location[:2] for a in env.agents]
        for t in range(turns):
            Ctx.turn = env.turn + 1
            env.take_turn()
            for a, agent in enumerate(env.agents):
                mem = agent.model.memory
                out["obs"][t, n, a] = mem.states[t].reshape(C, V, V)
                out["actions"][t, n, a] = mem.actions[t]
                out["rewards"][t, n, a] = mem.rewards[t]
                out["dones"][t, n, a] = mem.dones[t]
                out["pos"][t, n, a] = agent.location[:2]
            out["total_reward"][t, n] = env.world.total_reward
            out["grid"][t, n] = type_ids(env.world)
    return out


# --------------------------------------------------------------------------- #
# Tag: the first agent <-> agent interaction (sorrel/examples/tag)
# --------------------------------------------------------------------------- #
def tag_spec(height, width, num_agents, vision_radius, seed, reward_per_turn=10) -> O.Spec:
    """Types: 0 EmptyEntity (basic, passable, inert), 1 Wall, 2 TagAgent that is it (kind "It"),
    3 TagAgent that is not (kind "NotIt").  entity_list of examples/tag/env.py:40."""
    app = np.zeros((4, 4))
    app[1, 1] = app[2, 2] = app[3, 3] = 1.0
    return O.Spec(
        height=height, width=width, layers=1, num_agents=num_agents, vision_radius=vision_radius,
        num_types=4, num_channels=4, agent_layer=0, default_type=0, fill_type=1,
        action_dy=[-1, 1, 0, 0], action_dx=[0, 0, -1, 1], agent_type=[3] * num_agents,
        type_value=[0, -1, 0, 0], type_passable=[1, 0, 0, 0], type_rule=[0] * 4, spawn_prob=[0.0] * 4,
        spawn_choices=[[]] * 4, appearance=app, seed=seed, layer_fill_type=[0], layer_border_type=[1],
        agent_rule=O.AGENT_RULE_TAG, tag_it_type=2, tag_notit_type=3, tag_reward=reward_per_turn,
    )


def run_reference_tag(R, spec: O.Spec, env_ids, turns, epoch=0):
    """The reference's own TagAgent (pov with the it flag, act with move + tagging) and step loop."""
    _, CounterModel, _ = build_plugins(R)
    import sorrel.examples.tag.agents as tag_agents

    Environment = R["environment"].Environment
    Gridworld = R["worlds"].Gridworld
    ent = R["entities"]
    OneHot = R["observation_spec"].OneHotObservationSpec
    ActionSpec = R["action_spec"].ActionSpec
    TagAgent = tag_agents.TagAgent
    entity_list = ["EmptyEntity", "Wall", "It", "NotIt"]

    class TagHarness(Environment):
        def setup_agents(self):
            agents = []
            for slot in range(spec.num_agents):
                ospec = OneHot(entity_list, full_view=False, vision_radius=spec.vision_radius)
                n = int(np.prod(ospec.input_size)) + 1          # + the it flag (tag/env.py:47-48)
                ospec.override_input_size((n,))
                aspec = ActionSpec(["up", "down", "left", "right"])
                model = CounterModel(ospec.input_size, aspec.n_actions, memory_size=turns + 1, slot=slot)
                agents.append(TagAgent(ospec, aspec, model, reward_per_turn=spec.tag_reward))
            state0 = O.init_agent_state(spec, Ctx.env)          # stands in for np.random.choice (env.py:66-69)
            for a, agent in enumerate(agents):
                if state0[a] == spec.tag_it_type:
                    agent.it = True
            self.agents = agents

        def populate_environment(self):
            H, W = self.world.height, self.world.width
            for index in np.ndindex(self.world.map.shape):
                y, x, z = index
                if y in [0, H - 1] or x in [0, W - 1]:
                    self.world.add(index, ent.Wall())
            pos = O.place_agents(spec, Ctx.env, Ctx.epoch)
            for (y, x), agent in zip(pos, self.agents):
                self.world.add((int(y), int(x), 0), agent)

    def type_ids(world):
        H, W, Ls = world.map.shape
        g = np.zeros((Ls, H, W), dtype=np.uint8)
        for (y, x, z), e in np.ndenumerate(world.map):
            if isinstance(e, TagAgent):
                assert e.kind == ("It" if e.it else "NotIt")
                t = 2 if e.it else 3
            elif type(e) is ent.EmptyEntity:
                t = 0
            elif type(e) is ent.Wall:
                t = 1
            else:
                raise RuntimeError(f"unmapped entity {e!r}")
            g[z, y, x] = t
        return g

    E, A, C, V = len(env_ids), spec.num_agents, spec.num_channels, spec.window
    n = C * V * V
    out = dict(
        grid0=np.zeros((E, 1, spec.height, spec.width), np.uint8), pos0=np.zeros((E, A, 2), np.uint8),
        obs=np.zeros((turns, E, A, C, V, V), np.float32), actions=np.zeros((turns, E, A), np.uint8),
        rewards=np.zeros((turns, E, A), np.float32), dones=np.zeros((turns, E, A), np.float32),
        total_reward=np.zeros((turns, E), np.float64),
        grid=np.zeros((turns, E, 1, spec.height, spec.width), np.uint8), pos=np.zeros((turns, E, A, 2), np.uint8),
        state_at_pov=np.zeros((turns, E, A), np.uint8), agent_state=np.zeros((turns, E, A), np.uint8),
    )
    cfg = {"experiment": {"epochs": 1, "max_turns": turns, "record_period": 1}}
    for k, env_id in enumerate(env_ids):
        Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, Ctx.spec, Ctx.scripted = spec.seed, int(env_id), epoch, 0, spec, None
        env = TagHarness(Gridworld(spec.height, spec.width, 1, ent.EmptyEntity()), cfg)
        out["grid0"][k] = type_ids(env.world)
        out["pos0"][k] = [a.location[:2] for a in env.agents]
        for t in range(turns):
            Ctx.turn = env.turn + 1
            env.take_turn()
            for a, agent in enumerate(env.agents):
                mem = agent.model.memory
                out["obs"][t, k, a] = mem.states[t][:n].reshape(C, V, V)
                out["state_at_pov"][t, k, a] = 2 if mem.states[t][n] == 1.0 else 3      # the flag pov() appended
                out["actions"][t, k, a] = mem.actions[t]
                out["rewards"][t, k, a] = mem.rewards[t]
                out["dones"][t, k, a] = mem.dones[t]
                out["pos"][t, k, a] = agent.location[:2]
                out["agent_state"][t, k, a] = 2 if agent.it else 3
            out["total_reward"][t, k] = env.world.total_reward
            out["grid"][t, k] = type_ids(env.world)
    return out


# --------------------------------------------------------------------------- #
# Cleanup: cross-layer conditional transitions, timers, beams (sorrel/examples/cleanup)
# --------------------------------------------------------------------------- #
CLEANUP_KINDS = ["EmptyEntity", "Wall", "River", "Pollution", "AppleTree", "Apple", "CleanBeam", "ZapBeam", "CleanupAgent"]


def cleanup_spec(height, width, num_agents, vision_radius, seed, beam_radius=3, pollution_p=0.009, apple_p=0.002) -> O.Spec:
    """Types: 0 EmptyEntity, 1 Sand (kind EmptyEntity), 2 Wall, 3 River, 4 Pollution, 5 AppleTree, 6 Apple,
    7/8 CleanBeam fresh/aged, 9/10 ZapBeam fresh/aged, 11 CleanupAgent.  Layers: 0 objects, 1 agents, 2 beams."""
    chan = [0, 0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8]
    app = np.zeros((12, 9))
    for t, c in enumerate(chan):
        if c != 0:
            app[t, c] = 1.0
    beams = (1 << 7) | (1 << 8)
    names = ["up", "down", "left", "right", "clean", "zap"]
    return O.Spec(
        height=height, width=width, layers=3, num_agents=num_agents, vision_radius=vision_radius,
        num_types=12, num_channels=9, agent_layer=1, default_type=0, fill_type=2,
        action_dy=[-1, 1, 0, 0, 0, 0], action_dx=[0, 0, -1, 1, 0, 0], agent_type=[11] * num_agents,
        type_value=[0, 0, 0, 0, 0, 0, 1, 0, 0, -1, -1, 0], type_passable=[1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
        type_rule=[0, 0, 0, O.RULE_SPAWN, O.RULE_BECOME_IF, O.RULE_SPAWN, O.RULE_BECOME_IF,
                   O.RULE_BECOME_IF, O.RULE_BECOME_IF, O.RULE_BECOME_IF, O.RULE_BECOME_IF, 0],
        spawn_prob=[0, 0, 0, pollution_p, 0, apple_p, 0, 0, 0, 0, 0, 0],
        spawn_choices=[[], [], [], [4], [], [6], [], [], [], [], [], []],
        rule_layer=[0, 0, 0, 0, 2, 0, 1, -1, -1, -1, -1, 0],
        rule_mask=[0, 0, 0, 0, beams, 0, 1 << 11, 0, 0, 0, 0, 0],
        rule_become=[0, 0, 0, 0, 3, 0, 5, 8, 0, 10, 0, 0],
        appearance=app, seed=seed, layer_fill_type=[0, 0, 0], layer_border_type=[2, 2, 2],
        agent_rule=O.AGENT_RULE_CLEANUP, action_kind=[0, 0, 0, 0, O.ACTION_CLEAN, O.ACTION_ZAP], beam_radius=beam_radius,
        clean_beam_type=7, zap_beam_type=9, beam_block_mask=1 << 2, reward_total_factor=2,
    )


def run_reference_cleanup(R, spec: O.Spec, env_ids, turns, initial_apples=6, epoch=0):
    """The reference's own CleanupAgent / CleanupObservation / Pollution / Apple / Beam classes and step loop.
    Plugins: River / AppleTree draw from the counter RNG; populate_environment places apples and agents with it."""
    _, CounterModel, _ = build_plugins(R)
    import sorrel.examples.cleanup.agents as ca
    import sorrel.examples.cleanup.entities as ce
    import sorrel.examples.cleanup.world as cw

    Environment = R["environment"].Environment
    ActionSpec = R["action_spec"].ActionSpec
    OrigRiver, OrigTree = ce.River, ce.AppleTree

    class CounterRiver(OrigRiver):
        def __init__(self):
            super().__init__()
            self.kind = "River"

        def transition(self, world):
            y, x, z = self.location
            idx = int(O.cell_index(spec, y, x, z))
            if int(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, O.STREAM_SPAWN, idx)) < O.prob_threshold(world.pollution_spawn_chance):
                world.add(self.location, ce.Pollution())

    class CounterTree(OrigTree):
        def __init__(self):
            super().__init__()
            self.kind = "AppleTree"

        def transition(self, world):
            if not world.pollution > world.pollution_threshold:
                y, x, z = self.location
                idx = int(O.cell_index(spec, y, x, z))
                if int(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, O.STREAM_SPAWN, idx)) < O.prob_threshold(world.apple_spawn_chance):
                    world.add(self.location, ce.Apple())

    # Pollution.transition / Apple.transition create `River()` / `AppleTree()` through their module globals
    ce.River, ce.AppleTree = CounterRiver, CounterTree

    class Harness(Environment):
        def setup_agents(self):
            self.agents = []
            for slot in range(spec.num_agents):
                ospec = ca.CleanupObservation(entity_list=CLEANUP_KINDS, vision_radius=spec.vision_radius)
                aspec = ActionSpec(["up", "down", "left", "right", "clean", "zap"])
                flat = (int(np.prod(ospec.input_size)),)        # replay rows are flat vectors (input_size is (1, n))
                model = CounterModel(flat, aspec.n_actions, memory_size=turns + 1, slot=slot)
                self.agents.append(ca.CleanupAgent(observation_spec=ospec, action_spec=aspec, model=model))

        def populate_environment(self):
            # the layout rules of examples/cleanup/env.py:84-117 (mode DEFAULT), random picks from the counter RNG
            w = self.world
            spawn_points, apple_points = [], []
            for index in np.ndindex(w.map.shape):
                H, W, L = index
                if H in [0, w.height - 1] or W in [0, w.width - 1]:
                    w.add(index, ce.Wall())
                elif L == 0:
                    if (0 < H < (w.height // 3)) or (H < ((w.height // 3) * 2 - 1) and W in [w.width // 3, 1 + w.width // 3]):
                        w.add(index, CounterRiver())
                    elif (w.height - 1 - (w.height // 3)) < H < (w.height - 1):
                        w.add(index, CounterTree())
                        apple_points.append(index)
                    else:
                        w.add(index, ce.Sand())
                        spawn_points.append((index[0], index[1], w.agent_layer))

            def pick(n_from, k, stream):
                u = O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, 0, stream, np.arange(k))
                taken, out = [], []
                for i in range(k):
                    d = int((int(u[i]) * (n_from - i)) >> 32)
                    for t in taken:
                        if d >= t:
                            d += 1
                    taken.append(d)
                    taken.sort()
                    out.append(d)
                return out

            for i in pick(len(apple_points), w.initial_apples, O.STREAM_DENSE):
                w.add(tuple(apple_points[i]), ce.Apple())
            for i, agent in zip(pick(len(spawn_points), len(self.agents), O.STREAM_PLACE), self.agents):
                w.add(tuple(spawn_points[i]), agent)

    cfg = ref_loader.DictConfig({
        "experiment": {"epochs": 1, "max_turns": turns, "record_period": 1},
        "env": {"height": spec.height, "width": spec.width, "layers": 3, "full_mdp": False, "pollution_threshold": 0.5,
                "initial_apples": initial_apples, "apple_spawn_chance": spec.spawn_prob[5],
                "pollution_spawn_chance": spec.spawn_prob[3], "mode": "DEFAULT"},
        "agent": {"agent": {"num": spec.num_agents, "beam_radius": spec.beam_radius,
                            "obs": {"channels": 8, "vision": spec.vision_radius, "n_frames": 1, "embeddings": 3}}},
    })

    def type_ids(world):
        H, W, Ls = world.map.shape
        g = np.zeros((Ls, H, W), dtype=np.uint8)
        for (y, x, z), e in np.ndenumerate(world.map):
            name = type(e).__name__
            if isinstance(e, ca.CleanupAgent):
                t = 11
            elif isinstance(e, ca.CleanBeam):
                t = 7 + (1 if e.turn_counter >= 1 else 0)
            elif isinstance(e, ca.ZapBeam):
                t = 9 + (1 if e.turn_counter >= 1 else 0)
            else:
                t = {"EmptyEntity": 0, "Sand": 1, "Wall": 2, "CounterRiver": 3, "Pollution": 4, "CounterTree": 5, "Apple": 6}[name]
            g[z, y, x] = t
        return g

    E, A, C, V = len(env_ids), spec.num_agents, spec.num_channels, spec.window
    n = C * V * V
    out = dict(
        grid0=np.zeros((E, 3, spec.height, spec.width), np.uint8), pos0=np.zeros((E, A, 2), np.uint8),
        obs=np.zeros((turns, E, A, C, V, V), np.float32), pos_code=np.zeros((turns, E, A, 12), np.float32),
        actions=np.zeros((turns, E, A), np.uint8), rewards=np.zeros((turns, E, A), np.float32),
        dones=np.zeros((turns, E, A), np.float32), total_reward=np.zeros((turns, E), np.float64),
        grid=np.zeros((turns, E, 3, spec.height, spec.width), np.uint8), pos=np.zeros((turns, E, A, 2), np.uint8),
        agent_dir=np.zeros((turns, E, A), np.uint8),
    )
    try:
        for k, env_id in enumerate(env_ids):
            Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, Ctx.spec, Ctx.scripted = spec.seed, int(env_id), epoch, 0, spec, None
            env = Harness(cw.CleanupWorld(cfg, ce.EmptyEntity()), cfg)
            out["grid0"][k] = type_ids(env.world)
            out["pos0"][k] = [a.location[:2] for a in env.agents]
            for t in range(turns):
                Ctx.turn = env.turn + 1
                env.take_turn()
                for a, agent in enumerate(env.agents):
                    mem = agent.model.memory
                    st = np.asarray(mem.states[t]).reshape(-1)
                    out["obs"][t, k, a] = st[:n].reshape(C, V, V)
                    out["pos_code"][t, k, a] = st[n:]
                    out["actions"][t, k, a] = mem.actions[t]
                    out["rewards"][t, k, a] = mem.rewards[t]
                    out["dones"][t, k, a] = mem.dones[t]
                    out["pos"][t, k, a] = agent.location[:2]
                    out["agent_dir"][t, k, a] = agent.direction
                out["total_reward"][t, k] = env.world.total_reward
                out["grid"][t, k] = type_ids(env.world)
    finally:
        ce.River, ce.AppleTree = OrigRiver, OrigTree
    return out


# --------------------------------------------------------------------------- #
# stock Treasurehunt with the reference's own global np.random stream
# --------------------------------------------------------------------------- #
def run_reference_stock(R, height, width, num_agents, radius, spawn_prob, turns, np_seed):
    """Unmodified Treasurehunt entities/agents/world + RandomModel; only
    setup_agents is replaced (the stock one builds a PyTorch IQN)."""
    Environment = R["environment"].Environment
    th, tha, thw = R["th_entities"], R["th_agents"], R["th_world"]
    RandomModel = R["base_model"].RandomModel
    OneHot = R["observation_spec"].OneHotObservationSpec
    ActionSpec = R["action_spec"].ActionSpec
    import sorrel.examples.treasurehunt.env as th_env_mod
    entity_list = ["EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]

    class StockEnv(th_env_mod.TreasurehuntEnv):
        def setup_agents(self):
            self.agents = []
            for _ in range(num_agents):
                ospec = OneHot(entity_list, full_view=False, vision_radius=radius)
                ospec.override_input_size((int(np.prod(ospec.input_size)),))
                aspec = ActionSpec(["up", "down", "left", "right"])
                model = RandomModel(ospec.input_size, aspec.n_actions, memory_size=turns + 1)
                self.agents.append(tha.TreasurehuntAgent(ospec, aspec, model))

    cfg = {"world": {"height": height, "width": width, "gem_value": 10, "food_value": 5,
                     "bone_value": -10, "spawn_prob": spawn_prob},
           "model": {"agent_vision_radius": radius, "num_agents": num_agents},
           "experiment": {"epochs": 1, "max_turns": turns, "record_period": 1}}
    np.random.seed(np_seed)
    world = thw.TreasurehuntWorld(config=cfg, default_entity=th.EmptyEntity())
    env = StockEnv(world, cfg)
    A, C, V = num_agents, 6, 2 * radius + 1
    out = dict(
        grid0=type_ids_treasurehunt(R, env.world, th.EmptyEntity)[None],
        pos0=np.array([[a.location[:2] for a in env.agents]], np.uint8),
        obs=np.zeros((turns, 1, A, C, V, V), np.float32), actions=np.zeros((turns, 1, A), np.uint8),
        rewards=np.zeros((turns, 1, A), np.float32), total_reward=np.zeros((turns, 1), np.float64),
        grid=np.zeros((turns, 1, 2, height, width), np.uint8), pos=np.zeros((turns, 1, A, 2), np.uint8),
    )
    for t in range(turns):
        env.take_turn()
        for a, agent in enumerate(env.agents):
            mem = agent.model.memory
            out["obs"][t, 0, a] = mem.states[t].reshape(C, V, V)
            out["actions"][t, 0, a] = mem.actions[t]
            out["rewards"][t, 0, a] = mem.rewards[t]
            out["pos"][t, 0, a] = agent.location[:2]
        out["total_reward"][t, 0] = env.world.total_reward
        out["grid"][t, 0] = type_ids_treasurehunt(R, env.world, th.EmptyEntity)
    return out


# --------------------------------------------------------------------------- #
def spec_to_json(spec: O.Spec) -> str:
    d = {k: getattr(spec, k) for k in spec.__dataclass_fields__ if k != "appearance"}
    d = {k: (list(map(lambda v: list(v) if isinstance(v, (list, tuple)) else v, v)) if isinstance(v, (list, tuple)) else v)
         for k, v in d.items()}
    d["appearance"] = np.asarray(spec.appearance, dtype=np.float64).tolist()
    return json.dumps(d)


def save(name, spec, env_ids, ref, extra=None):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    payload = dict(ref)
    payload["spec_json"] = np.array(spec_to_json(spec))
    payload["env_ids"] = np.asarray(env_ids, dtype=np.int64)
    if extra:
        payload.update(extra)
    np.savez_compressed(path, **payload)
    print(f"  wrote {path} ({os.path.getsize(path)} bytes)")


def check_against_oracle(spec, env_ids, turns, ref, scripted=None, epoch=0, injected=False):
    mine = O.rollout(spec, env_ids, turns, epoch=epoch, actions=scripted,
                     initial=(ref["grid0"], ref["pos0"]) if injected else None)
    keys = ["grid0", "pos0", "obs", "actions", "rewards", "total_reward", "grid", "pos"]
    if "state_at_pov" in ref:
        keys += ["state_at_pov", "agent_state"]
    if "agent_dir" in ref:
        keys += ["agent_dir"]
    for k in keys:
        if not np.array_equal(mine[k], ref[k]):
            bad = np.argwhere(mine[k] != ref[k])[0]
            raise AssertionError(f"restatement differs from the reference in {k} at {bad}")
    assert not ref["dones"].any(), "reference stored done != 0 inside an epoch"


def main() -> int:
    if not ref_loader.reference_available():
        print("reference not available; fixtures cannot be regenerated here", file=sys.stderr)
        return 1
    R = _import_reference()
    t0 = time.time()
    moves = ("up", "down", "left", "right")

    print("c1_treasurehunt_10x10: BASELINE config 1 (10x10, 2 agents, r=2, 100 turns)")
    spec = O.treasurehunt_spec(10, 10, 2, 2, spawn_prob=0.005, seed=0)
    ref = run_reference_treasurehunt(R, spec, [0], 100)
    check_against_oracle(spec, [0], 100, ref)
    save("c1_treasurehunt_10x10", spec, [0], ref)

    print("c2_treasurehunt_16x16: config-2 shape, 4 envs x 16 turns")
    spec = O.treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.02, seed=1)
    ids = [0, 1, 2, 4095]
    ref = run_reference_treasurehunt(R, spec, ids, 16)
    check_against_oracle(spec, ids, 16, ref)
    save("c2_treasurehunt_16x16", spec, ids, ref)

    print("c3_treasurehunt_32x32: headline shape, 2 envs x 8 turns")
    spec = O.treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=0)
    ids = [0, 65535]
    ref = run_reference_treasurehunt(R, spec, ids, 8)
    check_against_oracle(spec, ids, 8, ref)
    save("c3_treasurehunt_32x32", spec, ids, ref)

    print("crowded_6x6: 6 agents on a 4x4 interior, p=0.2 (contention, bumps, pickups)")
    spec = O.treasurehunt_spec(6, 6, 6, 2, spawn_prob=0.2, seed=11)
    ids = [3, 9]
    ref = run_reference_treasurehunt(R, spec, ids, 40)
    check_against_oracle(spec, ids, 40, ref)
    save("crowded_6x6", spec, ids, ref)

    print("ragged_9x13_rmax: non-square odd grid, r = (min-1)//2 = 4, dense pre-seeding, epoch 3")
    spec = O.treasurehunt_spec(9, 13, 3, 4, spawn_prob=0.03, seed=5, dense_prob=0.25)
    ids = [7]
    ref = run_reference_treasurehunt(R, spec, ids, 25, epoch=3)
    check_against_oracle(spec, ids, 25, ref, epoch=3)
    save("ragged_9x13_rmax", spec, ids, ref, extra={"epoch": np.array(3)})

    print("c5_small_dense: config-5 flavour scaled down (24x24, 12 agents, r=5, dense)")
    spec = O.treasurehunt_spec(24, 24, 12, 5, spawn_prob=0.05, seed=2, dense_prob=0.25)
    ids = [0, 16383]
    ref = run_reference_treasurehunt(R, spec, ids, 6)
    check_against_oracle(spec, ids, 6, ref)
    save("c5_small_dense", spec, ids, ref)

    print("scripted_noop: 5 actions incl. a non-move name -> stay in place, reward = own value")
    names = ("up", "down", "left", "right", "noop")
    spec = O.treasurehunt_spec(8, 8, 2, 2, spawn_prob=0.1, seed=4)
    spec.action_dy, spec.action_dx = [-1, 1, 0, 0, 0], [0, 0, -1, 1, 0]
    rng = np.random.default_rng(0)
    scripted = rng.integers(0, 5, size=(20, 1, 2))
    scripted[3] = 4
    ref = run_reference_treasurehunt(R, spec, [1], 20, scripted=scripted, actions_names=names)
    check_against_oracle(spec, [1], 20, ref, scripted=scripted)
    save("scripted_noop", spec, [1], ref, extra={"scripted": scripted.astype(np.uint8)})

    print("basic_doublewall: basic entities, 2 layers, walls on both (layer sum = 2.0), float values")
    spec = basic_spec(11, 8, 2, 3, 3, seed=8, dense_prob=0.3, actions_names=moves)
    ids = [0, 5]
    ref = run_reference_basic(R, spec, ids, 25, moves)
    check_against_oracle(spec, ids, 25, ref)
    assert ref["obs"].max() == 2.0
    save("basic_doublewall", spec, ids, ref)

    print("basic_1layer: single-layer world (agents and items share the only layer)")
    spec = basic_spec(9, 9, 1, 2, 2, seed=9, dense_prob=0.4, actions_names=moves)
    ref = run_reference_basic(R, spec, [2], 30, moves)
    check_against_oracle(spec, [2], 30, ref)
    save("basic_1layer", spec, [2], ref)

    print("float_appearance_3layer: override_entity_map with arbitrary floats, 3 layers (f64 layer sum)")
    rng = np.random.default_rng(42)
    emap = {k: (rng.standard_normal(4) * 10.0 ** rng.integers(-3, 4, 4)) for k in ["EmptyEntity", "Wall", "Gem", "Walker"]}
    app = np.stack([emap["EmptyEntity"], emap["Wall"], emap["Gem"], emap["Gem"], emap["Walker"]])
    spec = basic_spec(10, 12, 3, 3, 2, seed=10, dense_prob=0.3, actions_names=moves, appearance=app)
    ref = run_reference_basic(R, spec, [4], 12, moves, entity_map_override=emap)
    check_against_oracle(spec, [4], 12, ref)
    save("float_appearance_3layer", spec, [4], ref)

    print("rgb_treasurehunt: RGBObservationSpec (uint8 colours summed over layers, clip(0,255)/255)")
    spec = O.treasurehunt_spec(12, 11, 3, 3, spawn_prob=0.05, seed=21, dense_prob=0.2)
    rgb_map = R["observation_spec"].RGBObservationSpec(
        ["EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"], full_view=False, vision_radius=3).entity_map
    kinds = ["EmptyEntity", "EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]
    spec.appearance = np.stack([rgb_map[k].astype(np.float64) for k in kinds])
    spec.num_channels, spec.obs_post = 3, 1
    ids = [0, 9]
    ref = run_reference_treasurehunt(R, spec, ids, 15, rgb=True)
    check_against_oracle(spec, ids, 15, ref)
    assert 0.0 < ref["obs"].max() <= 1.0
    save("rgb_treasurehunt", spec, ids, ref)

    print("tag_9x9: examples/tag TagAgent (move, tag an adjacent NotIt agent, reward for not being it)")
    spec = tag_spec(9, 9, 5, 2, seed=31)
    ids = [0, 4, 11]
    ref = run_reference_tag(R, spec, ids, 40)
    check_against_oracle(spec, ids, 40, ref)
    assert (ref["agent_state"][1:] != ref["agent_state"][:-1]).any(), "nobody was ever tagged"
    save("tag_9x9", spec, ids, ref)

    print("tag_crowded_6x7: 8 taggers on a 4x5 interior")
    spec = tag_spec(6, 7, 8, 2, seed=32, reward_per_turn=2.5)
    ids = [1, 2]
    ref = run_reference_tag(R, spec, ids, 30)
    check_against_oracle(spec, ids, 30, ref)
    save("tag_crowded_6x7", spec, ids, ref)

    print("cleanup_15x16: examples/cleanup (rivers/pollution/apples, conditional transitions, beam timers, beams)")
    spec = cleanup_spec(15, 16, 4, 3, seed=41, beam_radius=3, pollution_p=0.06, apple_p=0.03)
    ids = [0, 7]
    ref = run_reference_cleanup(R, spec, ids, 40, initial_apples=6)
    check_against_oracle(spec, ids, 40, ref, injected=True)
    assert (ref["grid"] == 7).any() and (ref["grid"] == 9).any() and (ref["grid"] == 4).any(), "beams / pollution never appeared"
    save("cleanup_15x16", spec, ids, ref)

    make_round2_fixtures(R)
    make_round3_fixtures(R)
    make_round5_fixtures(R)
    make_round6_fixtures(R)

    print("stock_np_random: unmodified Treasurehunt classes + RandomModel on np.random.seed(0)")
    ref = run_reference_stock(R, 10, 10, 2, 2, 0.05, 60, np_seed=0)
    spec = O.treasurehunt_spec(10, 10, 2, 2, spawn_prob=0.05, seed=0)
    from oracle import numpy_order
    mine = numpy_order.rollout_numpy_order(spec, 60, np_seed=0)
    for k in ("grid0", "pos0", "obs", "actions", "rewards", "total_reward", "grid", "pos"):
        assert np.array_equal(mine[k], ref[k]), f"numpy-order restatement differs in {k}"
    save("stock_np_random", spec, [0], ref, extra={"np_seed": np.array(0)})

    print("buffer_ring: reference replay Buffer semantics")
    make_buffer_fixture()
    print(f"done in {time.time() - t0:.1f}s")
    return 0



def make_round2_fixtures(R):
    """More reference-generated pins for the widened rule sets (round 2): Cleanup at the example's own shape and at a
    small odd one with another beam radius, Tag at the example's own shape."""
    print("cleanup_21x31_default: examples/cleanup at its configured shape (21x31x3, 10 agents, 11x11 window)")
    spec = cleanup_spec(21, 31, 10, 5, seed=43, beam_radius=3, pollution_p=0.05, apple_p=0.03)
    ids = [3]
    ref = run_reference_cleanup(R, spec, ids, 14, initial_apples=20)
    check_against_oracle(spec, ids, 14, ref, injected=True)
    save("cleanup_21x31_default", spec, ids, ref)

    print("cleanup_13x12_r2: small odd-sized Cleanup world, beam radius 2, 3 agents")
    spec = cleanup_spec(13, 12, 3, 2, seed=44, beam_radius=2, pollution_p=0.08, apple_p=0.05)
    ids = [1, 6]
    ref = run_reference_cleanup(R, spec, ids, 30, initial_apples=4)
    check_against_oracle(spec, ids, 30, ref, injected=True)
    assert (ref["grid"] == 7).any() or (ref["grid"] == 9).any(), "no beam was ever fired"
    save("cleanup_13x12_r2", spec, ids, ref)

    print("tag_11x11_default: examples/tag at its configured shape (11x11, 5 agents, 9x9 window)")
    spec = tag_spec(11, 11, 5, 4, seed=45)
    ids = [2, 5]
    ref = run_reference_tag(R, spec, ids, 25)
    check_against_oracle(spec, ids, 25, ref)
    save("tag_11x11_default", spec, ids, ref)


def make_round3_fixtures(R):
    """full_view observations (sorrel/observation/observation_spec.py:140-142, 197-203 -> visual_field.py:53-55: the whole
    map, appearance summed over layers, no shift / crop / fill): the reference's OWN OneHotObservationSpec and
    RGBObservationSpec with full_view=True observe the world of a running Treasurehunt harness after every turn."""
    print("full_view_treasurehunt: 9x11x2, 3 agents, 7 turns, whole-map one-hot and RGB observations")
    spec = O.treasurehunt_spec(9, 11, 3, 2, spawn_prob=0.08, seed=21, dense_prob=0.2)
    ids = [0, 6]
    turns = 7
    entity_list = ["EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]
    dims = (spec.height, spec.width)
    onehot = R["observation_spec"].OneHotObservationSpec(entity_list, full_view=True, env_dims=dims)
    rgb = R["observation_spec"].RGBObservationSpec(entity_list, full_view=True, env_dims=dims)
    assert tuple(onehot.input_size) == (6,) + dims and tuple(rgb.input_size) == (3,) + dims
    E = len(ids)
    full = dict(full_onehot0=np.zeros((E, 6) + dims), full_onehot=np.zeros((turns, E, 6) + dims),
                full_rgb=np.zeros((turns, E, 3) + dims))
    out = dict(grid0=np.zeros((E, spec.layers) + dims, np.uint8), pos0=np.zeros((E, spec.num_agents, 2), np.uint8),
               grid=np.zeros((turns, E, spec.layers) + dims, np.uint8), pos=np.zeros((turns, E, spec.num_agents, 2), np.uint8))
    for n, env_id in enumerate(ids):
        Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, Ctx.spec, Ctx.scripted = spec.seed, int(env_id), 0, 0, spec, None
        env, CounterEmpty = make_treasurehunt_env(R, spec, turns)
        out["grid0"][n] = type_ids_treasurehunt(R, env.world, CounterEmpty)
        out["pos0"][n] = [a.location[:2] for a in env.agents]
        v = onehot.observe(env.world)                              # <- the reference's own full-view observe
        assert v.dtype == np.float64 and v.shape == (6,) + dims
        full["full_onehot0"][n] = v
        for t in range(turns):
            Ctx.turn = env.turn + 1
            env.take_turn()
            out["grid"][t, n] = type_ids_treasurehunt(R, env.world, CounterEmpty)
            out["pos"][t, n] = [a.location[:2] for a in env.agents]
            full["full_onehot"][t, n] = onehot.observe(env.world, location=(1, 1, 1))    # (a location is ignored with full_view)
            full["full_rgb"][t, n] = rgb.observe(env.world)
    mine = O.rollout(spec, ids, turns)
    assert np.array_equal(mine["grid"], out["grid"]) and np.array_equal(mine["grid0"], out["grid0"]), "restatement differs from the reference"
    # the closed form the engine implements: appearance[type] summed over layers, channel-major
    want = spec.appearance[out["grid"]].sum(axis=2).transpose(0, 1, 4, 2, 3)
    assert np.array_equal(want, full["full_onehot"]), "closed form of the full view differs from the reference"
    rgb_map = {k: np.asarray(val, dtype=np.float64) for k, val in rgb.entity_map.items()}
    out.update(full)
    out["rgb_table"] = np.stack([rgb_map[k] for k in entity_list])
    save("full_view_treasurehunt", spec, ids, out)


def mixed_views(spec: O.Spec, agent_defs):
    """Per-agent views of a Treasurehunt-typed world for the restatement (`O.agent_view`): types are
    [Sand, EmptyEntity, Wall, Gem, Bone, Food, TreasurehuntAgent] with kinds ..."""
    kinds = ["EmptyEntity", "EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]
    moves = {"up": (-1, 0), "down": (1, 0), "left": (0, -1), "right": (0, 1)}
    views, full = [], []
    for d in agent_defs:
        emap = d["entity_map"]
        app = np.stack([np.asarray(emap[k], dtype=np.float64) for k in kinds])
        fill = kinds.index(d["fill"])
        views.append(O.agent_view(spec, d["radius"], app, fill, [moves.get(n, (0, 0))[0] for n in d["actions"]],
                                  [moves.get(n, (0, 0))[1] for n in d["actions"]]))
        full.append(bool(d["full_view"]))
    return views, full


def make_round5_fixtures(R):
    """Agents that DIFFER (sorrel/agents/agent.py:38-48: every agent holds its own observation_spec / action_spec; Agent.transition,
    agent.py:155-173, observes through self.observation_spec and acts through self.action_spec): one Treasurehunt world stepped by
    the reference's own take_turn with five TreasurehuntAgents -- radius 2, radius 4, full_view (the whole map, observation_spec.py:197-203),
    a reordered seven-channel entity list with another fill kind, and an overridden non-one-hot three-channel map -- and three different
    action lists.  Each agent's flattened float32 window as its replay Buffer stored it, per turn."""
    print("mixed_specs_treasurehunt: 12x13x2, five agents with five observation specs / three action lists, 12 turns")
    OneHot = R["observation_spec"].OneHotObservationSpec
    ActionSpec = R["action_spec"].ActionSpec
    Environment = R["environment"].Environment
    th, tha, thw = R["th_entities"], R["th_agents"], R["th_world"]
    CounterEmpty, CounterModel, _ = build_plugins(R)
    std = ["EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]
    H, W = 12, 13
    spec = O.treasurehunt_spec(H, W, 5, 2, spawn_prob=0.06, seed=77, dense_prob=0.3)
    odd_list = ["Wall", "EmptyEntity", "TreasurehuntAgent", "Gem", "Food", "Bone", "Lava"]
    float_map = {"EmptyEntity": np.array([0.0, 0.25, 0.0]), "Wall": np.array([1.0, 0.0, 0.5]), "Gem": np.array([0.0, 2.0, 0.0]),
                 "Bone": np.array([0.5, 0.5, 0.5]), "Food": np.array([0.0, 0.0, 3.0]), "TreasurehuntAgent": np.array([7.0, 0.0, 1.5])}
    defs = [
        dict(list=std, radius=2, full_view=False, fill="Wall", actions=["up", "down", "left", "right"], override=None),
        dict(list=std, radius=4, full_view=False, fill="Wall", actions=["up", "down", "left", "right"], override=None),
        dict(list=std, radius=0, full_view=True, fill="Wall", actions=["left", "right", "up", "down", "noop"], override=None),
        dict(list=odd_list, radius=3, full_view=False, fill="Gem", actions=["up", "down", "left", "right"], override=None),
        dict(list=std, radius=2, full_view=False, fill="EmptyEntity", actions=["down", "up", "right"], override=float_map),
    ]
    turns, ids = 12, [0, 3, 11]

    def ospec_of(d):
        if d["full_view"]:
            o = OneHot(d["list"], full_view=True, env_dims=(H, W), fill_entity_kind=d["fill"])
        else:
            o = OneHot(d["list"], full_view=False, vision_radius=d["radius"], fill_entity_kind=d["fill"])
        if d["override"] is not None:
            o.override_entity_map(d["override"])
        return o

    class MixedEnv(Environment):
        def setup_agents(self):
            self.agents = []
            for slot, d in enumerate(defs):
                ospec = ospec_of(d)
                n = len(next(iter(ospec.entity_map.values())))
                size = n * (H * W if d["full_view"] else (2 * d["radius"] + 1) ** 2)
                ospec.override_input_size((size,))
                aspec = ActionSpec(list(d["actions"]))
                model = CounterModel(ospec.input_size, aspec.n_actions, memory_size=turns + 1, slot=slot)
                self.agents.append(tha.TreasurehuntAgent(observation_spec=ospec, action_spec=aspec, model=model))

        def populate_environment(self):
            for index in np.ndindex(self.world.map.shape):
                y, x, z = index
                if (y in [0, H - 1] or x in [0, W - 1]) and z == 1:
                    self.world.add(index, th.Wall())
                elif z == 0:
                    self.world.add(index, th.Sand())
            for y in range(1, H - 1):
                for x in range(1, W - 1):
                    idx = int(O.cell_index(spec, y, x, 1))
                    if int(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, 0, O.STREAM_DENSE, idx)) < O.prob_threshold(spec.dense_prob):
                        k = int(O.categorical(O.rng_u32(Ctx.seed, Ctx.env, Ctx.epoch, 0, O.STREAM_DENSE_KIND, idx), 3))
                        v = self.world.values
                        self.world.add((y, x, 1), [th.Gem(v["gem"]), th.Food(v["food"]), th.Bone(v["bone"])][k])
            for (y, x), agent in zip(O.place_agents(spec, Ctx.env, Ctx.epoch), self.agents):
                self.world.add((int(y), int(x), 1), agent)

    cfg = {"world": {"height": H, "width": W, "gem_value": spec.type_value[3], "food_value": spec.type_value[5],
                     "bone_value": spec.type_value[4], "spawn_prob": spec.spawn_prob[1]},
           "experiment": {"epochs": 1, "max_turns": turns, "record_period": 1}}
    E, A = len(ids), len(defs)
    agent_defs = []
    probe = [ospec_of(d) for d in defs]
    for d, o in zip(defs, probe):
        agent_defs.append(dict(radius=d["radius"], full_view=d["full_view"], fill=d["fill"], actions=d["actions"],
                               entity_map={k: np.asarray(v, dtype=np.float64) for k, v in o.entity_map.items()}))
    shapes = [((len(next(iter(a["entity_map"].values()))), H, W) if a["full_view"]
               else (len(next(iter(a["entity_map"].values()))), 2 * a["radius"] + 1, 2 * a["radius"] + 1)) for a in agent_defs]
    out = dict(grid0=np.zeros((E, 2, H, W), np.uint8), pos0=np.zeros((E, A, 2), np.uint8),
               actions=np.zeros((turns, E, A), np.uint8), rewards=np.zeros((turns, E, A), np.float32),
               dones=np.zeros((turns, E, A), np.float32), total_reward=np.zeros((turns, E), np.float64),
               grid=np.zeros((turns, E, 2, H, W), np.uint8), pos=np.zeros((turns, E, A, 2), np.uint8))
    for a, s in enumerate(shapes):
        out[f"obs_a{a}"] = np.zeros((turns, E) + s, np.float32)
    for n, env_id in enumerate(ids):
        Ctx.seed, Ctx.env, Ctx.epoch, Ctx.turn, Ctx.spec, Ctx.scripted = spec.seed, int(env_id), 0, 0, spec, None
        world = thw.TreasurehuntWorld(config=cfg, default_entity=CounterEmpty())
        env = MixedEnv(world, cfg)
        out["grid0"][n] = type_ids_treasurehunt(R, env.world, CounterEmpty)
        out["pos0"][n] = [a.location[:2] for a in env.agents]
        for t in range(turns):
            Ctx.turn = env.turn + 1
            env.take_turn()                                  # <- the reference's own loop over agents that differ
            for a, agent in enumerate(env.agents):
                mem = agent.model.memory
                out[f"obs_a{a}"][t, n] = mem.states[t].reshape(shapes[a])
                out["actions"][t, n, a], out["rewards"][t, n, a], out["dones"][t, n, a] = mem.actions[t], mem.rewards[t], mem.dones[t]
                out["pos"][t, n, a] = agent.location[:2]
            out["total_reward"][t, n] = env.world.total_reward
            out["grid"][t, n] = type_ids_treasurehunt(R, env.world, CounterEmpty)
    views, full = mixed_views(spec, agent_defs)
    mine = O.rollout_mixed(views, full, ids, turns)
    for k in mine:
        if not np.array_equal(mine[k], out[k]):
            raise AssertionError(f"mixed restatement differs from the reference in {k} at {np.argwhere(mine[k] != out[k])[0]}")
    assert not out["dones"].any()
    assert out["rewards"].any() and len(np.unique(out["actions"][:, :, 2])) == 5, "the fixture should exercise pickups and all five actions"
    extra = {"agents_json": np.array(json.dumps([dict(radius=a["radius"], full_view=a["full_view"], fill=a["fill"], actions=a["actions"],
                                                       entity_list=list(a["entity_map"]),
                                                       entity_map={k: v.tolist() for k, v in a["entity_map"].items()}) for a in agent_defs]))}
    save("mixed_specs_treasurehunt", spec, ids, out, extra)


def make_round6_fixtures(R):
    """More than 64 agents (the reference steps any ``self.agents`` list, sorrel/environment.py:92-93): the reference's OWN take_turn over 80
    Treasurehunt agents on a 24x26 map (contention: a seventh of the interior is agents) and 70 Tag agents on 20x21."""
    print("many_agents_80_treasurehunt: 24x26, 80 agents, r=2, dense, 2 envs x 5 turns")
    spec = O.treasurehunt_spec(24, 26, 80, 2, spawn_prob=0.03, seed=17, dense_prob=0.2)
    ids = [0, 4099]
    ref = run_reference_treasurehunt(R, spec, ids, 5)
    check_against_oracle(spec, ids, 5, ref)
    save("many_agents_80_treasurehunt", spec, ids, ref)
    print("many_agents_70_tag: 20x21, 70 Tag agents, r=3, 1 env x 6 turns")
    spec = tag_spec(20, 21, 70, 3, seed=23)
    ids = [5]
    ref = run_reference_tag(R, spec, ids, 6)
    check_against_oracle(spec, ids, 6, ref)
    save("many_agents_70_tag", spec, ids, ref)


def make_buffer_fixture():
    """Replay ring semantics of the reference Buffer (sorrel/buffers.py:11-154): index arithmetic,
    n_frames stacking in current_state(), add_empty(), and sample() for given draws."""
    ref_loader.install()
    from sorrel.buffers import Buffer

    cap, nf, obs = 7, 3, 5
    rng = np.random.default_rng(3)
    buf = Buffer(capacity=cap, obs_shape=(obs,), n_frames=nf)
    T = 20
    states = rng.standard_normal((T, obs)).astype(np.float32)
    actions = rng.integers(0, 4, T)
    rewards = rng.standard_normal(T).astype(np.float32)
    dones = (rng.random(T) < 0.2).astype(np.float32)
    rec = dict(idx=[], size=[], cur=[])
    for t in range(T):
        buf.add(states[t], int(actions[t]), float(rewards[t]), bool(dones[t]))
        if t == 11:
            buf.add_empty()
        rec["idx"].append(buf.idx)
        rec["size"].append(buf.size)
        cur = buf.current_state()
        pad = np.full((nf - 1, obs), np.nan, np.float32)
        pad[: cur.shape[0]] = cur
        rec["cur"].append(pad)
    np.random.seed(5)
    s, a, r, ns, d, valid = buf.sample(3)
    np.random.seed(5)
    draws = np.random.choice(max(1, buf.size - nf - 1), 3, replace=False)
    out = dict(states=states, actions=actions, rewards=rewards, dones=dones, idx=np.array(rec["idx"]), size=np.array(rec["size"]),
               cur=np.stack(rec["cur"]), final_states=buf.states.copy(), final_actions=buf.actions.copy(),
               sample_draws=draws, s=s, a=a, r=r, ns=ns, d=d, valid=valid,
               params=np.array([cap, nf, obs, T]))
    path = os.path.join(GOLDEN_DIR, "buffer_ring.npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path)} bytes)")

    # Files written by the reference's OWN Buffer.save / SavedGames.save (sorrel/buffers.py:168-179, 361-379): what
    # the product's Buffer.load must read, and what its save must reproduce for the same history.
    from sorrel.buffers import SavedGames

    path = os.path.join(GOLDEN_DIR, "buffer_saved_by_reference.npz")
    buf.save(path)
    print(f"  wrote {path} ({os.path.getsize(path)} bytes)")
    # generate_memories' container (sorrel/environment.py:235-240, 297-300): two "games" of 4 turns appended from a
    # model memory that carries positions and is NOT cleared between games (the reference re-appends it whole)
    mem = Buffer(capacity=16, obs_shape=(obs,), n_frames=1, positions=(2,))
    sg = SavedGames(capacity=2 * 4, obs_shape=(obs,), n_frames=1, positions=(2,))
    k = 0
    for game in range(2):
        for _ in range(4):
            mem.add(states[k], int(actions[k]), float(rewards[k]), bool(dones[k]), positions=(k + 1, 2 * k))
            k += 1
        sg.add_from_buffer(mem)
    path = os.path.join(GOLDEN_DIR, "savedgames_by_reference.npz")
    sg.save(path)
    print(f"  wrote {path} ({os.path.getsize(path)} bytes)")

if __name__ == "__main__":
    if sys.argv[1:] == ["buffers"]:      # only the replay-buffer fixtures
        make_buffer_fixture()
        sys.exit(0)
    if sys.argv[1:] == ["round2"]:       # only the fixtures added in round 2
        make_round2_fixtures(_import_reference())
        sys.exit(0)
    if sys.argv[1:] == ["round5"]:       # only the fixture added in round 5
        make_round5_fixtures(_import_reference())
        sys.exit(0)
    if sys.argv[1:] == ["round6"]:       # only the fixtures added in round 6
        make_round6_fixtures(_import_reference())
        sys.exit(0)
    if sys.argv[1:] == ["round3"]:       # only the fixtures added in round 3
        make_round3_fixtures(_import_reference())
        sys.exit(0)
    sys.exit(main())

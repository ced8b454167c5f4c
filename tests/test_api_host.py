"""CPU tests of the host-side mirror of Sorrel's plugin API (no kernel launches)."""
import doctest

import numpy as np
import pytest
import torch

from tests import helpers as H
import sorrel_amd.location as location_mod
from sorrel_amd.action.action_spec import ActionSpec
from sorrel_amd.agents import Agent, MovingAgent
from sorrel_amd.entities import EmptyEntity, Entity, Gem, SpawnRule, Wall
from sorrel_amd.environment import Environment, _normalise_config
from sorrel_amd.examples.treasurehunt import entities as th
from sorrel_amd.examples.treasurehunt.env import ENTITY_LIST, TreasurehuntEnv
from sorrel_amd.examples.treasurehunt.main import make_config
from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld
from sorrel_amd.location import Location, Vector
from sorrel_amd.models import RandomModel
from sorrel_amd.observation.observation_spec import OneHotObservationSpec
from sorrel_amd.spec import RULE_SPAWN, treasurehunt_spec
from sorrel_amd.utils.helpers import nearest_2_power, one_hot_encode, shift
from sorrel_amd.worlds import Gridworld


class CpuTreasurehuntEnv(TreasurehuntEnv):
    """No GPU in the build container: skip the device reset, keep everything else."""

    def spawn_agents(self):
        self.world.agent_layer = 1


def make_env(h=16, w=16, a=4, r=2, E=8):
    cfg = make_config(h, w, a, r)
    world = TreasurehuntWorld(cfg, th.EmptyEntity(), num_envs=E, device="cpu", seed=3)
    return CpuTreasurehuntEnv(world, cfg)


def test_location_doctests():
    # the reference's only tests are these two doctests (sorrel/location.py:10-14)
    res = doctest.testmod(location_mod)
    assert res.attempted >= 2 and res.failed == 0
    assert Location(1, 2, 3) + Location(2, 4, 8) == Location(3, 6, 11)
    assert Location(2, 4) * 3 == Location(6, 12, 0)
    assert Location(3, 3, 1) + Vector(1, 0) == (2, 3, 1)           # forward, facing north = up = y-1
    assert Location(3, 3, 1) + Vector(1, 0, direction=2) == (4, 3, 1)
    assert [tuple(l) for l in Location(0, 0, 0).adjacent((5, 5, 1))] == [(0, 1, 0), (1, 0, 0)]
    with pytest.raises(TypeError):
        Location(1, 2) + 3


def test_action_spec():
    spec = ActionSpec(["up", "down", "left", "right"])
    assert spec.n_actions == 4 and spec.actions == {0: "up", 1: "down", 2: "left", 3: "right"}
    assert spec.get_readable_action(2) == "left" and spec.get_action_index("right") == 3
    assert spec.get_action_index("jump") is None


def test_entity_contract():
    e = Entity()
    assert (e.value, e.passable, e.has_transitions, e.kind) == (0, False, False, "Entity")
    with pytest.raises(AttributeError):
        e.location
    e.location = (1, 2, 0)
    assert e.location == (1, 2, 0)
    assert Wall().value == -1 and not Wall().passable and Wall().kind == "Wall"
    assert EmptyEntity().passable and Gem(7).value == 7 and Gem(7).passable
    assert th.Sand().kind == "EmptyEntity" and th.Food(5).kind == "Food"
    assert th.EmptyEntity().has_transitions and isinstance(th.EmptyEntity.transition_rule, SpawnRule)
    assert repr(Gem(3)) == "Gem(value=3)"


def test_helpers():
    assert np.array_equal(one_hot_encode(2, 4), [0, 0, 1, 0])
    with pytest.raises(AssertionError):
        one_hot_encode(4, 4)
    a = np.arange(12.0).reshape(3, 4)
    s = shift(a, [1, -1], cval=np.nan)
    assert np.isnan(s[0]).all() and np.isnan(s[:, -1]).all() and np.array_equal(s[1:, :-1], a[:-1, 1:])
    assert [nearest_2_power(n) for n in (0, 1, 3, 8, 9)] == [1, 1, 4, 8, 16]


def test_observation_spec_contract():
    with pytest.raises(TypeError):
        OneHotObservationSpec(ENTITY_LIST, full_view=False)                 # vision_radius missing
    with pytest.raises(TypeError):
        OneHotObservationSpec(ENTITY_LIST, full_view=True)                  # env_dims missing
    o = OneHotObservationSpec(ENTITY_LIST, full_view=False, vision_radius=3)
    assert o.input_size == (6, 7, 7) and o.fill_entity_kind == "Wall" and o.vision_radius == 3
    assert not o.entity_map["EmptyEntity"].any()                            # zero vector, but keeps channel 0
    assert np.array_equal(o.entity_map["Gem"], [0, 0, 1, 0, 0, 0])
    o.override_input_size((294,))
    assert o.input_size == (294,)
    with pytest.raises(TypeError):
        o.observe(object(), None)                                           # location required
    full = OneHotObservationSpec(ENTITY_LIST, full_view=True, env_dims=(5, 5))
    assert full.input_size == (6, 5, 5) and full.vision_radius == 0


def test_rgb_observation_spec_matches_reference_colours():
    """generate_map of the RGB spec == the appearance table stored in the reference-generated fixture."""
    from sorrel_amd.observation.observation_spec import RGBObservationSpec

    d, spec = H.load_golden("rgb_treasurehunt")
    o = RGBObservationSpec(ENTITY_LIST, full_view=False, vision_radius=3)
    assert o.input_size == (3, 7, 7) and o.obs_post == 1 and o.num_channels == 3
    kinds = ["EmptyEntity", "EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]
    assert np.array_equal(np.stack([o.entity_map[k].astype(np.float64) for k in kinds]), spec.appearance)
    assert o.entity_map["Wall"].dtype == np.uint8


def test_gridworld_host_api():
    w = Gridworld(6, 7, 2, EmptyEntity(), num_envs=3, device="cpu")
    assert w.grid.shape == (3, 2, 6, 7) and (w.height, w.width, w.layers) == (6, 7, 2)
    assert float(w.total_reward.sum()) == 0.0 and w.turn == 0 and w.max_turns == 0 and w.is_done is False
    w.add((2, 3, 1), Gem(5))
    assert w.observe((2, 3, 1)).value == 5 and w.observe((2, 3, 1), env=2).kind == "Gem"
    w.add((1, 1, 0), Wall(), env=1)
    assert w.observe((1, 1, 0), env=1).kind == "Wall" and w.observe((1, 1, 0), env=0).kind == "EmptyEntity"
    assert [e.kind for e in w.observe_all_layers((2, 3, 0))] == ["EmptyEntity", "Gem"]
    removed = w.remove((2, 3, 1))
    assert removed.kind == "Gem" and w.observe((2, 3, 1)).kind == "EmptyEntity"
    assert w.valid_location((5, 6, 1)) and not w.valid_location((6, 0, 0)) and not w.valid_location((-1, 0, 0))
    assert w.valid_location(Location(0, 0, 0))
    with pytest.raises(IndexError):
        w.valid_location((1, 1))
    with pytest.raises(IndexError):
        w.observe((6, 0, 0))
    g = Gem(2)
    w.add((4, 4, 1), g, env=0)
    assert w.move(g, (4, 5, 1)) and g.location == (4, 5, 1) and w.observe((4, 4, 1)).kind == "EmptyEntity"
    w.add((4, 6, 1), Wall(), env=0)
    assert not w.move(g, (4, 6, 1))                                         # impassable target
    kinds = w.get_entities_of_kind("Wall", env=0)
    assert [k.location for k in kinds] == [(4, 6, 1)]
    m = w.map
    assert m.shape == (6, 7, 2) and m[4, 5, 1].kind == "Gem" and m[4, 5, 1].location == (4, 5, 1)
    w.create_world()
    assert int(w.grid.max()) == w.default_type


def test_config_normalisation():
    c = _normalise_config({"experiment": {"epochs": 3}, "model": {"r": 2}})
    assert c.experiment.epochs == 3 and c.model.get("missing", 7) == 7
    d = _normalise_config(["experiment.epochs=4", "world.height=9"])
    assert str(d.experiment.epochs) == "4" and str(d.world.height) == "9"


def test_compile_spec_matches_canonical_tables():
    env = make_env(32, 32, 8, 3, E=4)
    s = env.compile_spec()
    ref = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=3)
    assert (s.height, s.width, s.layers, s.num_agents, s.vision_radius, s.num_channels, s.agent_layer) == \
           (ref.height, ref.width, ref.layers, ref.num_agents, ref.vision_radius, ref.num_channels, ref.agent_layer)
    names = s.type_names
    # same semantics per class, whatever ids the registry handed out
    for cls, rname in (("Sand", "Sand"), ("EmptyEntity", "EmptyEntity"), ("Wall", "Wall"), ("Gem", "Gem"),
                       ("Bone", "Bone"), ("Food", "Food"), ("TreasurehuntAgent", "TreasurehuntAgent")):
        t, rt = names.index(cls), ref.type_names.index(rname)
        assert s.type_value[t] == ref.type_value[rt] and s.type_passable[t] == ref.type_passable[rt]
        assert s.type_rule[t] == ref.type_rule[rt]
        assert np.array_equal(s.appearance[t], ref.appearance[rt])
    sp = names.index("EmptyEntity")
    assert s.type_rule[sp] == RULE_SPAWN and s.spawn_prob[sp] == 0.005
    assert [names[c] for c in s.spawn_choices[sp]] == ["Gem", "Food", "Bone"]            # entities.py:75-83 order
    assert names[s.default_type] == "EmptyEntity" and names[s.fill_type] == "Wall"
    assert [names[t] for t in s.layer_fill_type] == ["Sand", "EmptyEntity"]
    assert s.layer_border_type[0] == 255 and names[s.layer_border_type[1]] == "Wall"
    assert (s.action_dy, s.action_dx) == ([-1, 1, 0, 0], [0, 0, -1, 1])
    cfg = s.to_config(4, 0)
    assert cfg.num_types == 7 and cfg.num_envs == 4
    # the oracle accepts the compiled tables as they are
    H.oracle_spec(s).validate()


def test_unsupported_plugins_fail_loudly():
    class Teleporter(Entity):
        def __init__(self):
            super().__init__()
            self.has_transitions = True

        def transition(self, world):          # arbitrary Python: cannot run on the device
            pass

    env = make_env()
    env.world.add((3, 3, 1), Teleporter())
    with pytest.raises(ValueError, match="transition_rule"):
        env.compile_spec()

    class Ghost(Entity):
        pass

    env2 = make_env()
    env2.world.add((3, 3, 1), Ghost())        # kind "Ghost" is not in the entity_list
    with pytest.raises(KeyError):
        env2.compile_spec()


def test_agents_are_batched_slots():
    env = make_env(10, 10, 2, 2, E=5)
    assert [a.slot for a in env.agents] == [0, 1] and env.num_envs == 5
    a0 = env.agents[0]
    assert isinstance(a0, MovingAgent) and isinstance(a0, Agent) and a0.has_transitions and a0.kind == "TreasurehuntAgent"
    env.world.add((4, 5, 1), a0)
    assert a0.location == (4, 5, 1) and a0.locations.shape == (5, 3)
    assert a0.movement(0) == (3, 5, 1) and a0.movement(3) == (4, 6, 1)
    moved = a0.movement(torch.tensor([0, 1, 2, 3, 0]))
    assert moved.tolist() == [[3, 5, 1], [5, 5, 1], [4, 4, 1], [4, 6, 1], [3, 5, 1]]
    assert isinstance(a0.model, RandomModel) and a0.model.device_random


def test_border_precondition_is_checked():
    class OpenEnv(Environment):
        def setup_agents(self):
            o = OneHotObservationSpec(["EmptyEntity", "Wall", "Walker"], full_view=False, vision_radius=1)
            self.agents = [Walker(o, ActionSpec(["up", "down", "left", "right"]), RandomModel((27,), 4))]

        def populate_environment(self):
            self.world.add((2, 2, 0), self.agents[0])    # no walls at all

    class Walker(MovingAgent):
        def reset(self): ...
        def pov(self, world): ...
        def get_action(self, state): ...
        def is_done(self, world): return False

    env = OpenEnv(Gridworld(5, 5, 1, EmptyEntity(), num_envs=2, device="cpu"), {"experiment": {"epochs": 1}})
    with pytest.raises(ValueError, match="border"):
        env._validate_border()


def test_replay_buffer_matches_reference_ring_semantics():
    """The device replay ring (num_envs = 1) against a trace of the reference Buffer: index arithmetic,
    n_frames stacking of current_state(), add_empty(), and sample() for the same draws."""
    from sorrel_amd.buffers import Buffer

    d = np.load(H.GOLDEN_DIR + "/buffer_ring.npz")
    cap, nf, obs, T = (int(v) for v in d["params"])
    buf = Buffer(capacity=cap, obs_shape=(obs,), n_frames=nf, num_envs=1, device="cpu")
    for t in range(T):
        buf.add(torch.from_numpy(d["states"][t][None]), torch.tensor([int(d["actions"][t])]),
                torch.tensor([float(d["rewards"][t])]), float(d["dones"][t]))
        if t == 11:
            buf.add_empty()
        assert (buf.idx, buf.size) == (int(d["idx"][t]), int(d["size"][t]))
        cur = buf.current_state()[:, 0].numpy()
        want = d["cur"][t]
        want = want[~np.isnan(want).any(axis=1)]
        assert cur.shape == want.shape and np.array_equal(cur, want), t
    assert np.array_equal(buf.states[:, 0].numpy(), d["final_states"])
    assert np.array_equal(buf.actions[:, 0].numpy(), d["final_actions"])
    s, a, r, ns, dn, valid = buf.sample(3, starts=d["sample_draws"], envs=[0, 0, 0])
    for mine, ref in ((s, d["s"]), (a, d["a"]), (r, d["r"]), (ns, d["ns"]), (dn, d["d"]), (valid, d["valid"])):
        assert np.array_equal(mine.numpy(), ref)


# ------------------------------------------------------------------ Cleanup: classes -> tables, pinned by the reference fixture
CLEANUP_CFG = {
    "experiment": {"epochs": 1, "max_turns": 40},
    "env": {"height": 15, "width": 16, "layers": 3, "pollution_threshold": 0.5, "initial_apples": 6,
            "apple_spawn_chance": 0.03, "pollution_spawn_chance": 0.06, "mode": "DEFAULT"},
    "agent": {"agent": {"num": 4, "beam_radius": 3, "obs": {"vision": 3, "embeddings": 3}}},
}


def make_cleanup_env(E=3, seed=41, device="cpu", model_factory=None):
    from sorrel_amd.examples.cleanup.entities import EmptyEntity as CEmpty
    from sorrel_amd.examples.cleanup.env import CleanupEnv
    from sorrel_amd.examples.cleanup.world import CleanupWorld

    return CleanupEnv(CleanupWorld(CLEANUP_CFG, CEmpty(), num_envs=E, device=device, seed=seed), CLEANUP_CFG,
                      model_factory=model_factory)


def cleanup_type_map(ws):
    """fixture type id (oracle/make_golden.py cleanup_spec) -> id in the class-compiled spec."""
    first, second = {}, {}
    for t, n in enumerate(ws.type_names):
        (second if n in first else first)[n] = t
    order = ["EmptyEntity", "Sand", "Wall", "River", "Pollution", "AppleTree", "Apple"]
    return np.array([first[n] for n in order] + [first["CleanBeam"], second["CleanBeam"], first["ZapBeam"], second["ZapBeam"],
                                                  first["CleanupAgent"]], dtype=np.uint8)


def test_cleanup_classes_compile_to_the_reference_behaviour():
    """The Cleanup example's entity / agent classes, compiled to tables and run by the C oracle from
    the fixture's start state, reproduce what the REFERENCE produced (tests/golden/cleanup_15x16.npz):
    pins BecomeIfRule / AgeRule / CleanupRule compilation without a GPU."""
    d, _ = H.load_golden("cleanup_15x16")
    env = make_cleanup_env(E=1)
    ws = env.compile_spec()
    assert ws.agent_rule == 2 and ws.reward_total_factor == 2 and ws.action_kind == [0, 0, 0, 0, 1, 2]
    assert ws.type_rule.count(RULE_SPAWN) == 2 and ws.type_rule.count(2) == 6
    tmap = cleanup_type_map(ws)
    assert sorted(tmap.tolist()) == list(range(12))
    # the host-built template equals the reference's populate_environment map (apples / agents aside)
    tmpl = env.world.grid[0].numpy().copy()
    g0 = tmap[d["grid0"][0]]
    static = (g0 != tmap[6]) & (g0 != tmap[11]) & (tmpl != tmap[6]) & (tmpl != tmap[11])
    assert np.array_equal(tmpl[static], g0[static])
    assert int((tmpl == tmap[6]).sum()) == 6 and int((tmpl[1] == tmap[11]).sum()) == 4
    for n, env_id in enumerate(int(e) for e in d["env_ids"]):
        co = H.COracle(ws, 1, first_env_id=env_id)
        co.grid[0], co.pos[0], co.total[0] = tmap[d["grid0"][n]], d["pos0"][n], 0.0
        for t in range(d["obs"].shape[0]):
            assert co.step(0, t + 1, random_actions=True) == 0
            assert np.array_equal(co.obs[0], d["obs"][t, n]), f"obs turn {t}"
            assert np.array_equal(co.actions[0], d["actions"][t, n])
            assert np.array_equal(co.rewards[0], d["rewards"][t, n])
            assert co.total[0] == d["total_reward"][t, n]
            assert np.array_equal(co.grid[0], tmap[d["grid"][t, n]]), f"grid turn {t}"
            assert np.array_equal(co.pos[0], d["pos"][t, n]) and np.array_equal(co.agent_dir[0], d["agent_dir"][t, n])


def test_cleanup_observation_positional_code_matches_reference_fixture():
    """CleanupObservation = flattened visual field ++ positional code; the code rows the reference
    stored (float32) equal the host-built table gathered at the agents' cells."""
    from sorrel_amd.observation import embedding

    d, _ = H.load_golden("cleanup_15x16")
    env = make_cleanup_env(E=2)
    tab = embedding.positional_embedding_table(env.world, (3, 3)).numpy()
    assert tab.shape == (15, 16, 12)
    pos_at_pov = np.concatenate([d["pos0"][None], d["pos"][:-1]], axis=0)      # agent a observes before it moves
    for t in range(d["obs"].shape[0]):
        for n in range(d["obs"].shape[1]):
            for a in range(4):
                y, x = pos_at_pov[t, n, a]    # its own position only changes in its own act, after its pov
                assert np.array_equal(tab[y, x], d["pos_code"][t, n, a]), (t, n, a)
    spec = env.agents[0].observation_spec
    assert spec.input_size == (1, 9 * 49 + 12)
    with pytest.raises(ValueError):
        spec.observe(env.world, None)


def test_cleanup_populate_draws_per_env_without_replacement():
    env = make_cleanup_env(E=64)
    ws = env.compile_spec()
    tmap = cleanup_type_map(ws)
    g = env.world.grid.numpy()
    assert ((g[:, 0] == tmap[6]).reshape(64, -1).sum(1) == 6).all()            # 6 apples in every env
    assert ((g[:, 1] == tmap[11]).reshape(64, -1).sum(1) == 4).all()           # 4 agents on distinct cells
    pos = env.world.agent_pos.numpy()
    for e in range(64):
        for a in range(4):
            assert g[e, 1, pos[e, a, 0], pos[e, a, 1]] == tmap[11]
            assert g[e, 0, pos[e, a, 0], pos[e, a, 1]] == tmap[1]              # agents start on sand
    assert len({pos[e].tobytes() for e in range(64)}) > 32                      # placements differ across envs
    before = g.copy()
    env.reset()                                                                 # new epoch -> new draw, same template
    assert not np.array_equal(env.world.grid.numpy(), before)


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under sorrel_amd/ (Python or HIP) may import, load or name it,
    nor the reference tree."""
    import os
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sorrel_amd")
    bad = []
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) or "libgridstep_oracle" in src or "/root/reference" in src:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_engine_rejects_observation_destinations_the_kernel_would_overrun():
    """GridEngine.step(obs_out=...) / observe(out=..., pos=...) hand raw pointers to the kernels: anything but a
    contiguous tensor of exactly the engine's observation dtype / shape / device must be refused on the host
    (a uint8 ring slot would receive 4x its size in float32).  Pure host logic, no GPU."""
    import torch

    from sorrel_amd.engine import GridEngine
    from sorrel_amd.spec import treasurehunt_spec

    eng = GridEngine.__new__(GridEngine)            # the checks need no device handle
    eng.spec, eng.num_envs, eng.obs_dtype, eng.device, eng._h = treasurehunt_spec(16, 16, 4, 2), 6, torch.float32, torch.device("cpu"), None
    good = torch.zeros((6, 4, 6, 5, 5), dtype=torch.float32)
    assert eng._check_obs(good, "obs_out") is good
    for bad in (good.to(torch.uint8), good[:5], good[:, :2], good.double(), good.transpose(3, 4), torch.zeros((6, 600)), "nope"):
        with pytest.raises(ValueError):
            eng._check_obs(bad, "obs_out")
    pos = torch.zeros((6, 4, 2), dtype=torch.uint8)
    assert eng._check_pos(pos) is pos
    for bad in (pos.long(), pos[:, :3], pos.transpose(0, 1)):
        with pytest.raises(ValueError):
            eng._check_pos(bad)


REFERENCE_STYLE_SCRIPT = '''
# Written against the REFERENCE's import paths (the shape of sorrel/examples/treasurehunt/main.py + env.py), nothing from sorrel_amd:
import numpy as np

from sorrel.action.action_spec import ActionSpec
from sorrel.agents import Agent, MovingAgent
from sorrel.buffers import Buffer
from sorrel.entities import EmptyEntity, Entity, Gem, Wall
from sorrel.environment import Environment
from sorrel.location import Location
from sorrel.models.base_model import BaseModel, RandomModel
from sorrel.observation.observation_spec import OneHotObservationSpec
from sorrel.observation.visual_field import visual_field
from sorrel.utils.helpers import one_hot_encode, shift
from sorrel.worlds import Gridworld
from sorrel.worlds.gridworld import Gridworld as G2


class Walker(MovingAgent):
    def reset(self):
        pass

    def pov(self, world):
        return self.observation_spec.observe(world, self.location)

    def get_action(self, state):
        return self.model.take_action(state)

    def is_done(self, world):
        return world.is_done


class Env(Environment):
    def setup_agents(self):
        spec = OneHotObservationSpec(["EmptyEntity", "Wall", "Gem", "Walker"], full_view=False, vision_radius=2)
        actions = ActionSpec(["up", "down", "left", "right"])
        self.agents = [Walker(spec, actions, RandomModel(spec.input_size, actions.n_actions)) for _ in range(2)]

    def populate_environment(self):
        w = self.world
        for index in np.ndindex(w.height, w.width, w.layers):
            y, x, z = index
            if y in (0, w.height - 1) or x in (0, w.width - 1):
                w.add(index, Wall())
        w.add((3, 3, 0), Gem(5))
        for agent, loc in zip(self.agents, [(1, 1, 0), (5, 5, 0)]):
            w.add(loc, agent)


world = Gridworld(8, 9, 1, EmptyEntity(), device="cpu")
env = Env(world, {"experiment": {"epochs": 1, "max_turns": 5, "record_period": 1}})
spec = env.compile_spec()
assert G2 is Gridworld and issubclass(Walker, Agent) and issubclass(Gem, Entity)
assert Location(1, 2, 3) + Location(2, 4, 8) == Location(3, 6, 11)
print("OK", spec.height, spec.width, spec.layers, spec.num_agents, spec.num_channels, spec.vision_radius,
      sorted(set(spec.type_names)), [float(v) for v in spec.type_value], world.observe((3, 3, 0)).kind, env.agents[1].location)
import sorrel.environment, sorrel_amd.environment
assert sorrel.environment.Environment is sorrel_amd.environment.Environment
try:
    import sorrel.utils.logging
except ModuleNotFoundError as exc:
    print("out of scope:", "outside the hot path" in str(exc))
'''


def test_reference_import_paths_resolve_to_the_mirror(tmp_path):
    """SURVEY 8(b) "Import paths to mirror": a script written with the reference's own import lines builds a world and
    compiles its spec -- through ``python -m sorrel_amd.compat script.py`` and through ``compat.install()`` -- and parts
    of the reference outside the hot path fail with a message that says so.  (Subprocesses: the alias must not leak into
    this test process, where other tests import the real reference under the same name.)"""
    import subprocess
    import sys

    script = tmp_path / "experiment.py"
    script.write_text(REFERENCE_STYLE_SCRIPT)
    env = dict(__import__("os").environ, PYTHONPATH=H.ROOT, PYTHONDONTWRITEBYTECODE="1")
    for cmd in ([sys.executable, "-m", "sorrel_amd.compat", str(script)],
                [sys.executable, "-c", f"import sorrel_amd.compat as c, runpy; c.install(); c.install(); runpy.run_path({str(script)!r}, run_name='__main__')"]):
        out = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-3000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("OK")][0]
        assert line.startswith("OK 8 9 1 2 4 2 ['EmptyEntity', 'Gem', 'Walker', 'Wall']"), line
        assert "Gem (5, 5, 0)" in line and "out of scope: True" in out.stdout
    # without the alias the reference's names do not exist here (nothing is shadowed silently)
    out = subprocess.run([sys.executable, "-c", "import sorrel.environment"], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert out.returncode != 0 and "No module named 'sorrel'" in out.stderr
    # install() refuses to shadow a real `sorrel` distribution unless forced
    (tmp_path / "sorrel").mkdir()
    (tmp_path / "sorrel" / "__init__.py").write_text("REAL = True\n")
    code = ("import sorrel_amd.compat as c\n"
            "try:\n    c.install()\n    print('shadowed')\nexcept ImportError as e:\n    print('refused')\n"
            "c.install(force=True)\nimport sorrel.worlds\nprint(sorrel.worlds.Gridworld.__module__)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path),
                         env=dict(env, PYTHONPATH=H.ROOT + ":" + str(tmp_path)))
    assert out.returncode == 0 and out.stdout.split() == ["refused", "sorrel_amd.worlds.gridworld"], out.stdout + out.stderr


def test_buffer_add_copies_only_what_the_kernels_have_not_written():
    """``Buffer.add`` skips the state / reward / action copies when the kernels already wrote them into the row
    (``sgw_act``'s ``reward_row`` / ``action_row``, a window rendered into the state row) -- and ONLY then; the all-zero
    ``dones`` rows are rewritten only once a non-zero ``done`` has ever been stored."""
    from sorrel_amd.buffers import Buffer

    E = 4
    buf = Buffer(capacity=3, obs_shape=(5,), num_envs=E, device="cpu")
    state = torch.arange(E * 5, dtype=torch.float32).reshape(E, 5)
    action = torch.tensor([3, 1, 0, 2])
    reward = torch.tensor([1.0, -1.0, 0.0, 10.0])
    buf.add(state, action, reward, False)                            # nothing prefilled: everything is copied
    assert torch.equal(buf.states[0], state) and torch.equal(buf.actions[0], action) and torch.equal(buf.rewards[0], reward)
    assert buf.idx == 1 and buf.size == 1 and not buf._dones_dirty
    # the kernels wrote row 1 in place: the state row, the reward row and (announced through _prefilled) the action row
    buf.states[1].copy_(state * 2)
    buf.rewards[1].copy_(reward * 3)
    buf.actions[1].copy_(action + 1)
    other_action = action + 1                                        # the policy's tensor (another storage than the row)
    buf._prefilled = (1, other_action.data_ptr())
    buf.add(buf.states[1], other_action, buf.rewards[1], False)
    assert torch.equal(buf.states[1], state * 2) and torch.equal(buf.rewards[1], reward * 3) and torch.equal(buf.actions[1], action + 1)
    assert buf._prefilled is None
    # a _prefilled mark for another row, or another action tensor, is not trusted
    buf._prefilled = (0, action.data_ptr())
    buf.add(state, action, reward * 5, True)
    assert torch.equal(buf.actions[2], action) and torch.equal(buf.rewards[2], reward * 5) and bool((buf.dones[2] == 1).all())
    assert buf._dones_dirty and buf.idx == 0 and buf.size == 3
    buf.add(state, action, reward, False)                            # row 0 again: the ring is dirty now, zeros are written
    assert bool((buf.dones[0] == 0).all())
    # a reward tensor of another dtype sharing the row's address is still copied (never the case in practice; cheap to keep right)
    buf.add(state, action, reward.double(), 0)
    assert torch.equal(buf.rewards[1], reward)


def test_mark_dirty_and_invalidate_hooks():
    env = make_env()
    w = env.world
    m = w.mutations
    w.mark_dirty()
    assert w.mutations == m + 1
    spec = env.agents[0].observation_spec
    key = Environment._ospec_key(spec)
    assert Environment._ospec_key(spec) is key                       # cached on the spec
    spec.entity_map["Gem"][0] = 0.5                                   # in-place edit: not seen ...
    assert Environment._ospec_key(spec) is key
    spec.invalidate()                                                 # ... until the spec is told
    assert Environment._ospec_key(spec) != key


def test_buffer_add_clears_terminal_flags_of_rows_that_came_in_by_copy_or_load(tmp_path):
    """``Buffer.add`` always stores ``done`` in the reference (``sorrel/buffers.py:60``).  The mirror skips the store while every
    dones row is known to be zero; rows that arrive through ``load`` / ``add_from_buffer`` / ``add_turns`` may hold terminal flags,
    so a later ``add(done=False)`` onto such a row must clear it (round-3 advisor finding)."""
    import torch
    from sorrel_amd.buffers import Buffer, SavedGames

    def filled():
        b = Buffer(2, (3,), num_envs=2, device="cpu")
        b.add(torch.ones(2, 3), torch.zeros(2, dtype=torch.int64), torch.zeros(2), False)
        b.add(torch.ones(2, 3), torch.zeros(2, dtype=torch.int64), torch.zeros(2), True)        # the last row is terminal
        return b

    src = filled()
    src.save(tmp_path / "b.npz")
    loaded = Buffer.load(tmp_path / "b.npz")
    assert loaded.dones[1].tolist() == [1.0, 1.0] and loaded.idx == 0
    loaded.add(torch.zeros(2, 3), torch.zeros(2, dtype=torch.int64), torch.zeros(2), False)
    loaded.add(torch.zeros(2, 3), torch.zeros(2, dtype=torch.int64), torch.zeros(2), False)     # lands on the row that was terminal
    assert not loaded.dones.any()
    dst = Buffer(4, (3,), num_envs=2, device="cpu")
    dst.add_from_buffer(src)
    dst.idx = 1
    dst.add(torch.zeros(2, 3), torch.zeros(2, dtype=torch.int64), torch.zeros(2), False)
    assert not dst.dones[1].any()
    sg = SavedGames(3, (3,), num_envs=2, device="cpu")
    sg.add_turns(src.states, src.actions, src.rewards, src.dones)
    sg.idx = 1
    sg.add(torch.zeros(2, 3), torch.zeros(2, dtype=torch.int64), torch.zeros(2), False)
    assert not sg.dones[1].any()
    fresh = Buffer(2, (3,), num_envs=2, device="cpu")
    fresh.add(torch.ones(2, 3), torch.zeros(2, dtype=torch.int64), torch.zeros(2), True)
    fresh.clear()
    assert not fresh._dones_dirty and not fresh.dones.any()


def test_buffer_add_batch_wraps_with_a_flag_per_row():
    """``add_batch`` = k consecutive ``add`` calls (``sorrel/buffers.py:46-63``), also when the ring wraps in the middle of the batch and
    ``done`` carries a flag per row (round-5 advisor finding: capacity 5, idx 3, k 3 raised after the first segment was written)."""
    import torch
    from sorrel_amd.buffers import Buffer

    def rows(k, E=2):
        obs = torch.arange(k * E * 3, dtype=torch.float32).reshape(k, E, 3)
        act = torch.arange(k * E, dtype=torch.int64).reshape(k, E)
        rew = torch.arange(k * E, dtype=torch.float32).reshape(k, E) * 0.5
        done = (torch.arange(k * E).reshape(k, E) % 3 == 0).float()
        return obs, act, rew, done

    for done_form in ("per_row", "per_env", "scalar", "false"):
        a, b = Buffer(5, (3,), num_envs=2, device="cpu"), Buffer(5, (3,), num_envs=2, device="cpu")
        for buf in (a, b):
            for _ in range(3):                                    # idx 3: a batch of three rows wraps after two
                buf.add(torch.zeros(2, 3), torch.zeros(2, dtype=torch.int64), torch.zeros(2), True)
        obs, act, rew, done = rows(3)
        d = {"per_row": done, "per_env": done[0], "scalar": True, "false": False}[done_form]
        a.add_batch(obs, act, rew, d)
        for j in range(3):
            b.add(obs[j], act[j], rew[j], d[j] if done_form == "per_row" else d)
        assert (a.idx, a.size) == (b.idx, b.size) == (1, 5), done_form
        for name in ("states", "actions", "rewards", "dones"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (done_form, name)
    # a mis-shaped flag tensor is refused BEFORE anything is written
    c = Buffer(5, (3,), num_envs=2, device="cpu")
    c.idx = 3
    obs, act, rew, done = rows(3)
    import pytest
    with pytest.raises(ValueError):
        c.add_batch(obs, act, rew, done[:2])
    assert c.idx == 3 and c.size == 0 and not c.states.any() and not c.actions.any()


def test_alias_imports_leave_the_mirrors_module_specs_alone():
    """``import sorrel.buffers`` hands out the mirror's module object; its ``__spec__`` must stay ``sorrel_amd.buffers``'s (reload and
    relative imports go by it) -- round-3 advisor finding."""
    import importlib

    from sorrel_amd import compat

    compat.install()
    try:
        import sorrel.buffers as aliased
        import sorrel_amd.buffers as own

        assert aliased is own and own.__spec__.name == "sorrel_amd.buffers" and own.__name__ == "sorrel_amd.buffers"
        assert importlib.reload(own) is own
    finally:
        compat.uninstall() if hasattr(compat, "uninstall") else None


def test_compile_spec_accepts_agents_that_differ():
    """``sorrel/agents/agent.py:38-48``: every agent holds its own observation and action spec.  Each distinct pair compiles to its own
    table set (radius, channels, fill type, action deltas) over the same type registry; nothing is rejected."""
    from tests.mixed_env import make_mixed_env

    env, (d, base, views, full, defs) = make_mixed_env(2, "cpu", on_device=False)
    keys = [env._agent_key(a) for a in env.agents]
    assert len(set(keys)) == 5
    for agent, view, fv in zip(env.agents, views, full):
        ws = env.compile_spec(agent.observation_spec, agent.action_spec)
        assert ws.vision_radius == (0 if fv else view.vision_radius) and ws.num_channels == view.num_channels
        assert list(ws.action_dy) == list(view.action_dy) and list(ws.action_dx) == list(view.action_dx)
        kind_of = {"Sand": "EmptyEntity", "_FillEntity": agent.observation_spec.fill_entity_kind}
        for t, name in enumerate(ws.type_names):              # every registered type shows the appearance of its kind in THIS agent's map
            want = np.asarray(agent.observation_spec.entity_map[kind_of.get(name, name)], dtype=np.float64)
            assert np.array_equal(np.asarray(ws.appearance)[t], want), (name, t)
        assert np.array_equal(np.asarray(ws.appearance)[ws.fill_type], view.appearance[view.fill_type])
    assert env.compile_spec().vision_radius == 2          # default: agent 0's specs


def test_standard_hooks_rule_of_the_fast_and_the_speculative_loops():
    """Which agents the Environment may step without calling their hooks one by one (``Environment._standard_hooks``): the class that declares
    ``speculative_ok`` and subclasses that override no hook of the turn -- not a subclass with its own pov / get_action / act / transition /
    add_memory, not an instance with a hook patched on, not a class that never said so."""
    import types

    from sorrel_amd.agents import MovingAgent
    from sorrel_amd.environment import Environment
    from sorrel_amd.examples.treasurehunt.agents import TreasurehuntAgent

    def make(cls):
        return cls.__new__(cls)              # (the rule looks at classes and the instance dict only)

    class Renamed(TreasurehuntAgent):
        def is_done(self, world):
            return False

    class OwnPov(TreasurehuntAgent):
        def pov(self, world):
            return super().pov(world)

    class OwnMemory(Renamed):
        def add_memory(self, state, action, reward, done):
            pass

    class SaysNo(TreasurehuntAgent):
        speculative_ok = False

    class Plain(MovingAgent):
        def reset(self): ...
        def pov(self, world): ...
        def get_action(self, state): ...
        def is_done(self, world): ...

    ok = Environment._standard_hooks
    assert ok(make(TreasurehuntAgent)) and ok(make(Renamed))
    assert not ok(make(OwnPov)) and not ok(make(OwnMemory)) and not ok(make(SaysNo)) and not ok(make(Plain))
    patched = make(TreasurehuntAgent)
    patched.get_action = types.MethodType(lambda self, state: 0, patched)
    assert not ok(patched)

"""Test plumbing: the Treasurehunt world of ``tests/golden/mixed_specs_treasurehunt.npz`` through the product's mirror classes --
five agents that each hold their own observation spec and action list (``sorrel/agents/agent.py:38-48``)."""
import numpy as np

from tests import helpers as H


def make_mixed_env(E, device, model_factory=None, defs=None, on_device=True):
    from sorrel_amd.action.action_spec import ActionSpec
    from sorrel_amd.examples.treasurehunt.agents import TreasurehuntAgent
    from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
    from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
    from sorrel_amd.examples.treasurehunt.main import make_config
    from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld
    from sorrel_amd.models import RandomModel
    from sorrel_amd.observation.observation_spec import OneHotObservationSpec

    d, base, views, full, fixture_defs = H.load_mixed()
    defs = fixture_defs if defs is None else defs
    h, w = base.height, base.width

    class MixedEnv(TreasurehuntEnv):
        def setup_agents(self):
            self.agents = []
            for slot, a in enumerate(defs):
                if a["full_view"]:
                    ospec = OneHotObservationSpec(a["entity_list"], full_view=True, env_dims=(h, w), fill_entity_kind=a["fill"])
                else:
                    ospec = OneHotObservationSpec(a["entity_list"], full_view=False, vision_radius=a["radius"], fill_entity_kind=a["fill"])
                want = {k: np.asarray(v, dtype=np.float64) for k, v in a["entity_map"].items()}
                if any(not np.array_equal(ospec.entity_map[k], want[k]) for k in want):
                    ospec.override_entity_map(want)          # (the float map of the fixture's last agent)
                n_ch = len(want[next(iter(want))])
                ospec.override_input_size((n_ch * (h * w if a["full_view"] else (2 * a["radius"] + 1) ** 2),))
                aspec = ActionSpec(list(a["actions"]))
                model = (model_factory or RandomModel)(ospec.input_size, aspec.n_actions) if model_factory is None else \
                    model_factory(ospec.input_size, aspec.n_actions, slot)
                self.agents.append(TreasurehuntAgent(ospec, aspec, model))

        def spawn_agents(self):
            if on_device:
                return super().spawn_agents()
            self.world.agent_layer = 1          # (no GPU in the build container: skip the device reset, keep everything else)

    cfg = make_config(h, w, len(defs), 2, spawn_prob=base.spawn_prob[1])
    cfg["world"]["dense_prob"] = base.dense_prob
    world = TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device=device, seed=base.seed)
    return MixedEnv(world, cfg), (d, base, views, full, defs)


FIXTURE_TYPE_NAMES = ["Sand", "EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]   # type ids of the fixture / the oracle's treasurehunt_spec


def to_fixture_ids(env, grid: np.ndarray) -> np.ndarray:
    """The product's grid (type ids in registration order) in the fixture's numbering, by class name."""
    names = env.compile_spec().type_names
    lut = np.full(256, 255, np.uint8)
    for t, n in enumerate(names):
        if n in FIXTURE_TYPE_NAMES:
            lut[t] = FIXTURE_TYPE_NAMES.index(n)
    return lut[grid]

"""``CleanupEnv`` (``sorrel/examples/cleanup/env.py:41-140``) on the batched engine."""
import numpy as np
import torch

from sorrel_amd.action.action_spec import ActionSpec
from sorrel_amd.environment import Environment
from sorrel_amd.examples.cleanup.agents import CleanupAgent, CleanupObservation
from sorrel_amd.examples.cleanup.entities import Apple, AppleTree, River, Sand, Wall
from sorrel_amd.models import RandomModel

ENTITY_LIST = ["EmptyEntity", "Wall", "River", "Pollution", "AppleTree", "Apple", "CleanBeam", "ZapBeam", "CleanupAgent"]


class CleanupEnv(Environment):
    """config keys as in ``sorrel/examples/cleanup/configs``: ``env.*``, ``agent.agent.num``,
    ``agent.agent.beam_radius``, ``agent.agent.obs.vision`` / ``.embeddings``."""

    def __init__(self, world, config, model_factory=None):
        self._model_factory = model_factory
        super().__init__(world, config)

    def setup_agents(self):
        ac = self.config.agent.agent
        agents = []
        for _ in range(int(ac.num)):
            ospec = CleanupObservation(ENTITY_LIST, vision_radius=int(ac.obs.vision), embedding_size=int(ac.obs.get("embeddings", 3)))
            aspec = ActionSpec(["up", "down", "left", "right", "clean", "zap"])
            model = self._model_factory(ospec.input_size, aspec.n_actions) if self._model_factory else \
                RandomModel(ospec.input_size, aspec.n_actions)
            agents.append(CleanupAgent(ospec, aspec, model, beam_radius=int(ac.beam_radius)))
        self.agents = agents

    def override_agents(self, agents) -> None:
        self.agents = agents
        self._attach_agents()

    def populate_environment(self):
        """Walls around every layer; river in the top third (plus a two-column tongue), orchard in
        the bottom third, sand between (env.py:85-124).  The map is identical in every env; apples
        and agents are then drawn per env without replacement from the orchard / sand cells
        (env.py:126-140) -- from a seeded torch generator where the reference uses ``np.random``."""
        w = self.world
        H, W = w.height, w.width
        tmpl = np.empty((H, W, w.layers), dtype=object)
        spawn_points, apple_points = [], []
        for (y, x, z), _ in np.ndenumerate(tmpl):
            if y in (0, H - 1) or x in (0, W - 1):
                tmpl[y, x, z] = Wall()
            elif z == 0:
                if w.mode != "APPLE":
                    if (0 < y < H // 3) or (y < (H // 3) * 2 - 1 and x in (W // 3, 1 + W // 3)):
                        tmpl[y, x, z] = River()
                    elif H - 1 - H // 3 < y < H - 1:
                        tmpl[y, x, z] = AppleTree()
                        apple_points.append((y, x, z))
                    else:
                        tmpl[y, x, z] = Sand()
                        spawn_points.append((y, x, w.agent_layer))
                else:
                    tmpl[y, x, z] = AppleTree()
                    if y % 2 == 0 and x % 2 == 0:
                        spawn_points.append((y, x, w.agent_layer))
                    else:
                        apple_points.append((y, x, z))
        w.set_template(tmpl)
        gen = torch.Generator().manual_seed((w.seed * 1000003 + self.epoch) * 65537 + getattr(w, "first_env_id", 0))
        w.scatter_random(apple_points, int(w.initial_apples), Apple(), gen)
        pick = w.scatter_random(spawn_points, len(self.agents), self.agents[0], gen)       # [E, A]
        pts = torch.tensor(spawn_points, dtype=torch.uint8)
        w.agent_pos.copy_(pts[pick][..., :2].to(w.device))

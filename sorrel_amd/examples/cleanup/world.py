"""``CleanupWorld`` (``sorrel/examples/cleanup/world.py:16-50``): three layers (objects, agents,
beams); spawn chances and the initial apple count read from the config."""
import torch

from sorrel_amd.environment import _normalise_config
from sorrel_amd.worlds import Gridworld


class CleanupWorld(Gridworld):
    def __init__(self, config, default_entity, num_envs: int = 1, device=None, seed: int = 0):
        config = _normalise_config(config)
        self.config = config
        self.object_layer, self.agent_layer_index, self.beam_layer = 0, 1, 2
        self.pollution = 0          # never updated by the reference's step loop either (world.py:29)
        super().__init__(config.env.height, config.env.width, config.env.layers, default_entity,
                         num_envs=num_envs, device=device, seed=seed)
        self.agent_layer = self.agent_layer_index
        self.mode = config.env.get("mode", "DEFAULT")
        self.max_turns = config.experiment.max_turns
        self.pollution_threshold = config.env.pollution_threshold
        self.pollution_spawn_chance = config.env.pollution_spawn_chance
        self.apple_spawn_chance = config.env.apple_spawn_chance
        self.initial_apples = config.env.initial_apples

    def measure_pollution(self) -> torch.Tensor:
        """Polluted fraction of the river of every env, ``[E]`` (world.py:40-50)."""
        kinds = [p.kind for p in self.registry.prototypes]
        is_pol = torch.tensor([k == "Pollution" for k in kinds], device=self.device)
        is_riv = torch.tensor([k in ("Pollution", "River") for k in kinds], device=self.device)
        g = self.grid.long()
        return is_pol[g].flatten(1).sum(1) / is_riv[g].flatten(1).sum(1)

"""``TagEnv`` (``sorrel/examples/tag/env.py:24-117``) on the batched engine."""
from sorrel_amd.action.action_spec import ActionSpec
from sorrel_amd.entities import EmptyEntity, Wall
from sorrel_amd.environment import Environment
from sorrel_amd.examples.tag.agents import TagAgent
from sorrel_amd.models import RandomModel
from sorrel_amd.observation.observation_spec import OneHotObservationSpec

ENTITY_LIST = ["EmptyEntity", "Wall", "It", "NotIt"]


class TagEnv(Environment):
    """config keys: ``agent.num_agents``, ``agent.vision_radius``, optional ``agent.reward_per_turn``."""

    def __init__(self, world, config, model_factory=None):
        self._model_factory = model_factory
        super().__init__(world, config)

    def setup_agents(self):
        agents = []
        for _ in range(int(self.config.agent.num_agents)):
            ospec = OneHotObservationSpec(ENTITY_LIST, full_view=False, vision_radius=int(self.config.agent.vision_radius))
            n = 1
            for d in ospec.input_size:
                n *= d
            ospec.override_input_size((n + 1,))          # one more input for the it flag (env.py:47-48)
            aspec = ActionSpec(["up", "down", "left", "right"])
            model = self._model_factory(ospec.input_size, aspec.n_actions) if self._model_factory else \
                RandomModel(ospec.input_size, aspec.n_actions)
            agents.append(TagAgent(ospec, aspec, model, reward_per_turn=self.config.agent.get("reward_per_turn", 10)))
        self.agents = agents        # the initial "it" agent of every env is drawn by the engine (env.py:66-69)

    def populate_environment(self):
        """Walls around the single layer, agents on distinct random interior cells (env.py:84-117)."""
        self.world.set_layout(layer_fill=[EmptyEntity()] * self.world.layers, layer_border=[Wall()] * self.world.layers)
        self.spawn_agents()

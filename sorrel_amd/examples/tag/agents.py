"""``TagAgent`` (``sorrel/examples/tag/agents.py:15-112``), batched."""
import copy

import torch

from sorrel_amd.agents import MovingAgent
from sorrel_amd.agents.rules import TagRule


class TagAgent(MovingAgent):
    """One agent slot of every env.  Who is "it" is per-env state (``its`` -> bool ``[E]``), kept
    by the engine as the agent's current entity type; it survives ``Environment.reset`` as in
    the reference (agents are not re-created)."""

    speculative_ok = True        # pov = the engine's row (window + the "it" flag, row_tail), get_action = model.take_action: the fast eager loop applies

    def __init__(self, observation_spec, action_spec, model, reward_per_turn=10):
        super().__init__(observation_spec, action_spec, model)
        self.reward_per_turn = reward_per_turn
        self.interaction_rule = TagRule(reward_per_turn)
        self.kind = "NotIt"          # default appearance (agents.py:23)

    def as_kind(self, kind: str) -> "TagAgent":
        """Prototype of this agent with another appearance (the engine registers both types)."""
        c = copy.copy(self)
        c.kind = kind
        return c

    @property
    def its(self) -> torch.Tensor:
        """bool ``[E]``: is this agent "it" in each env."""
        env = self._world._environment
        eng = env._ensure_engine()
        return eng.agent_state[:, self.slot] == env.compile_spec().tag_it_type

    @property
    def it(self) -> bool:
        """The reference's scalar flag: env 0."""
        return bool(self.its[0])

    def reset(self) -> None:
        self.model.reset()

    def row_tail(self, world):
        from sorrel_amd import _native as N

        return (N.TAIL_AGENT_IS_IT, None)        # pov appends [self.it]: the engine writes it behind the window itself

    def pov(self, world) -> torch.Tensor:
        """Flattened visual field + the it flag: ``[E, C*V*V + 1]`` (agents.py:57-65).  In a policy-driven turn the engine has
        written both into this agent's row already (``sgw_bind_row_tail``: no concatenation on the host)."""
        row = world._environment._pov_row(self.slot)
        if row is not None:
            return row
        image = self.observation_spec.observe(world, self)
        flat = image.reshape(image.shape[0], -1)
        return torch.cat([flat, self.its.to(flat.dtype)[:, None]], dim=1)

    def get_action(self, state: torch.Tensor) -> torch.Tensor:
        return self.model.take_action(state)

    # act(): MovingAgent.act -> one sgw_step phase; the engine runs TagRule (move, tag, reward)

    def is_done(self, world) -> bool:
        return world.is_done

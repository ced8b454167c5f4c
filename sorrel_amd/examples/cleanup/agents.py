"""``CleanupObservation`` / ``CleanupAgent`` (``sorrel/examples/cleanup/agents.py:21-177``), batched."""
import torch

from sorrel_amd.agents import MovingAgent
from sorrel_amd.agents.rules import CleanupRule
from sorrel_amd.examples.cleanup.entities import CleanBeam, ZapBeam
from sorrel_amd.observation import embedding
from sorrel_amd.observation.observation_spec import OneHotObservationSpec


class CleanupObservation(OneHotObservationSpec):
    """One-hot visual field, flattened, followed by the positional code of the agent's cell:
    ``[E, C*V*V + 4*embedding_size]`` (agents.py:21-60)."""

    def __init__(self, entity_list, full_view: bool = False, vision_radius=None, embedding_size: int = 3):
        super().__init__(entity_list, full_view, vision_radius)
        self.embedding_size = embedding_size
        if self.full_view:
            n = len(entity_list) * 21 * 31          # the reference hard-codes its 21 x 31 map here
        else:
            n = len(entity_list) * (2 * self.vision_radius + 1) ** 2
        self.input_size = (1, n + 4 * embedding_size)
        self._table = None

    def table(self, world) -> torch.Tensor:
        """``[H, W, 4 * embedding_size]`` float32: the positional code of every cell (what the engine gathers by agent position)."""
        if self._table is None or self._table.device != world.device or self._table.shape[:2] != (world.height, world.width):
            self._table = embedding.positional_embedding_table(world, (self.embedding_size, self.embedding_size)).contiguous()
        return self._table

    def pos_code(self, world, agent) -> torch.Tensor:
        yx = world.agent_pos[:, agent.slot].long()
        return self.table(world)[yx[:, 0], yx[:, 1]]

    def observe(self, world, location=None):
        if location is None:
            raise ValueError("Location must not be None for CleanupObservation.")
        slot = getattr(location, "slot", None)
        if slot is not None:          # an agent's own pov in a policy-driven turn: window and positional code are in its row already
            row = world._environment._pov_row(slot)
            if row is not None:
                return row
        image = super().observe(world, location)
        flat = image.reshape(image.shape[0], -1)
        return torch.cat([flat, self.pos_code(world, location).to(flat.dtype)], dim=1)


class CleanupAgent(MovingAgent):
    """One agent slot of every env.  ``directions`` (uint8 ``[E]``: 0 up, 1 right, 2 down, 3 left)
    is per-env state kept by the engine; it starts at 2 and survives resets (agents.py:74)."""

    speculative_ok = True        # pov = the engine's row (window + positional code, row_tail), get_action = model.take_action: the fast eager loop applies

    def __init__(self, observation_spec, action_spec, model, beam_radius: int = 3):
        super().__init__(observation_spec, action_spec, model)
        self.interaction_rule = CleanupRule(beam_radius, CleanBeam, ZapBeam)
        self.encounters = {}

    @property
    def directions(self) -> torch.Tensor:
        return self._world._environment._ensure_engine().agent_dir[:, self.slot]


    @property
    def direction(self) -> int:
        """The reference's scalar attribute: env 0."""
        if self._world is None or getattr(self._world, "agent_dir", None) is None:
            return 2          # before the engine exists: the constructor's value
        return int(self.directions[0])

    def reset(self) -> None:
        self.model.reset()

    def row_tail(self, world):
        from sorrel_amd import _native as N

        table = getattr(self.observation_spec, "table", None)
        return (N.TAIL_POSITION_TABLE, table(world)) if callable(table) else None

    def pov(self, world) -> torch.Tensor:
        return self.observation_spec.observe(world, self)

    def get_action(self, state: torch.Tensor) -> torch.Tensor:
        return self.model.take_action(state)

    # act(): MovingAgent.act -> one sgw_step phase; the engine runs CleanupRule (turn, beams, reward, move)

    def is_done(self, world) -> bool:
        return world.is_done

"""``TreasurehuntAgent`` (``sorrel/examples/treasurehunt/agents.py:15-69``), batched."""
import torch

from sorrel_amd.agents import MovingAgent


class TreasurehuntAgent(MovingAgent):
    speculative_ok = True        # pov = the flattened window, get_action = model.take_action (frame stacks excepted: checked per turn)

    def __init__(self, observation_spec, action_spec, model):
        super().__init__(observation_spec, action_spec, model)

    def reset(self) -> None:
        self.model.reset()

    def pov(self, world) -> torch.Tensor:
        """Flattened visual field of this agent in every env: ``[E, C*V*V]``."""
        image = self.observation_spec.observe(world, self)
        return image.reshape(image.shape[0], -1)

    def get_action(self, state: torch.Tensor) -> torch.Tensor:
        mem = getattr(self.model, "memory", None)
        if mem is not None and mem.n_frames > 1:
            prev = mem.current_state()                                   # [n_frames-1, E, obs]
            state = torch.cat([prev.permute(1, 0, 2).reshape(state.shape[0], -1), state], dim=1)
        return self.model.take_action(state)

    # act() is MovingAgent.act: reward read before the move, then Gridworld.move

    def is_done(self, world) -> bool:
        return world.is_done

"""Turns of agents that hold different observation / action specs (``sorrel/agents/agent.py:38-48``)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from sorrel_amd.agents.agent import Agent


class MixedSpecTurns:
    """Mixed into ``sorrel_amd.environment.Environment``."""

    # ------------------------------------------------------------------ agents that differ (sorrel/agents/agent.py:38-48)
    def _take_turn_mixed(self, eng, actions) -> None:
        """``take_turn`` for agents that hold different observation / action specs (or observe the whole map, ``full_view``): the
        entity sweep once, then agent after agent on the handle compiled from ITS specs -- its window (its radius, table and fill
        kind; or the whole layer-summed map) from the grid as the agents before it left it, then its act through its own action list
        (``Agent.transition``, ``agent.py:155-173``).  1 + 2 A launches; the fused one-launch turn needs agents that share their specs.
        ``actions`` ``[E, A]``: indices into each agent's OWN action list."""
        self.turn += 1
        for g in self._all_engines():
            g.epoch, g.turn = self.epoch, self.turn
        self._fresh_obs = None
        self._turn_windows = None
        if actions is not None:
            eng.actions.copy_(actions.to(device=eng.device, dtype=torch.uint8).reshape(eng.actions.shape))
        eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=self.turn)     # the sweep alone
        for a, agent in enumerate(self.agents):
            if actions is not None or getattr(agent.model, "device_random", False):
                g = self._agent_engine[a]
                self._mixed_window(a)
                g.step(g.actions, random_actions=actions is None, sweep=False, write_obs=False, agent_begin=a, agent_end=a + 1,
                       turn=self.turn)
            else:
                agent.transition(self.world)

    def _mixed_window(self, a: int) -> torch.Tensor:
        """Agent ``a``'s observation through its own spec, from the grid as it stands: ``[E, C, V, V]`` (or ``[E, C, H, W]`` with
        ``full_view``), rendered by its handle into the replay row its ``add_memory`` is about to fill where that applies, else into a
        tensor of its own."""
        g = self._agent_engine[a]
        ospec = self.agents[a].observation_spec
        if ospec.full_view:
            out = self._mixed_obs[a]
            if out is None or out.dtype != g.obs_dtype:
                out = None
            self._mixed_obs[a] = g.observe_full(out)
            return self._mixed_obs[a]
        shape = (g.num_envs,) + tuple(g.spec.obs_shape[1:])
        dest = self._replay_slot(a, None, g)
        if dest is None:
            dest = self._mixed_obs[a]
            if dest is None or dest.dtype != g.obs_dtype or tuple(dest.shape) != shape:
                dest = torch.zeros(shape, dtype=g.obs_dtype, device=g.device)
        self._mixed_obs[a] = dest
        g.step(g.actions, sweep=False, agent_begin=a, agent_end=a, obs_next=True, obs_next_out=dest, turn=self.turn)
        return dest.view(shape)

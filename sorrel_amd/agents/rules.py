"""Declarative agent interaction rules: what ``Agent.act`` does on the device.

``MovingAgent.act`` (reward = value of the target, then move) is the default.  An agent
class whose ``act`` does more declares it with a rule object in ``interaction_rule``;
arbitrary Python ``act`` bodies cannot run inside the step kernel."""
from __future__ import annotations


class TagRule:
    """``TagAgent.act`` of ``sorrel/examples/tag/agents.py:76-106``: move (no reward from the
    target), then an agent that is "it" tags the first adjacent agent that is not (neighbours in
    ``Location.adjacent`` order); every turn an agent that is not "it" earns ``reward_per_turn``.
    Being "it" is the agent's ``kind`` (``it_kind`` / ``notit_kind``), so it shows in observations."""

    def __init__(self, reward_per_turn=10, it_kind: str = "It", notit_kind: str = "NotIt"):
        self.reward_per_turn = reward_per_turn
        self.it_kind = it_kind
        self.notit_kind = notit_kind


class CleanupRule:
    """``CleanupAgent.act`` of ``sorrel/examples/cleanup/agents.py:92-177``: move actions turn
    the agent (even when the move fails) and move it; ``clean`` / ``zap`` place a beam entity on
    the layer above the agent -- ``beam_radius`` cells ahead and ``beam_radius`` cells ahead of
    its right and left neighbours, skipping cells that hold a ``blocked`` entity (by class
    name); the reward is the summed ``value`` over ALL layers of the target cell, read before the
    move.  The reference adds it to ``world.total_reward`` inside ``act`` and again in
    ``Agent.transition`` -- ``count_total_twice`` keeps that.

    ``clean_beam`` / ``zap_beam`` are entities or zero-argument callables (classes)."""

    def __init__(self, beam_radius: int, clean_beam, zap_beam, blocked=("Wall",), clean_action="clean",
                 zap_action="zap", count_total_twice: bool = True):
        self.beam_radius = int(beam_radius)
        self._clean, self._zap = clean_beam, zap_beam
        self.blocked = tuple(blocked)
        self.clean_action, self.zap_action = clean_action, zap_action
        self.count_total_twice = bool(count_total_twice)
        self._resolved = None

    def beams(self):
        if self._resolved is None:
            mk = lambda b: b() if callable(b) else b
            self._resolved = (mk(self._clean), mk(self._zap))
        return self._resolved

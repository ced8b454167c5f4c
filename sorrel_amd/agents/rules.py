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

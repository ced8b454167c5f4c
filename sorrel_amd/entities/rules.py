"""Declarative transition rules: the ``Entity.transition`` plugins the device can run.

In Sorrel ``Entity.transition(world)`` is arbitrary Python executed for every cell
every turn (``sorrel/environment.py:88-91``).  A GPU cannot run arbitrary Python, so
an entity class that sets ``has_transitions = True`` declares *what* its transition
does with one of the rule objects below (class attribute or instance attribute
``transition_rule``); anything else is rejected loudly when the engine is compiled.
"""
from __future__ import annotations

from typing import Callable, Sequence, Union


class TransitionRule:
    """Base class of the closed rule set."""


class SpawnRule(TransitionRule):
    """With probability ``prob`` replace the entity's own cell by one of ``choices``,
    chosen uniformly -- Treasurehunt's ``EmptyEntity.transition``
    (``sorrel/examples/treasurehunt/entities.py:69-85``).

    ``prob`` and ``choices`` may be callables of the world (the reference reads
    ``world.spawn_prob`` / ``world.values`` at transition time)."""

    def __init__(self, prob: Union[float, Callable], choices: Union[Sequence, Callable]):
        self.prob = prob
        self.choices = choices

    def resolve(self, world):
        prob = self.prob(world) if callable(self.prob) else self.prob
        choices = self.choices(world) if callable(self.choices) else self.choices
        return float(prob), list(choices)

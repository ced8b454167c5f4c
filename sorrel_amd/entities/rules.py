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


class BecomeIfRule(TransitionRule):
    """Replace the entity's own cell by ``become`` when the occupant of the same (y, x) on
    ``layer`` has one of ``kinds`` -- Cleanup's ``Pollution.transition`` (a ``CleanBeam`` on the
    beam layer turns it back into ``River``) and ``Apple.transition`` (an agent standing on it
    turns it back into ``AppleTree``), ``sorrel/examples/cleanup/entities.py:66-73,104-110``.
    ``layer=None`` makes the change unconditional.

    ``become`` is an entity or a callable of the world returning one; ``layer`` an int, the name
    of a world attribute (``"beam_layer"``) or None."""

    def __init__(self, become, layer=None, kinds: Sequence[str] = ()):
        self.become = become
        self.layer = layer
        self.kinds = tuple(kinds)
        self._resolved = None

    def resolve(self, world):
        if self._resolved is None:          # one entity object per rule, so its type id is stable
            self._resolved = self.become(world) if callable(self.become) else self.become
        layer = getattr(world, self.layer) if isinstance(self.layer, str) else self.layer
        return self._resolved, (-1 if layer is None else int(layer)), self.kinds


class AgeRule(TransitionRule):
    """The entity survives ``turns`` entity sweeps and is replaced by ``then`` on the next one --
    Cleanup's ``Beam.transition`` with its ``turn_counter`` (``sorrel/examples/cleanup/agents.py:191-210``).
    A cell stores a type id, not an object, so the counter becomes ``turns`` extra "aged" types
    that look identical; the engine compiles the chain fresh -> aged -> ... -> ``then``."""

    def __init__(self, turns: int, then):
        if turns < 0:
            raise ValueError("turns must be >= 0")
        self.turns = int(turns)
        self.then = then
        self._chains = {}

    def chain(self, proto, world):
        """[(entity, BecomeIfRule)] for ages 0..turns of ``proto``; age 0 is ``proto`` itself."""
        import copy

        key = proto.type_key()
        if key not in self._chains:
            then = self.then(world) if callable(self.then) else self.then
            stages = [proto]
            for age in range(1, self.turns + 1):
                aged = copy.copy(proto)
                aged._location = None
                aged.age = age
                stages.append(aged)
            rules = []
            for age, ent in enumerate(stages):
                nxt = stages[age + 1] if age < self.turns else then
                rule = BecomeIfRule(nxt)
                if age > 0:
                    ent.transition_rule = rule      # distinct rule object => distinct type key
                rules.append(rule)
            self._chains[key] = list(zip(stages, rules))
        return self._chains[key]

"""``Entity`` with the attribute contract of ``sorrel/entities/entity.py:9-68``."""
from __future__ import annotations

from pathlib import Path
from typing import Optional

from sorrel_amd.entities.rules import TransitionRule


class Entity:
    """Base element class.

    Attributes (identical to the reference): ``location`` (raises ``AttributeError``
    while unset), ``value`` (reward on contact, default 0), ``passable`` (default
    False), ``has_transitions`` (default False), ``kind`` (class name unless
    overridden), ``sprite``.

    Batched-engine addition: ``transition_rule`` -- the declarative form of
    ``transition()`` (see ``sorrel_amd.entities.rules``).  In the batched world a
    cell stores an entity *type* id; two entities are the same type when class,
    kind, value, passable, has_transitions and rule all agree.
    """

    _location: Optional[tuple]
    value: float
    passable: bool
    has_transitions: bool
    kind: str
    sprite: Path
    transition_rule: Optional[TransitionRule] = None

    def __init__(self):
        self._location = None
        self.value = 0
        self.passable = False
        self.has_transitions = False
        self.kind = str(self)

    def __str__(self):
        return str(self.__class__.__name__)

    def __repr__(self):
        return f"{self.__class__.__name__}(value={self.value})"

    @property
    def location(self) -> tuple:
        if self._location is None:
            raise AttributeError(f"{self.kind} location is None.")
        return self._location

    @location.setter
    def location(self, value: tuple):
        self._location = value

    def transition(self, world):
        """Entities do not have a transition function by default.  On the device only
        ``transition_rule`` runs; overriding this method without declaring a rule is
        rejected when the engine is compiled."""
        pass

    # ------------------------------------------------------------------ batched engine
    def type_key(self):
        rule = self.transition_rule if self.has_transitions else None
        return (type(self).__module__, type(self).__qualname__, self.kind, float(self.value), bool(self.passable),
                bool(self.has_transitions), id(rule) if rule is not None else None)

"""Basic entities (``sorrel/entities/basic_entities.py:13-47``); none has a transition."""
from __future__ import annotations

from sorrel_amd.entities.entity import Entity


class Wall(Entity):
    """Impassable; penalises contact (value -1)."""

    def __init__(self):
        super().__init__()
        self.value = -1


class EmptyEntity(Entity):
    """Passable empty space."""

    def __init__(self):
        super().__init__()
        self.passable = True


class Gem(Entity):
    """Passable, rewarding object."""

    def __init__(self, value):
        super().__init__()
        self.passable = True
        self.value = value

"""Cleanup's entities (``sorrel/examples/cleanup/entities.py`` and the beams of
``agents.py:191-223``) with their ``transition`` bodies written as declarative rules."""
from sorrel_amd.entities.entity import Entity
from sorrel_amd.entities.rules import AgeRule, BecomeIfRule, SpawnRule


class EmptyEntity(Entity):
    def __init__(self):
        super().__init__()
        self.passable = True


class Sand(Entity):
    """A different sprite for the empty tile: same ``kind``, so it looks empty to the agents
    (entities.py:24-33).  It stays a distinct entity type."""

    def __init__(self):
        super().__init__()
        self.passable = True
        self.kind = "EmptyEntity"


class Wall(Entity):
    pass


class Pollution(Entity):
    """Turns back into ``River`` while a cleaning beam is above it (entities.py:58-73)."""

    transition_rule = BecomeIfRule(lambda world: River(), layer="beam_layer", kinds=("CleanBeam",))

    def __init__(self):
        super().__init__()
        self.has_transitions = True


class River(Entity):
    """Pollutes with probability ``world.pollution_spawn_chance`` per turn (entities.py:44-55)."""

    transition_rule = SpawnRule(lambda world: world.pollution_spawn_chance, lambda world: [Pollution()])

    def __init__(self):
        super().__init__()
        self.has_transitions = True


class Apple(Entity):
    """Eaten (back to ``AppleTree``) while an agent stands on it (entities.py:93-110)."""

    transition_rule = BecomeIfRule(lambda world: AppleTree(), layer="agent_layer", kinds=("CleanupAgent",))

    def __init__(self):
        super().__init__()
        self.value = 1
        self.has_transitions = True


class AppleTree(Entity):
    """Grows an apple with probability ``world.apple_spawn_chance`` per turn; the pollution gate
    (``world.pollution > world.pollution_threshold``) reads an attribute the reference never
    updates, so it is evaluated once at compile time (entities.py:76-90)."""

    transition_rule = SpawnRule(
        lambda world: 0.0 if world.pollution > world.pollution_threshold else world.apple_spawn_chance,
        lambda world: [Apple()])

    def __init__(self):
        super().__init__()
        self.has_transitions = True


class Beam(Entity):
    """Beams persist for one full turn, then disappear (agents.py:191-205)."""

    transition_rule = AgeRule(1, lambda world: EmptyEntity())

    def __init__(self):
        super().__init__()
        self.has_transitions = True


class CleanBeam(Beam):
    pass


class ZapBeam(Beam):
    def __init__(self):
        super().__init__()
        self.value = -1

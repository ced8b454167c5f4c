from sorrel_amd.entities.basic_entities import EmptyEntity, Gem, Wall
from sorrel_amd.entities.entity import Entity
from sorrel_amd.entities.rules import AgeRule, BecomeIfRule, SpawnRule, TransitionRule

__all__ = ["Entity", "EmptyEntity", "Gem", "Wall", "SpawnRule", "BecomeIfRule", "AgeRule", "TransitionRule"]

from sorrel_amd.worlds.gridworld import Gridworld, World

__all__ = ["Gridworld", "World"]

from sorrel_amd.agents.agent import Agent, MovingAgent

__all__ = ["Agent", "MovingAgent"]

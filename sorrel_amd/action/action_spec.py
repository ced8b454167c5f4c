"""ActionSpec with the interface of ``sorrel/action/action_spec.py:4-47``."""
from __future__ import annotations

from typing import Dict, List, Optional


class ActionSpec:
    """Positional mapping action index <-> action name (``["up", "down", "left", "right"]``)."""

    n_actions: int
    actions: Dict[int, str]

    def __init__(self, actions: List[str]):
        self.n_actions = len(actions)
        self.actions = dict(enumerate(actions))
        self._action_to_index = {name: i for i, name in self.actions.items()}

    def get_readable_action(self, action: int) -> str:
        return self.actions[action]

    def get_action_index(self, action_str: str) -> Optional[int]:
        return self._action_to_index.get(action_str)

    @property
    def names(self) -> List[str]:
        return [self.actions[i] for i in range(self.n_actions)]

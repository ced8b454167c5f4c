"""Declarative world specification: what the Sorrel-style classes compile to.

A Sorrel world is a numpy object array of ``Entity`` instances
(``sorrel/worlds/gridworld.py:56``).  The batched engine stores one ``uint8``
*entity type id* per cell instead and keeps what the step loop reads from an
entity -- ``kind``, ``value``, ``passable``, ``has_transitions`` + transition
rule (``sorrel/entities/entity.py:29-39``) -- in per-type tables.  ``WorldSpec``
is that table set plus the observation / action / reset-layout parameters; it
converts 1:1 into the ``sgw_config`` struct of ``include/sgw.h``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from . import _native as N

RULE_NONE = N.RULE_NONE
RULE_SPAWN = N.RULE_SPAWN
RULE_BECOME_IF = N.RULE_BECOME_IF
NO_BORDER = N.NO_BORDER

_MOVES = {"up": (-1, 0), "down": (1, 0), "left": (0, -1), "right": (0, 1)}


def resolve_device(device=None):
    """``torch.device`` with an explicit index: ``"cuda"`` / ``None`` mean the current HIP device (tensors report
    ``cuda:0``, so a bare ``cuda`` would never compare equal), ``None`` falls back to the CPU without a GPU."""
    import torch

    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    device = torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def alloc_grid(num_envs: int, layers: int, height: int, width: int, device):
    """``uint8 [E, L, H, W]`` view whose env stride is padded to a multiple of 16 bytes, so that
    worlds of any byte count (e.g. the tutorial's 21x21x2 = 882 B) can use the 16-byte load/store
    kernels.  Index it like a dense tensor; ``.cpu().numpy()`` gives a dense copy."""
    import torch

    cells = layers * height * width
    stride = (cells + 15) // 16 * 16
    storage = torch.zeros((num_envs, stride), dtype=torch.uint8, device=device)
    return storage[:, :cells].view(num_envs, layers, height, width)


def action_deltas(action_names: Sequence[str]):
    """MovingAgent.movement (``sorrel/agents/agent.py:187-213``): only the four
    names move; any other action name leaves the agent where it is."""
    dy = [_MOVES.get(n, (0, 0))[0] for n in action_names]
    dx = [_MOVES.get(n, (0, 0))[1] for n in action_names]
    return dy, dx


@dataclass
class WorldSpec:
    height: int
    width: int
    layers: int
    num_agents: int
    vision_radius: int
    num_channels: int
    agent_layer: int
    default_type: int
    fill_type: int
    action_dy: List[int]
    action_dx: List[int]
    agent_type: List[int]
    type_value: List[float]
    type_passable: List[int]
    type_rule: List[int]
    spawn_prob: List[float]
    spawn_choices: List[List[int]]
    appearance: np.ndarray                      # float64 [num_types][num_channels]
    seed: int = 0
    layer_fill_type: List[int] = field(default_factory=list)
    layer_border_type: List[int] = field(default_factory=list)
    dense_prob: float = 0.0
    dense_choices: List[int] = field(default_factory=list)
    type_names: List[str] = field(default_factory=list)   # debugging only
    obs_post: int = 0   # N.OBS_POST_*: 1 = RGBObservationSpec's clip(0, 255) / 255
    agent_rule: int = 0          # N.AGENT_RULE_*
    tag_it_type: int = 0
    tag_notit_type: int = 0
    tag_reward: float = 0.0
    rule_layer: List[int] = field(default_factory=list)      # RULE_BECOME_IF, per type
    rule_mask: List[int] = field(default_factory=list)
    rule_become: List[int] = field(default_factory=list)
    action_kind: List[int] = field(default_factory=list)     # AGENT_RULE_CLEANUP
    beam_radius: int = 0
    clean_beam_type: int = 0
    zap_beam_type: int = 0
    beam_block_mask: int = 0
    reward_total_factor: int = 1

    @property
    def num_types(self) -> int:
        return len(self.type_value)

    @property
    def num_actions(self) -> int:
        return len(self.action_dy)

    @property
    def window(self) -> int:
        return 2 * self.vision_radius + 1

    @property
    def obs_shape(self):
        return (self.num_agents, self.num_channels, self.window, self.window)

    def grid_bytes_per_env(self) -> int:
        return self.layers * self.height * self.width

    def algorithmic_bytes_per_env_step(self) -> int:
        """grid u8 read+write, per agent obs f32 store + action +
        reward + position load/store, total_reward f64 load+store."""
        v2 = self.window * self.window
        return 2 * self.grid_bytes_per_env() + self.num_agents * (self.num_channels * v2 * 4 + 1 + 4 + 4) + 16

    def to_config(self, num_envs: int, first_env_id: int = 0) -> N.SgwConfig:
        T = self.num_types
        if T > N.MAX_TYPES:
            raise ValueError(f"{T} entity types exceed the engine limit of {N.MAX_TYPES}")
        if self.num_agents > N.MAX_AGENTS:
            raise ValueError(f"{self.num_agents} agents exceed the engine limit of {N.MAX_AGENTS}")
        if self.num_channels > N.MAX_CHANNELS:
            raise ValueError(f"{self.num_channels} channels exceed the engine limit of {N.MAX_CHANNELS}")
        if self.num_actions > N.MAX_ACTIONS:
            raise ValueError(f"{self.num_actions} actions exceed the engine limit of {N.MAX_ACTIONS}")
        if self.layers > 8:
            raise ValueError("too many layers")
        c = N.SgwConfig()
        c.height, c.width, c.layers = self.height, self.width, self.layers
        c.num_agents, c.vision_radius = self.num_agents, self.vision_radius
        c.num_types, c.num_channels, c.num_actions = T, self.num_channels, self.num_actions
        c.agent_layer, c.default_type, c.fill_type = self.agent_layer, self.default_type, self.fill_type
        c.obs_post = int(self.obs_post)
        c.agent_rule, c.tag_it_type, c.tag_notit_type = int(self.agent_rule), int(self.tag_it_type), int(self.tag_notit_type)
        c.tag_reward = float(self.tag_reward)
        for t in range(T):
            c.rule_layer[t] = int(self.rule_layer[t]) if t < len(self.rule_layer) else 0
            c.rule_mask[t] = int(self.rule_mask[t]) & 0xFFFFFFFF if t < len(self.rule_mask) else 0
            c.rule_become[t] = int(self.rule_become[t]) if t < len(self.rule_become) else 0
        for i in range(self.num_actions):
            c.action_kind[i] = int(self.action_kind[i]) if i < len(self.action_kind) else 0
        c.beam_radius, c.clean_beam_type, c.zap_beam_type = int(self.beam_radius), int(self.clean_beam_type), int(self.zap_beam_type)
        c.beam_block_mask, c.reward_total_factor = int(self.beam_block_mask) & 0xFFFFFFFF, int(self.reward_total_factor)
        for i in range(self.num_actions):
            c.action_dy[i], c.action_dx[i] = int(self.action_dy[i]), int(self.action_dx[i])
        for a in range(self.num_agents):
            c.agent_type[a] = int(self.agent_type[a])
        app = np.asarray(self.appearance, dtype=np.float64)
        for t in range(T):
            c.type_value[t] = float(self.type_value[t])
            c.type_passable[t] = 1 if self.type_passable[t] else 0
            c.type_rule[t] = int(self.type_rule[t])
            c.spawn_prob[t] = float(self.spawn_prob[t])
            ch = list(self.spawn_choices[t]) if t < len(self.spawn_choices) else []
            if len(ch) > N.MAX_CHOICES:
                raise ValueError("too many spawn choices")
            c.spawn_count[t] = len(ch)
            for k, v in enumerate(ch):
                c.spawn_choice[t][k] = int(v)
            for k in range(self.num_channels):
                c.appearance[t][k] = float(app[t, k])
        for z in range(self.layers):
            c.layer_fill_type[z] = int(self.layer_fill_type[z]) if z < len(self.layer_fill_type) else self.default_type
            c.layer_border_type[z] = int(self.layer_border_type[z]) if z < len(self.layer_border_type) else NO_BORDER
        c.dense_prob = float(self.dense_prob)
        c.dense_count = len(self.dense_choices)
        for k, v in enumerate(self.dense_choices):
            c.dense_choice[k] = int(v)
        c.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        c.first_env_id = int(first_env_id)
        c.num_envs = int(num_envs)
        return c


def treasurehunt_spec(height: int, width: int, num_agents: int, vision_radius: int, spawn_prob: float = 0.005,
                      seed: int = 0, gem_value=10, food_value=5, bone_value=-10, dense_prob: float = 0.0) -> WorldSpec:
    """The canonical synthetic world of SURVEY.md 8(d): Sorrel's Treasurehunt
    example (``sorrel/examples/treasurehunt``): two layers, Sand below, walls
    around the top layer, spawning ``EmptyEntity`` inside, 6 observation
    channels ``[EmptyEntity, Wall, Gem, Bone, Food, TreasurehuntAgent]``
    (``env.py:43-50``), 4 actions (``env.py:79``)."""
    names = ["Sand", "EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]
    channel = [0, 0, 1, 2, 3, 4, 5]
    app = np.zeros((7, 6), dtype=np.float64)
    for t, k in enumerate(channel):
        if k != 0:      # kind "EmptyEntity" is the zero vector (observation_spec.py:168-169)
            app[t, k] = 1.0
    dy, dx = action_deltas(["up", "down", "left", "right"])
    return WorldSpec(
        height=height, width=width, layers=2, num_agents=num_agents, vision_radius=vision_radius,
        num_channels=6, agent_layer=1, default_type=1, fill_type=2, action_dy=dy, action_dx=dx,
        agent_type=[6] * num_agents,
        type_value=[0, 0, -1, gem_value, bone_value, food_value, 0],
        type_passable=[1, 1, 0, 1, 1, 1, 0],
        type_rule=[0, RULE_SPAWN, 0, 0, 0, 0, 0],
        spawn_prob=[0, spawn_prob, 0, 0, 0, 0, 0],
        spawn_choices=[[], [3, 5, 4], [], [], [], [], []],
        appearance=app, seed=seed, layer_fill_type=[0, 1], layer_border_type=[NO_BORDER, 2],
        dense_prob=dense_prob, dense_choices=[3, 5, 4], type_names=names,
    )

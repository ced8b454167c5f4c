"""Batched replay ring on the device: the consumer of the step's outputs
(``sorrel/buffers.py:11-154``).

One ring slot holds one turn of one agent for ALL ``num_envs`` envs, so ``add`` is a
single device-to-device copy of the kernel's output tensors (no 617 MB/step PCIe
round trip).  dtypes follow the reference: states float32, actions int64, rewards
float32, dones float32 (``sorrel/buffers.py:31-34``)."""
from __future__ import annotations

from typing import Sequence

import torch

from sorrel_amd.spec import resolve_device


class Buffer:
    def __init__(self, capacity: int, obs_shape: Sequence[int], n_frames: int = 1, num_envs: int = 1, device=None):
        self.capacity, self.obs_shape, self.n_frames, self.num_envs = capacity, tuple(obs_shape), n_frames, num_envs
        self.device = resolve_device(device)
        E = num_envs
        self.states = torch.zeros((capacity, E, *self.obs_shape), dtype=torch.float32, device=self.device)
        self.actions = torch.zeros((capacity, E), dtype=torch.int64, device=self.device)
        self.rewards = torch.zeros((capacity, E), dtype=torch.float32, device=self.device)
        self.dones = torch.zeros((capacity, E), dtype=torch.float32, device=self.device)
        self.idx = 0
        self.size = 0

    def add(self, obs, action, reward, done):
        """Append one turn: ``obs [E, *obs_shape]``, ``action [E]``, ``reward [E]``, ``done`` scalar or ``[E]``."""
        i = self.idx
        self.states[i].copy_(obs.reshape(self.states[i].shape))
        self.actions[i].copy_(action)
        self.rewards[i].copy_(reward)
        self.dones[i] = done
        self.idx = (self.idx + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def add_empty(self):
        self.idx = (self.idx + self.n_frames - 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def clear(self):
        for t in (self.states, self.actions, self.rewards, self.dones):
            t.zero_()
        self.idx = self.size = 0

    def getidx(self):
        return self.idx

    def current_state(self) -> torch.Tensor:
        """The last ``n_frames - 1`` stored observations ``[n_frames-1, E, *obs_shape]``
        (``sorrel/buffers.py:143-154``), wrapping around the ring."""
        k = self.n_frames - 1
        if k == 0:
            return self.states[0:0]
        sel = [(self.idx - k + j) % self.capacity for j in range(k)]
        return self.states[sel]

    def sample(self, batch_size: int, starts=None, envs=None):
        """Uniform sample of (turn, env) pairs with ``n_frames`` stacking:
        states, actions, rewards, next_states, dones, valid (``sorrel/buffers.py:98-124``).
        ``starts`` / ``envs`` override the random draws (first frame index and env of each sample)."""
        hi = max(1, self.size - self.n_frames - 1)
        t0 = torch.randint(0, hi, (batch_size,)) if starts is None else torch.as_tensor(starts, dtype=torch.long)
        e = (torch.randint(0, self.num_envs, (batch_size,)) if envs is None else torch.as_tensor(envs, dtype=torch.long)).to(self.device)
        idx = (t0[:, None] + torch.arange(self.n_frames)[None, :]).to(self.device)        # [B, n_frames]
        ee = e[:, None].expand_as(idx)
        states = self.states[idx, ee].reshape(batch_size, -1)
        next_states = self.states[idx + 1, ee].reshape(batch_size, -1)
        last = idx[:, -1]
        actions = self.actions[last, e].reshape(batch_size, -1)
        rewards = self.rewards[last, e].reshape(batch_size, -1)
        dones = self.dones[last, e].reshape(batch_size, -1)
        valid = (1.0 - (self.dones[idx[:, :-1], ee[:, :-1]] != 0).any(dim=-1).float()).reshape(batch_size, -1)
        return states, actions, rewards, next_states, dones, valid

    def __repr__(self):
        return f"Buffer(capacity={self.capacity}, obs_shape={self.obs_shape}, num_envs={self.num_envs})"


class TurnBuffer:
    """Joint ring over ALL agents for fused rollouts: one slot holds one ``take_turn`` of every env --
    ``obs [capacity, E, A, *obs_shape]`` float32 (or uint8 for the compact format), ``actions`` uint8,
    ``rewards`` float32 ``[capacity, E, A]``.  ``Environment.collect`` points the step kernel's observation
    output at the slot, so the 617 MB of a config-3 turn are written once, by the kernel, where the learner reads
    them (the per-agent ``Buffer`` above gets a device-to-device copy per agent instead).  ``agent_view(a)``
    gives the reference's per-agent layout back as views."""

    def __init__(self, capacity: int, num_envs: int, obs_shape: Sequence[int], device=None, obs_dtype=torch.float32):
        self.capacity, self.num_envs, self.obs_shape = capacity, num_envs, tuple(obs_shape)    # obs_shape = (A, C, V, V)
        self.device = resolve_device(device)
        A = self.obs_shape[0]
        self.obs = torch.zeros((capacity, num_envs, *self.obs_shape), dtype=obs_dtype, device=self.device)
        self.actions = torch.zeros((capacity, num_envs, A), dtype=torch.uint8, device=self.device)
        self.rewards = torch.zeros((capacity, num_envs, A), dtype=torch.float32, device=self.device)
        self.dones = torch.zeros((capacity, num_envs, A), dtype=torch.float32, device=self.device)   # all-zero inside an epoch
        self.idx = 0
        self.size = 0

    def slot(self) -> int:
        """The slot the next turn goes to (``commit`` advances)."""
        return self.idx

    def commit(self, actions: torch.Tensor, rewards: torch.Tensor) -> None:
        i = self.idx
        self.actions[i].copy_(actions)
        self.rewards[i].copy_(rewards)
        self.idx = (self.idx + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def agent_view(self, a: int):
        """(states ``[capacity, E, C, V, V]``, actions, rewards, dones ``[capacity, E]``) of one agent slot: views."""
        return self.obs[:, :, a], self.actions[:, :, a], self.rewards[:, :, a], self.dones[:, :, a]

    def clear(self):
        self.idx = self.size = 0

    def __len__(self):
        return self.size

"""Batched replay ring on the device: the consumer of the step's outputs
(``sorrel/buffers.py:11-154``).

One ring slot holds one turn of one agent for ALL ``num_envs`` envs, so ``add`` is a
single device-to-device copy of the kernel's output tensors (no 617 MB/step PCIe
round trip).  dtypes follow the reference: states float32, actions int64, rewards
float32, dones float32 (``sorrel/buffers.py:31-34``)."""
from __future__ import annotations

from pathlib import Path
from typing import Sequence

import numpy as np
import torch

from sorrel_amd.spec import resolve_device


class Buffer:
    """``extra`` keyword arguments declare additional int64 columns exactly as in the reference
    (``Buffer(capacity, obs_shape, positions=(2,))``, ``sorrel/buffers.py:39-44``): a tuple gives the trailing
    shape, anything else a scalar column; ``add(..., positions=...)`` fills them."""

    def __init__(self, capacity: int, obs_shape: Sequence[int], n_frames: int = 1, num_envs: int = 1, device=None,
                 **extra):
        self.capacity, self.obs_shape, self.n_frames, self.num_envs = capacity, tuple(obs_shape), n_frames, num_envs
        self.device = resolve_device(device)
        E = num_envs
        self.states = torch.zeros((capacity, E, *self.obs_shape), dtype=torch.float32, device=self.device)
        self.actions = torch.zeros((capacity, E), dtype=torch.int64, device=self.device)
        self.rewards = torch.zeros((capacity, E), dtype=torch.float32, device=self.device)
        self.dones = torch.zeros((capacity, E), dtype=torch.float32, device=self.device)
        self.idx = 0
        self.size = 0
        self._prefilled = None        # (row, data_ptr of the action tensor): sgw_act already wrote that action into actions[row]
        self._dones_dirty = False     # some dones row may be non-zero (a done was stored, or rows were copied / loaded in): until then
                                      # the rows are all zero already and add(done=False) has nothing to write
        self._deferred = False        # a captured policy turn is being recorded / replayed: the engine's own kernels fill the row
        self._deferred_adds = 0       # (device-side row count, sgw_turn_end) -- add() only keeps the host's idx / size in step
        self._prev_rows = None        # ... and current_state() is gathered on the device by that same count (sgw_turn_prev_rows)
        self.extra_data = {}
        for key, value in extra.items():
            shape = (capacity, E, *value) if isinstance(value, tuple) else (capacity, E)
            self.extra_data[key] = torch.zeros(shape, dtype=torch.int64, device=self.device)

    def add(self, obs, action, reward, done, **extra):
        """Append one turn: ``obs [E, *obs_shape]``, ``action [E]``, ``reward [E]``, ``done`` scalar or ``[E]``.
        Whatever the kernels have already written where it belongs is not copied again: the state (a window rendered
        straight into this row), the reward (``sgw_act``'s ``reward_row``) and the action (``sgw_act``'s ``action_row``,
        announced through ``_prefilled``)."""
        if self._deferred:
            self._deferred_adds += 1
            self.idx = (self.idx + 1) % self.capacity
            self.size = min(self.size + 1, self.capacity)
            return
        i = self.idx
        row = self.states[i]
        src = obs.reshape(row.shape)
        if src.data_ptr() != row.data_ptr():       # (the step kernel may have written the state straight into this row)
            row.copy_(src)
        pre, self._prefilled = self._prefilled, None
        if not (pre is not None and pre[0] == i and torch.is_tensor(action) and action.data_ptr() == pre[1]):
            self.actions[i].copy_(action)
        if not (torch.is_tensor(reward) and reward.data_ptr() == self.rewards[i].data_ptr() and reward.dtype == torch.float32):
            self.rewards[i].copy_(reward)
        if torch.is_tensor(done) or done:
            self.dones[i] = done
            self._dones_dirty = True
        elif self._dones_dirty:
            self.dones[i] = 0
        for key, value in extra.items():
            self.extra_data[key][i] = torch.as_tensor(value, device=self.device)
        self.idx = (self.idx + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def add_batch(self, obs, actions, rewards, done=False) -> None:
        """``k`` consecutive ``add`` calls at once -- ``obs [k, E, *obs_shape]``, ``actions`` / ``rewards`` ``[k, E]`` -- as the agents
        that share this buffer would have made them in list order (``sorrel/buffers.py:46-63``: row ``idx``, ``idx + 1``, ...,
        wrapping around): three copies instead of ``3 k``."""
        k = int(obs.shape[0])
        if k > self.capacity:
            raise ValueError(f"add_batch of {k} rows into a ring of {self.capacity}")
        if self._deferred:
            raise RuntimeError("add_batch inside a recorded turn")
        per_row = torch.is_tensor(done) and done.dim() == 2          # [k, E]: a flag per row; a scalar or [E] broadcasts over the rows
        if per_row and tuple(done.shape) != (k, self.num_envs):        # (checked before anything is written: the ring stays consistent)
            raise ValueError(f"done must be a scalar, [{self.num_envs}] or [{k}, {self.num_envs}]; got {tuple(done.shape)}")
        first = min(k, self.capacity - self.idx)
        for lo, hi, at in ((0, first, self.idx), (first, k, 0)):
            if hi <= lo:
                continue
            n = hi - lo
            if obs[lo].data_ptr() != self.states[at].data_ptr():          # (the windows may have been rendered straight into these rows)
                self.states[at:at + n].copy_(obs[lo:hi].reshape((n,) + tuple(self.states.shape[1:])))
            self.actions[at:at + n].copy_(actions[lo:hi])
            self.rewards[at:at + n].copy_(rewards[lo:hi])
            if torch.is_tensor(done) or done:
                self.dones[at:at + n] = done[lo:hi] if per_row else done     # (a wrap-around splits the rows: each segment takes ITS flags)
                self._dones_dirty = True
            elif self._dones_dirty:
                self.dones[at:at + n] = 0
        self.idx = (self.idx + k) % self.capacity
        self.size = min(self.size + k, self.capacity)

    def add_from_buffer(self, buffer: "Buffer") -> None:
        """Append the first ``min(capacity - idx, buffer.size)`` rows of another buffer, the reference's
        ``add_from_buffer`` exactly (``sorrel/buffers.py:71-99``): no wrap-around, ``idx`` only advances, ``size``
        is left alone; extra columns the source carries are created on demand."""
        if tuple(self.obs_shape) != tuple(buffer.obs_shape):
            raise AssertionError("Cannot add from a buffer with different state shapes.")
        n = min(self.capacity - self.idx, buffer.size)
        lo, hi = self.idx, self.idx + n
        self.states[lo:hi].copy_(buffer.states[:n])
        self.actions[lo:hi].copy_(buffer.actions[:n])
        self.rewards[lo:hi].copy_(buffer.rewards[:n])
        self.dones[lo:hi].copy_(buffer.dones[:n])
        for key, value in buffer.extra_data.items():
            if key not in self.extra_data:
                self.extra_data[key] = torch.zeros((self.capacity, *value.shape[1:]), dtype=value.dtype, device=self.device)
            self.extra_data[key][lo:hi].copy_(value[:n])
        self.idx = hi
        self._dones_dirty = True      # (the copied rows may hold terminal flags: later add(done=False) must clear them)

    # -- files: the reference's ``Buffer.save`` / ``Buffer.load`` format (``sorrel/buffers.py:168-201``)
    def _file_arrays(self) -> dict:
        """One env: exactly the reference's arrays (all ``capacity`` rows, scalar ``idx``).  Several envs: the rows
        of env ``e`` follow those of env ``e - 1`` (each block = that env's ring in storage order, what the reference
        would have saved for that one world), plus ``num_envs`` so that ``load`` can fold them back; a reader that
        does not know about batches (the reference's ``Buffer.load``) sees one long valid buffer."""
        E = self.num_envs

        def flat(t):
            t = t.detach().cpu()
            return t.transpose(0, 1).reshape((E * self.capacity,) + tuple(t.shape[2:])).contiguous().numpy()

        out = dict(states=flat(self.states), actions=flat(self.actions), rewards=flat(self.rewards), dones=flat(self.dones),
                   n_frames=self.n_frames, idx=self.idx if E == 1 else E * self.idx)
        for key, value in self.extra_data.items():
            out[key] = flat(value)
        if E > 1:
            out["num_envs"] = E
        return out

    def save(self, output_file) -> None:
        arrays = self._file_arrays()
        arrays = {k: v for k, v in arrays.items() if k in ("states", "actions", "rewards", "dones", "n_frames", "idx", "num_envs")}
        np.savez_compressed(Path(output_file), **arrays)

    @classmethod
    def load(cls, input_file, device="cpu") -> "Buffer":
        """Reads files written by this class or by the reference's ``Buffer.save`` / ``SavedGames.save``; like the
        reference, the loaded buffer counts as full (``size = len(states)`` per env)."""
        with np.load(Path(input_file)) as data:
            arrays = {k: data[k] for k in data.files}
        E = int(arrays.pop("num_envs", 1))
        n_frames, idx = int(arrays.pop("n_frames")), int(arrays.pop("idx"))
        rows = len(arrays["actions"]) // E
        extra = {k: tuple(v.shape[1:]) or None for k, v in arrays.items() if k not in ("states", "actions", "rewards", "dones")}
        out = cls(capacity=rows, obs_shape=arrays["states"].shape[1:], n_frames=n_frames, num_envs=E, device=device,
                  **{k: (v if v is not None else 0) for k, v in extra.items()})

        def fold(a, like):
            t = torch.from_numpy(np.ascontiguousarray(a)).reshape((E, rows) + tuple(a.shape[1:])).transpose(0, 1)
            return t.to(device=like.device, dtype=like.dtype).contiguous()

        out.states, out.actions = fold(arrays["states"], out.states), fold(arrays["actions"], out.actions)
        out.rewards, out.dones = fold(arrays["rewards"], out.rewards), fold(arrays["dones"], out.dones)
        for k in extra:
            out.extra_data[k] = fold(arrays[k], out.extra_data[k])
        out.idx = idx if E == 1 else idx // E
        out.size = rows
        out._dones_dirty = True       # loaded rows may hold terminal flags
        return out

    def __len__(self):
        return self.size

    def __getitem__(self, i):
        return self.states[i], self.actions[i], self.rewards[i], self.dones[i]

    def add_empty(self):
        self.idx = (self.idx + self.n_frames - 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def clear(self):
        for t in (self.states, self.actions, self.rewards, self.dones, *self.extra_data.values()):
            t.zero_()
        self.idx = self.size = 0
        self._dones_dirty = False

    def getidx(self):
        return self.idx

    def current_state(self) -> torch.Tensor:
        """The last ``n_frames - 1`` stored observations ``[n_frames-1, E, *obs_shape]``
        (``sorrel/buffers.py:143-154``), wrapping around the ring."""
        k = self.n_frames - 1
        if k == 0:
            return self.states[0:0]
        if self._deferred and self._prev_rows is not None:       # a recorded turn: the engine gathers by its own row count
            return self._prev_rows()
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


class SavedGames(Buffer):
    """The container ``generate_memories`` fills and writes (``sorrel/buffers.py:358-379``): a ``Buffer`` whose
    ``save`` also stores the extra columns (``positions``)."""

    def save(self, output_file) -> None:
        np.savez_compressed(Path(output_file), **self._file_arrays())

    def add_turns(self, states, actions, rewards, dones, **extra) -> None:
        """Append ``T`` turns at once from ``[T, E, ...]`` tensors (views of a ``TurnBuffer``); truncates at capacity
        like ``add_from_buffer``."""
        n = min(self.capacity - self.idx, states.shape[0])
        lo, hi = self.idx, self.idx + n
        self.states[lo:hi].copy_(states[:n].reshape((n,) + tuple(self.states.shape[1:])))
        self.actions[lo:hi].copy_(actions[:n])
        self.rewards[lo:hi].copy_(rewards[:n])
        self.dones[lo:hi].copy_(dones[:n])
        for key, value in extra.items():
            if value is not None and key in self.extra_data:
                self.extra_data[key][lo:hi].copy_(value[:n])
        self.idx = hi
        self.size = min(self.size + n, self.capacity)
        self._dones_dirty = True


class TurnBuffer:
    """Joint ring over ALL agents for fused rollouts: one slot holds one ``take_turn`` of every env --
    ``obs [capacity, E, A, *obs_shape]`` float32 (or uint8 for the compact format), ``actions`` uint8,
    ``rewards`` float32 ``[capacity, E, A]``.  ``Environment.collect`` points the step kernel's observation
    output at the slot, so the 617 MB of a config-3 turn are written once, by the kernel, where the learner reads
    them (the per-agent ``Buffer`` above gets a device-to-device copy per agent instead).  ``agent_view(a)``
    gives the reference's per-agent layout back as views."""

    def __init__(self, capacity: int, num_envs: int, obs_shape: Sequence[int], device=None, obs_dtype=torch.float32,
                 positions: bool = False):
        self.capacity, self.num_envs, self.obs_shape = capacity, num_envs, tuple(obs_shape)    # obs_shape = (A, C, V, V)
        self.device = resolve_device(device)
        A = self.obs_shape[0]
        self.obs = torch.zeros((capacity, num_envs, *self.obs_shape), dtype=obs_dtype, device=self.device)
        self.actions = torch.zeros((capacity, num_envs, A), dtype=torch.uint8, device=self.device)
        self.rewards = torch.zeros((capacity, num_envs, A), dtype=torch.float32, device=self.device)
        self.dones = torch.zeros((capacity, num_envs, A), dtype=torch.float32, device=self.device)   # all-zero inside an epoch
        # optional: every agent's (y, x) AFTER its move, what add_memory stores as ``positions`` (sorrel/agents/agent.py:127-130)
        self.positions = torch.zeros((capacity, num_envs, A, 2), dtype=torch.uint8, device=self.device) if positions else None
        self.idx = 0
        self.size = 0

    def slot(self) -> int:
        """The slot the next turn goes to (``commit`` advances)."""
        return self.idx

    def commit(self, actions: torch.Tensor, rewards: torch.Tensor, agent_pos: torch.Tensor = None) -> None:
        i = self.idx
        self.actions[i].copy_(actions)
        self.rewards[i].copy_(rewards)
        if self.positions is not None and agent_pos is not None:
            self.positions[i].copy_(agent_pos)
        self.idx = (self.idx + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def advance(self, n: int) -> None:
        """``n`` consecutive slots starting at ``idx`` were filled in place (``Environment.collect`` through
        ``sgw_rollout``); the caller guarantees they do not wrap."""
        self.idx = (self.idx + n) % self.capacity
        self.size = min(self.size + n, self.capacity)

    def agent_view(self, a: int):
        """(states ``[capacity, E, C, V, V]``, actions, rewards, dones ``[capacity, E]``) of one agent slot: views."""
        return self.obs[:, :, a], self.actions[:, :, a], self.rewards[:, :, a], self.dones[:, :, a]

    def clear(self):
        self.idx = self.size = 0

    def __len__(self):
        return self.size

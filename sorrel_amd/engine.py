"""GridEngine: owns the state tensors of E environments and drives the HIP kernels.

PyTorch is plumbing here (device memory, streams); all step/observe/reset
arithmetic runs in ``sorrel_amd/csrc/`` (``sgw.hip`` + the kernel headers it includes) through the C ABI of
``include/sgw.h``.  There is no CPU path: constructing an engine on a non-GPU
device raises.
"""
from __future__ import annotations

import ctypes as C
import itertools
from typing import Optional

import torch

from . import _native as N
from .spec import WorldSpec, alloc_grid, resolve_device


class _NoSwitch:
    """Context manager that does nothing: the engine's device is already the current one."""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


class GridEngine:
    """State + kernels for ``num_envs`` independent worlds on one GPU.

    Tensors (all on ``device``, per-env contiguous):
      grid ``uint8 [E,L,H,W]`` · agent_pos ``uint8 [E,A,2]`` · actions ``uint8 [E,A]``
      obs ``float32 [E,A,C,V,V]`` · rewards ``float32 [E,A]`` · total_reward ``float64 [E]``
    """

    def __init__(self, spec: WorldSpec, num_envs: int, device="cuda", first_env_id: int = 0,
                 allocate_obs: bool = True, tensors: Optional[dict] = None, obs_dtype=torch.float32):
        self.spec = spec
        self.num_envs = int(num_envs)
        self.first_env_id = int(first_env_id)
        self.device = resolve_device(device)
        if self.device.type != "cuda":
            raise N.SgwError(
                "GridEngine needs a HIP device (torch device 'cuda'); the step/observe path has no CPU fallback"
            )
        if not torch.cuda.is_available():
            raise N.SgwError("no HIP device visible to PyTorch; the step/observe path has no CPU fallback")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._lib = N.load()
        self.uid = next(GridEngine._uids)        # names this engine in caches (id() values are reused once an object is gone)
        if GridEngine._pending_closes and not torch.cuda.is_current_stream_capturing():
            GridEngine.drain_pending_closes()    # (engines dropped inside an earlier capture: their handles are destroyed no later than here)
        E, A = self.num_envs, spec.num_agents
        dev = self.device
        tensors = tensors or {}
        # the grid may be a view with a padded env stride (spec.alloc_grid); inner dims must be dense
        g = tensors.get("grid")
        if g is None:
            g = alloc_grid(E, spec.layers, spec.height, spec.width, dev)
        hw = spec.height * spec.width
        if tuple(g.shape) != (E, spec.layers, spec.height, spec.width) or g.dtype != torch.uint8 or g.device != dev \
                or (E > 0 and tuple(g.stride()[1:]) != (hw, spec.width, 1)) or (E > 1 and g.stride(0) < spec.layers * hw):
            raise ValueError("grid must be uint8 [E, L, H, W] on the engine's device with dense (L, H, W) strides")
        self.grid = g
        self.config = spec.to_config(self.num_envs, self.first_env_id)
        cells = spec.layers * hw
        if E > 1:
            self.config.grid_env_stride = int(g.stride(0))
        else:   # one env: any stride is right; claim the padded one if the allocation really has the pad bytes
            pad = (cells + 15) // 16 * 16
            avail = g.untyped_storage().nbytes() - g.storage_offset()
            self.config.grid_env_stride = pad if avail >= pad else 0
        self._h = C.c_void_p()
        with self._on_device():
            N.check(self._lib.sgw_create(C.byref(self.config), C.byref(self._h)))

        def adopt(name, shape, dtype):
            t = tensors.get(name)
            if t is None:
                return torch.zeros(shape, dtype=dtype, device=dev)
            if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev or not t.is_contiguous():
                raise ValueError(f"tensor {name!r} must be contiguous {dtype} {tuple(shape)} on {dev}")
            return t

        # state tensors may be adopted from the caller (the batched Gridworld owns them)
        self.agent_pos = adopt("agent_pos", (E, A, 2), torch.uint8)
        self.actions = adopt("actions", (E, A), torch.uint8)
        self.rewards = adopt("rewards", (E, A), torch.float32)
        self.total_reward = adopt("total_reward", (E,), torch.float64)
        self.metrics = torch.zeros((4,), dtype=torch.float64, device=dev)
        # float32 is the contract format (what the reference's replay buffer stores); uint8 is a compact
        # extra for one-hot specs (same layout, same values, 4x fewer bytes)
        if obs_dtype not in (torch.float32, torch.uint8):
            raise ValueError("obs_dtype must be torch.float32 or torch.uint8")
        self.obs_dtype = obs_dtype
        if obs_dtype == torch.uint8:
            N.check(self._lib.sgw_set_obs_format(self._h, N.OBS_U8))
        self.obs = (torch.zeros((E,) + spec.obs_shape, dtype=obs_dtype, device=dev) if allocate_obs else None)
        self.epoch = 0
        self.turn = 0
        self.row_tail = 0               # elements bind_row_tail appends behind every window of a row
        self._tail_table = None
        self._scratch_obs = None
        self.max_turns = 0              # set_auto_reset: epoch length (0 = no auto-reset)
        self.episode_return = None      # [E] float64: total_reward of the epoch that just ended
        # per-env agent state (current entity type of each agent): needed by interaction rules (Tag)
        self.agent_state = self.state_at_pov = None
        if spec.agent_rule == N.AGENT_RULE_TAG:
            self.agent_state = adopt("agent_state", (E, A), torch.uint8)
            self.state_at_pov = torch.zeros((E, A), dtype=torch.uint8, device=dev)
            with self._on_device():
                N.check(self._lib.sgw_bind_agent_state(self._h, self._ptr(self.agent_state), self._ptr(self.state_at_pov)))
                if "agent_state" not in tensors:
                    N.check(self._lib.sgw_init_agent_state(self._h, self._ptr(self.agent_state), self._stream()))
        # per-env facing of each agent, 0 up / 1 right / 2 down / 3 left (Cleanup's beams)
        self.agent_dir = None
        if spec.agent_rule == N.AGENT_RULE_CLEANUP:
            self.agent_dir = tensors.get("agent_dir")
            if self.agent_dir is None:
                self.agent_dir = torch.full((E, A), 2, dtype=torch.uint8, device=dev)   # CleanupAgent.__init__: facing down (agents.py:74); like Tag's flag it survives resets
            elif tuple(self.agent_dir.shape) != (E, A) or self.agent_dir.dtype != torch.uint8 or self.agent_dir.device != dev \
                    or not self.agent_dir.is_contiguous():
                raise ValueError("agent_dir must be contiguous uint8 [E, A] on the engine's device")
            N.check(self._lib.sgw_bind_agent_dir(self._h, self._ptr(self.agent_dir)))

    # ------------------------------------------------------------------ util
    def _stream(self):
        """The caller's current stream on the engine's device (raw handle; the private fast accessor PyTorch's own
        compilers use where it exists -- the public one costs ~5 us per call, a third of a small-batch step)."""
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return C.c_void_p(raw(self._dev_index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self):
        """Context that makes the engine's device current for a library call; a no-op object when it already is (the
        usual case -- entering ``torch.cuda.device`` costs several microseconds per call)."""
        return _NO_SWITCH if torch.cuda.current_device() == self._dev_index else torch.cuda.device(self.device)

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]):
        return C.c_void_p(0 if t is None else t.data_ptr())

    def _check_obs(self, t: torch.Tensor, name: str) -> torch.Tensor:
        """An observation destination handed to a kernel as a raw pointer must be exactly what the kernel writes:
        ``E * A * C * V * V`` elements of the engine's observation dtype, contiguous, on the engine's device --
        anything else would be an out-of-bounds device write."""
        want = (self.num_envs,) + tuple(self.spec.obs_shape)
        if not torch.is_tensor(t) or t.dtype != self.obs_dtype or tuple(t.shape) != want or t.device != self.device \
                or not t.is_contiguous():
            got = (tuple(t.shape), t.dtype, t.device) if torch.is_tensor(t) else type(t)
            raise ValueError(f"{name} must be a contiguous {self.obs_dtype} tensor of shape {want} on {self.device} "
                             f"(got {got}); the step kernel writes exactly that many bytes")
        return t

    def _check_window(self, t: torch.Tensor, name: str, tail: int = 0) -> torch.Tensor:
        """One window per env (``SGW_STEP_OBS_NEXT_PACKED``): exactly ``E * C * V * V`` contiguous elements of the engine's
        observation dtype on its device, whatever the trailing shape (``[E, C, V, V]`` or a replay row ``[E, C*V*V]``)."""
        per_env = 1
        for d in self.spec.obs_shape[1:]:
            per_env *= int(d)
        per_env += tail
        if not torch.is_tensor(t) or t.dtype != self.obs_dtype or t.device != self.device or not t.is_contiguous() \
                or t.dim() < 2 or t.shape[0] != self.num_envs or t.numel() != self.num_envs * per_env:
            got = (tuple(t.shape), t.dtype, t.device) if torch.is_tensor(t) else type(t)
            raise ValueError(f"{name} must be a contiguous {self.obs_dtype} tensor with {self.num_envs} rows of {per_env} "
                             f"elements on {self.device} (got {got}); the step kernel writes exactly that many bytes")
        return t

    def _check_pos(self, t: torch.Tensor) -> torch.Tensor:
        want = (self.num_envs, self.spec.num_agents, 2)
        if not torch.is_tensor(t) or t.dtype != torch.uint8 or tuple(t.shape) != want or t.device != self.device \
                or not t.is_contiguous():
            raise ValueError(f"pos must be a contiguous uint8 tensor of shape {want} on {self.device}")
        return t

    _pending_closes: list = []        # (library, handle) of engines dropped while a stream was capturing: destroyed by drain_pending_closes()
    _uids = itertools.count(1)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            h, self._h = self._h, C.c_void_p()
            # sgw_destroy frees device memory, which HIP forbids while a stream of the process is capturing (the capture fails and torch
            # aborts while it unwinds): an engine that dies inside a capture -- the collector's doing or a plain refcount -- waits
            try:
                capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
            except Exception:
                capturing = False
            if capturing:
                GridEngine._pending_closes.append((self._lib, h))
            else:
                self._lib.sgw_destroy(h)

    @classmethod
    def drain_pending_closes(cls) -> int:
        """Destroy the engines whose ``close()`` came while a stream was capturing.  Called before and after ``capture_turn``'s capture."""
        n = 0
        while cls._pending_closes:
            lib, h = cls._pending_closes.pop()
            lib.sgw_destroy(h)
            n += 1
        return n

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ ops
    def reset(self, epoch: Optional[int] = None):
        """create_world + populate_environment for every env (K3)."""
        if epoch is not None:
            self.epoch = int(epoch)
        self.turn = 0
        with self._on_device():
            N.check(self._lib.sgw_reset(self._h, self._ptr(self.grid), self._ptr(self.agent_pos),
                                        self._ptr(self.total_reward), self.epoch, self._stream()))

    def observe(self, agent_begin: int = 0, agent_end: Optional[int] = None, out: Optional[torch.Tensor] = None,
                pos: Optional[torch.Tensor] = None):
        """Stateless egocentric observation of the agents (K1).  ``pos`` (uint8 ``[E, A, 2]``)
        observes from other cells than the agents' own."""
        out = self.obs if out is None else self._check_obs(out, "out")
        if out is None:
            raise ValueError("engine was built with allocate_obs=False; pass `out`")
        pos = self.agent_pos if pos is None else self._check_pos(pos)
        agent_end = self.spec.num_agents if agent_end is None else agent_end
        with self._on_device():
            N.check(self._lib.sgw_observe(self._h, self._ptr(self.grid), self._ptr(pos), self._ptr(out),
                                          agent_begin, agent_end, self._stream()))
        return out

    # ------------------------------------------------------------------ policy-driven turns without re-rendering
    def capabilities(self) -> int:
        """``N.CAP_OBSERVE_ROWS`` | ``N.CAP_ACT`` (include/sgw.h)."""
        return int(self._lib.sgw_capabilities(self._h))

    def window_rows(self, dests=None):
        """The per-agent window destinations ``sgw_observe_rows`` / ``sgw_act`` take: ``(ctypes pointer array, env stride in
        elements, the tensors)``.  ``dests`` = one tensor per agent, each ``num_envs`` contiguous rows of exactly one window
        (``C*V*V`` elements of the engine's observation dtype, e.g. a row of that agent's replay buffer); default = the
        slots of the observation tensor ``self.obs[:, a]``."""
        A = self.spec.num_agents
        per_env = 1
        for d in self.spec.obs_shape[1:]:
            per_env *= int(d)
        arr = (C.c_void_p * A)()
        if dests is None:
            if self.obs is None:
                raise ValueError("engine was built with allocate_obs=False; pass per-agent destinations")
            base, esz = self.obs.data_ptr(), self.obs.element_size()
            for a in range(A):
                arr[a] = base + a * per_env * esz
            return arr, A * per_env, None
        if len(dests) != A:
            raise ValueError(f"need one destination per agent ({A}), got {len(dests)}")
        for a, t in enumerate(dests):
            arr[a] = self._check_window(t, f"dests[{a}]", self.row_tail).data_ptr()
        return arr, per_env + self.row_tail, list(dests)

    def sweep_observe_rows(self, rows, sweep: bool = True, turn: Optional[int] = None):
        """Steps 1 and 2 of a policy-driven turn in ONE launch (``sgw_sweep_observe_rows``, ``CAP_SWEEP_ROWS``): the entity sweep, then every
        agent's window of the swept grid into its own destination (``rows`` from ``window_rows``, exactly one window per env and row)."""
        arr, stride, _ = rows
        turn = self.turn if turn is None else turn
        with self._on_device():
            N.check(self._lib.sgw_sweep_observe_rows(self._h, self._ptr(self.grid), self._ptr(self.agent_pos), arr, stride, self.epoch, turn,
                                                     N.STEP_SWEEP if sweep else 0, self._stream()))

    def observe_rows(self, rows, agent_begin: int = 0, agent_end: Optional[int] = None):
        """Every agent's window, once, into its own destination (``rows`` from ``window_rows``): step 2 of a policy-driven
        turn.  Needs ``CAP_OBSERVE_ROWS``; otherwise ``observe()`` into the observation tensor does the same job."""
        arr, stride, _ = rows
        agent_end = self.spec.num_agents if agent_end is None else agent_end
        with self._on_device():
            N.check(self._lib.sgw_observe_rows(self._h, self._ptr(self.grid), self._ptr(self.agent_pos), arr, stride,
                                               agent_begin, agent_end, self._stream()))

    _ACTION_KINDS = {torch.uint8: N.ACT_U8, torch.int32: N.ACT_I32, torch.int64: N.ACT_I64}

    def is_action_values(self, action) -> bool:
        """A policy's VALUE output -- float32 ``[E, n_actions]`` -- which ``act`` / ``turn_act`` take as it is (``SGW_ACT_QF32``: the
        kernel picks the first index of each row's maximum, and explores by ``turn_epsilon`` under the turn protocol)."""
        return torch.is_tensor(action) and action.dim() == 2 and action.dtype == torch.float32 and action.device == self.device \
            and tuple(action.shape) == (self.num_envs, self.spec.num_actions) and action.is_contiguous()

    def _action_arg(self, action):
        """(pointer, SGW_ACT_* kind) of a caller's action tensor: ``[E]`` uint8 / int32 / int64, or float32 ``[E, n_actions]`` action values."""
        if action is None:
            return 0, 0
        if self.is_action_values(action):
            return action.data_ptr(), N.ACT_QF32
        kind = self._ACTION_KINDS.get(action.dtype)
        if kind is None or action.device != self.device or action.numel() != self.num_envs or not action.is_contiguous():
            raise ValueError(f"action must be a contiguous uint8 / int32 / int64 tensor of {self.num_envs} elements -- or float32 action values "
                             f"[{self.num_envs}, {self.spec.num_actions}] -- on {self.device}")
        return action.data_ptr(), kind

    def act(self, agent: int, rows=None, action: Optional[torch.Tensor] = None, reward_row: Optional[torch.Tensor] = None,
            action_row: Optional[torch.Tensor] = None):
        """``MovingAgent.act`` of ONE agent for every env; the at most two cells the move changes are rewritten in the
        windows (``rows``) of the agents after it that contain them, so that each later agent's window shows the grid after
        the moves of all agents before it, without being rendered again.  The agent's actions are ``self.actions[:, agent]``
        or, with ``action``, that ``[E]`` tensor as it is (uint8 / int32 / int64, contiguous, on the device: a policy's
        output needs no narrowing copy; ``self.actions[:, agent]`` still records what was taken).  Rewards land in
        ``self.rewards[:, agent]`` and, with ``reward_row`` (float32 ``[E]``) / ``action_row`` (int64 ``[E]``), a second
        time in those rows -- e.g. the rows of the agent's replay buffer."""
        arr, stride = (None, 0) if rows is None else (rows[0], rows[1])
        E, dev = self.num_envs, self.device
        prr = par = 0
        pa, kind = self._action_arg(action)
        if reward_row is not None:
            if reward_row.dtype != torch.float32 or reward_row.device != dev or reward_row.numel() != E or not reward_row.is_contiguous():
                raise ValueError(f"reward_row must be a contiguous float32 tensor of {E} elements on {dev}")
            prr = reward_row.data_ptr()
        if action_row is not None:
            if action_row.dtype != torch.int64 or action_row.device != dev or action_row.numel() != E or not action_row.is_contiguous():
                raise ValueError(f"action_row must be a contiguous int64 tensor of {E} elements on {dev}")
            par = action_row.data_ptr()
        # (plain integers for the pointers: this call sits in the per-agent loop of a policy turn, where the host is the
        # bottleneck below ~16 k envs)
        with self._on_device():
            rc = self._lib.sgw_act(self._h, self.grid.data_ptr(), self.agent_pos.data_ptr(), self.actions.data_ptr(), arr, stride,
                                   self.rewards.data_ptr(), self.total_reward.data_ptr(), int(agent), pa or None, kind,
                                   prr or None, par or None, self._stream())
        if rc:
            N.check(rc)
        return self.rewards[:, agent] if reward_row is None else reward_row

    def bind_row_tail(self, kind: int, table: Optional[torch.Tensor] = None):
        """``sgw_bind_row_tail``: what the agents' ``pov`` appends behind the flattened window, written by ``observe_rows`` (and kept
        current by ``act``) instead of a ``torch.cat`` on the host: ``N.TAIL_AGENT_IS_IT`` (Tag: one element) or
        ``N.TAIL_POSITION_TABLE`` with ``table`` float32 ``[H, W, tail_len]`` on the device (Cleanup's positional code).  Rows passed
        to ``window_rows`` then hold ``C*V*V + tail`` elements per env."""
        n = 0
        if kind == N.TAIL_POSITION_TABLE:
            if not torch.is_tensor(table) or table.dtype != torch.float32 or table.device != self.device or not table.is_contiguous() \
                    or table.dim() != 3 or tuple(table.shape[:2]) != (self.spec.height, self.spec.width):
                raise ValueError(f"table must be a contiguous float32 [H, W, tail_len] tensor on {self.device}")
            n = int(table.shape[2])
        elif kind == N.TAIL_AGENT_IS_IT:
            n = 1
        if self.obs_dtype != torch.float32 and kind != N.TAIL_NONE:
            raise ValueError("row tails are float32 (the observation format must be float32)")
        with self._on_device():
            N.check(self._lib.sgw_bind_row_tail(self._h, int(kind), n, self._ptr(table) if kind == N.TAIL_POSITION_TABLE else None))
        self.row_tail, self._tail_table = n, table

    # ------------------------------------------------------------------ a whole policy turn as one capturable submission
    def turn_bind(self, rings=None):
        """``sgw_turn_bind``: the agents' replay rings -- one ``(states, rewards, actions, dones, row, step)`` per agent (or ``None``
        for an agent that keeps no replay rows): ``states [capacity, E, >= C*V*V]`` of the engine's observation dtype, ``rewards``
        float32 / ``actions`` int64 / ``dones`` float32 ``[capacity, E]`` (each may be ``None``; ``dones`` is zeroed for the turn's
        row), ``row`` = the ring row the NEXT turn fills, ``step`` = rows the ring advances per turn.  ``rings=None`` unbinds.
        Blocking; not for the inside of a capture."""
        A, E = self.spec.num_agents, self.num_envs
        self._turn_rings = None
        if rings is None:
            with self._on_device():
                N.check(self._lib.sgw_turn_bind(self._h, None))
            return
        if len(rings) != A:
            raise ValueError(f"need one ring (or None) per agent ({A}), got {len(rings)}")
        rows = N.SgwTurnRows()
        keep = []
        for a, ring in enumerate(rings):
            if ring is None:
                continue
            states, rewards, actions, dones, row, step = ring
            cap = None
            for name, t, dt in (("states", states, self.obs_dtype), ("rewards", rewards, torch.float32), ("actions", actions, torch.int64),
                                ("dones", dones, torch.float32)):
                if t is None:
                    continue
                if not torch.is_tensor(t) or t.dtype != dt or t.device != self.device or not t.is_contiguous() or t.dim() < 2 or t.shape[1] != E:
                    raise ValueError(f"rings[{a}].{name} must be a contiguous {dt} tensor [capacity, {E}, ...] on {self.device}")
                if cap is not None and t.shape[0] != cap:
                    raise ValueError(f"rings[{a}]: states / rewards / actions disagree on the capacity")
                cap = int(t.shape[0])
            if cap is None:
                continue
            if states is not None:
                rows.states[a] = states.data_ptr()
                rows.row_elems[a] = states[0, 0].numel()
            if rewards is not None:
                if rewards[0].numel() != E:
                    raise ValueError(f"rings[{a}].rewards must be [capacity, {E}]")
                rows.rewards[a] = rewards.data_ptr()
            if actions is not None:
                if actions[0].numel() != E:
                    raise ValueError(f"rings[{a}].actions must be [capacity, {E}]")
                rows.actions[a] = actions.data_ptr()
            if dones is not None:
                if dones[0].numel() != E:
                    raise ValueError(f"rings[{a}].dones must be [capacity, {E}]")
                rows.dones[a] = dones.data_ptr()
            rows.capacity[a], rows.row[a], rows.step[a] = cap, int(row), int(step)
            keep.append((states, rewards, actions, dones))
        self._turn_rings = keep            # (the device state holds raw pointers into these)
        with self._on_device():
            N.check(self._lib.sgw_turn_bind(self._h, C.byref(rows)))

    def turn_set(self, epoch: Optional[int] = None, turn: Optional[int] = None):
        """``sgw_turn_set``: the turn the device has counted up to (stream-ordered; ``Environment.reset`` -> ``(epoch, 0)``)."""
        with self._on_device():
            N.check(self._lib.sgw_turn_set(self._h, self.epoch if epoch is None else int(epoch), self.turn if turn is None else int(turn),
                                           self._stream()))

    def turn_begin(self, sweep: bool = True):
        """``sgw_turn_begin``: ``turn += 1`` on the device, then the sweep and EVERY agent's window into ``self.obs``."""
        if self.obs is None:
            raise ValueError("turn_begin needs the observation tensor (allocate_obs=True)")
        with self._on_device():
            rc = self._lib.sgw_turn_begin(self._h, self.grid.data_ptr(), self.agent_pos.data_ptr(), self.actions.data_ptr(), self.obs.data_ptr(),
                                          self.rewards.data_ptr(), self.total_reward.data_ptr(), N.STEP_SWEEP if sweep else 0, self._stream())
        if rc:
            N.check(rc)
        return self.obs

    def turn_act(self, agent: int, action: Optional[torch.Tensor] = None):
        """``sgw_turn_act``: ``act`` of one agent with the windows in ``self.obs``; reward and int64 action also go to the agent's ring
        row of the turn in flight (by the device's own row count)."""
        pa, kind = self._action_arg(action)
        with self._on_device():
            rc = self._lib.sgw_turn_act(self._h, self.grid.data_ptr(), self.agent_pos.data_ptr(), self.actions.data_ptr(), self.obs.data_ptr(),
                                        self.rewards.data_ptr(), self.total_reward.data_ptr(), int(agent), pa or None, kind, self._stream())
        if rc:
            N.check(rc)
        return self.rewards[:, agent]

    def turn_begin_rows(self, rows, sweep: bool = True):
        """``sgw_turn_begin_rows``: the sweep alone, then every agent's window into its row of ``rows`` (``window_rows(dests)``) AND -- by
        the device's row count -- into its replay row of the turn in flight."""
        arr, stride, _ = rows
        with self._on_device():
            rc = self._lib.sgw_turn_begin_rows(self._h, self.grid.data_ptr(), self.agent_pos.data_ptr(), self.actions.data_ptr(), self.rewards.data_ptr(),
                                               self.total_reward.data_ptr(), arr, stride, N.STEP_SWEEP if sweep else 0, self._stream())
        if rc:
            N.check(rc)

    def turn_act_rows(self, agent: int, rows, action: Optional[torch.Tensor] = None):
        """``sgw_turn_act_rows``: ``act`` of one agent, the repairs written to the later agents' rows of ``rows`` and to their replay rows."""
        arr, stride, _ = rows
        pa, kind = self._action_arg(action)
        with self._on_device():
            rc = self._lib.sgw_turn_act_rows(self._h, self.grid.data_ptr(), self.agent_pos.data_ptr(), self.actions.data_ptr(), arr, stride,
                                             self.rewards.data_ptr(), self.total_reward.data_ptr(), int(agent), pa or None, kind, self._stream())
        if rc:
            N.check(rc)
        return self.rewards[:, agent]

    def turn_end(self, commit_windows: bool = True):
        """``sgw_turn_end``: the turn's windows -> the agents' ring rows, every ring advances."""
        with self._on_device():
            rc = self._lib.sgw_turn_end(self._h, self.obs.data_ptr() if (commit_windows and self.obs is not None) else None, self._stream())
        if rc:
            N.check(rc)

    # ------------------------------------------------------------------ speculative policy turns
    def speculation_rows(self, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``[A, E, C*V*V]`` float32, agent-major: the windows the policies read in a speculative turn (``turn_resolve``).  Agents that
        share a model are one contiguous ``[A_g * E, C*V*V]`` batch of it.  ``rows``: the caller's own tensor of that shape -- e.g. the
        ``A`` rows of a shared replay ring this turn fills -- instead of the engine's scratch."""
        A, E = self.spec.num_agents, self.num_envs
        per_env = 1
        for d in self.spec.obs_shape[1:]:
            per_env *= int(d)
        if getattr(self, "_spec_state", None) is None:
            self._spec_state = torch.zeros((4, E, A), dtype=torch.uint8, device=self.device)      # env_done (the first E bytes of [0]), pristine, dirty, previous moves
            self._spec_list = torch.zeros((2, E * A), dtype=torch.int64, device=self.device)      # the dirty rows of the odd / even passes
            self._spec_ctr = torch.zeros((8,), dtype=torch.int32, device=self.device)             # ... and how many (pass & 7)
            self._spec_rows = None
            self._spec_cache = {}
        if rows is None:
            if self._spec_rows is None:
                self._spec_rows = torch.zeros((A, E, per_env), dtype=torch.float32, device=self.device)
            rows = self._spec_rows
        elif tuple(rows.shape) != (A, E, per_env) or rows.dtype != torch.float32 or rows.device != self.device or not rows.is_contiguous():
            raise ValueError(f"rows must be a contiguous float32 [{A}, {E}, {per_env}] tensor on {self.device}")
        return rows

    def gather_rows(self, flat: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """``flat[idx]`` for float32 rows (``sgw_gather_rows``: the dirty rows of a speculative pass as one contiguous batch) into a buffer
        the engine keeps; valid until the next call."""
        n, ne = int(idx.numel()), int(flat.shape[1])
        buf = getattr(self, "_gather_buf", None)
        if buf is None or buf.shape[1] != ne or buf.shape[0] < n:
            buf = self._gather_buf = torch.empty((max(n, 4096), ne), dtype=torch.float32, device=self.device)
        if flat.dtype != torch.float32 or not flat.is_contiguous() or idx.dtype != torch.int64 or not idx.is_contiguous() or idx.device != self.device:
            raise ValueError("gather_rows: float32 contiguous rows and a contiguous int64 index vector on the engine's device")
        with self._on_device():
            N.check(self._lib.sgw_gather_rows(flat.data_ptr(), ne, idx.data_ptr(), n, buf.data_ptr(), self._stream()))
        return buf[:n]

    def choose_actions(self, values: torch.Tensor, idx: Optional[torch.Tensor], epoch: int, turn: int) -> torch.Tensor:
        """``sgw_choose_actions``: the actions ``sgw_act`` would take from these action values ``[n, num_actions]`` (row k belongs to the
        (agent, env) pair ``idx[k] = agent * E + env``; ``idx`` None: ``k`` itself) -- argmax, or with probability epsilon[agent]
        (``turn_epsilon``) the engine's keyed draw for (env, ``turn``, agent).  int64 ``[n]``."""
        n = int(values.shape[0])
        if values.dim() != 2 or values.shape[1] != self.spec.num_actions or not values.is_floating_point():
            raise ValueError(f"action values must be floating point [n, {self.spec.num_actions}]; got {values.dtype} {tuple(values.shape)}")
        values = values.to(device=self.device, dtype=torch.float32).contiguous()
        if idx is not None and (idx.dtype != torch.int64 or idx.device != self.device or not idx.is_contiguous() or idx.numel() < n):
            raise ValueError("choose_actions: idx must be a contiguous int64 vector of at least n entries on the engine's device")
        out = torch.empty((n,), dtype=torch.int64, device=self.device)
        with self._on_device():
            N.check(self._lib.sgw_choose_actions(self._h, values.data_ptr(), idx.data_ptr() if idx is not None else None, n, int(epoch), int(turn),
                                                 out.data_ptr(), self._stream()))
        return out

    def verify_rows(self, played: "GridEngine", rows: torch.Tensor) -> None:
        """``sgw_verify_rows``: ``played`` is a scratch engine that has just played this turn's current actions as ONE sequential turn
        (``step(actions, sweep=False)``: its ``obs`` are the windows the agents really had, its ``state_at_pov`` Tag's flags); every row of
        ``rows`` ``[A, E, C*V*V + tail]`` that differs is rewritten and listed (``verify_count`` / ``verify_list``)."""
        A, E = self.spec.num_agents, self.num_envs
        if rows.dtype != torch.float32 or rows.device != self.device or not rows.is_contiguous() or rows.dim() != 3 or tuple(rows.shape[:2]) != (A, E):
            raise ValueError(f"rows must be a contiguous float32 [{A}, {E}, row] tensor on {self.device}")
        if getattr(self, "_verify_list", None) is None:
            self._verify_list = torch.zeros((E * A,), dtype=torch.int64, device=self.device)
            self._verify_ctr = torch.zeros((1,), dtype=torch.int32, device=self.device)
        with self._on_device():
            N.check(self._lib.sgw_verify_rows(self._h, played.obs.data_ptr(), self._ptr(played.state_at_pov), rows.data_ptr(), int(rows.shape[2]),
                                              self._verify_list.data_ptr(), self._verify_ctr.data_ptr(), self._stream()))

    def verify_count(self) -> int:
        """How many rows the last ``verify_rows`` rewrote (synchronises: the one read-back per pass)."""
        return int(self._verify_ctr.item())

    def apply_actions(self, idx: Optional[torch.Tensor], fresh: torch.Tensor, n: int) -> None:
        """``sgw_apply_actions``: ``actions[e, a] = fresh[k]`` for the k-th row index ``a * E + e`` of ``idx`` (None: k itself), k < n."""
        if fresh.dtype != torch.int64 or fresh.device != self.device or not fresh.is_contiguous() or fresh.numel() < n:
            raise ValueError("apply_actions: fresh must be a contiguous int64 vector of at least n entries on the engine's device")
        with self._on_device():
            N.check(self._lib.sgw_apply_actions(self._h, self.actions.data_ptr(), None if idx is None else idx.data_ptr(), fresh.data_ptr(), int(n), self._stream()))

    def speculation_windows(self, rows: Optional[torch.Tensor] = None, sweep_turn: Optional[int] = None) -> torch.Tensor:
        """Every agent's PRE-move window into ``speculation_rows(rows)``: ``sgw_observe_rows`` where the engine has a row kernel for the
        world (one-hot tables), else ``sgw_turn_resolve``'s render mode (any table).  ``sweep_turn``: the entity sweep of that turn runs
        first -- in the same launch where the engine has ``CAP_SWEEP_ROWS``."""
        rows = self.speculation_rows(rows)
        caps = self.capabilities()
        if caps & N.CAP_OBSERVE_ROWS and not self.row_tail:
            key = rows.data_ptr()
            wr = self._spec_cache.get(key)
            if wr is None:
                if len(self._spec_cache) > 64:
                    self._spec_cache.clear()
                wr = self._spec_cache[key] = self.window_rows([rows[a] for a in range(self.spec.num_agents)])
            if sweep_turn is not None and caps & N.CAP_SWEEP_ROWS:
                self.sweep_observe_rows(wr, sweep=True, turn=sweep_turn)
                return rows
            if sweep_turn is not None:
                self.step(self.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=sweep_turn)
            self.observe_rows(wr)
        else:
            if sweep_turn is not None:
                self.step(self.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=sweep_turn)
            self.turn_resolve(0, rows)
        return rows

    def turn_resolve(self, pass_no: int, rows: Optional[torch.Tensor] = None, new_actions: Optional[torch.Tensor] = None,
                     reward_rows: Optional[torch.Tensor] = None, action_rows: Optional[torch.Tensor] = None) -> None:
        """``sgw_turn_resolve``: pass ``pass_no`` (1, 2, ...; 0 = render the pre-move windows) of a speculative policy turn over
        ``speculation_rows(rows)``.  ``new_actions``: the policy's int64 output -- for every row (``[A * E]``, agent-major) in pass 1, for
        the previous pass's dirty rows (``spec_dirty(pass_no - 1)`` order) afterwards; it is written into ``self.actions`` first.  Envs
        without a dirty agent are committed (grid, positions, ``self.rewards``, ``total_reward``; and ``reward_rows`` float32 /
        ``action_rows`` int64 ``[A, E]``, e.g. the rows of a replay ring).  ``spec_dirty(pass_no)`` then says which rows to think about again."""
        rows = self.speculation_rows(rows)
        st = self._spec_state
        A, E = self.spec.num_agents, self.num_envs
        pn, n_new = 0, 0
        if new_actions is not None:
            if new_actions.dtype != torch.int64 or new_actions.device != self.device or not new_actions.is_contiguous() or new_actions.dim() != 1 \
                    or (pass_no == 1 and new_actions.numel() != A * E) or new_actions.numel() > A * E:
                raise ValueError(f"new_actions must be a contiguous int64 vector on {self.device} ({A * E} elements in pass 1)")
            pn, n_new = new_actions.data_ptr(), int(new_actions.numel())
        for name, t, dt in (("reward_rows", reward_rows, torch.float32), ("action_rows", action_rows, torch.int64)):
            if t is not None and (t.dtype != dt or t.device != self.device or not t.is_contiguous() or t.numel() != A * E):
                raise ValueError(f"{name} must be a contiguous {dt} tensor of [{A}, {E}] on {self.device}")
        with self._on_device():
            rc = self._lib.sgw_turn_resolve(self._h, self.grid.data_ptr(), self.agent_pos.data_ptr(), self.actions.data_ptr(), rows.data_ptr(),
                                            int(rows.shape[2]), self.rewards.data_ptr(), self.total_reward.data_ptr(), st.data_ptr(),
                                            self._spec_list.data_ptr(), self._spec_ctr.data_ptr(), pn or None, n_new,
                                            None if reward_rows is None else reward_rows.data_ptr(),
                                            None if action_rows is None else action_rows.data_ptr(), int(pass_no), self._stream())
        if rc:
            N.check(rc)

    def spec_count(self, pass_no: int) -> int:
        """How many rows pass ``pass_no`` rewrote (synchronises: the one read-back per pass)."""
        return int(self._spec_ctr[pass_no & 7].item())

    def spec_dirty(self, pass_no: int) -> torch.Tensor:
        """The rows pass ``pass_no`` rewrote, as int64 indices ``agent * E + env`` into the flattened ``[A * E, N]`` rows (synchronises: the
        host reads the count)."""
        n = int(self._spec_ctr[pass_no & 7].item())
        return self._spec_list[pass_no & 1, :n]

    def turn_prev_rows(self, agent: int, count: int, out: torch.Tensor) -> torch.Tensor:
        """``sgw_turn_prev_rows``: ``Buffer.current_state`` by the device's row count -- the ``count`` rows of the agent's bound replay
        states before the row the turn in flight fills, oldest first, into ``out`` ``[count, E, row_elems]`` (contiguous, the engine's
        observation dtype, on the device)."""
        if not torch.is_tensor(out) or out.dtype != self.obs_dtype or out.device != self.device or not out.is_contiguous() \
                or out.dim() < 2 or out.shape[0] != count or out.shape[1] != self.num_envs:
            raise ValueError(f"out must be a contiguous {self.obs_dtype} tensor [{count}, {self.num_envs}, ...] on {self.device}")
        with self._on_device():
            N.check(self._lib.sgw_turn_prev_rows(self._h, int(agent), int(count), out.data_ptr(), self._stream()))
        return out

    def turn_epsilon(self, epsilon: float, agent: int = -1):
        """``sgw_turn_epsilon``: the exploration rate of action-value acts (``turn_act*`` with float32 ``[E, n_actions]``) of one agent, or of
        every agent (``-1``): with that probability the act takes the engine's own uniform draw for (env, turn, agent) instead of the
        row's argmax.  Stream-ordered, kept on the device: a recorded turn follows it."""
        with self._on_device():
            N.check(self._lib.sgw_turn_epsilon(self._h, int(agent), float(epsilon), self._stream()))

    def turn_state(self):
        """(epoch, turn, [row per agent]) as the device has counted them (synchronising)."""
        et = (C.c_uint32 * 2)()
        rows = (C.c_int64 * N.MAX_AGENTS)()
        with self._on_device():
            N.check(self._lib.sgw_turn_state(self._h, et, rows, self._stream()))
        return int(et[0]), int(et[1]), [int(rows[a]) for a in range(self.spec.num_agents)]

    def observe_full(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The whole map as an observation, ``[E, C, H, W]`` (``ObservationSpec(full_view=True).observe``): every cell's
        appearance summed over the layers."""
        want = (self.num_envs, self.spec.num_channels, self.spec.height, self.spec.width)
        if out is None:
            out = torch.empty(want, dtype=self.obs_dtype, device=self.device)
        elif tuple(out.shape) != want or out.dtype != self.obs_dtype or out.device != self.device or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous {self.obs_dtype} tensor of shape {want} on {self.device}")
        with self._on_device():
            N.check(self._lib.sgw_observe_full(self._h, self._ptr(self.grid), self._ptr(out), self._stream()))
        return out

    def pick_obs_placement(self, candidates: int = 4, launches: int = 30) -> dict:
        """Where the observation tensor lies in device memory moves the launch time of kernels that write it with many small stores -- config
        5's walking workgroups: 83 ... 92 us for the same kernel on the same card, stable per allocation, indifferent to the offset inside one
        (``profiles/r05_c5_placement.txt``).  This allocates ``candidates`` observation tensors, times ``launches`` fused turns on each (the state
        tensors are restored afterwards) and keeps the fastest as ``self.obs``.  Returns the timings.  A no-op without an observation tensor."""
        if self.obs is None or candidates < 2:
            return {"candidates_us": [], "picked": 0}
        keep = {k: getattr(self, k).clone() for k in ("grid", "agent_pos", "actions", "rewards", "total_reward")}
        epoch, turn = self.epoch, self.turn
        tried, times = [self.obs], []
        for _ in range(candidates - 1):
            tried.append(torch.zeros_like(self.obs))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for cand in tried:
            self.obs = cand
            for _ in range(launches):           # (warm: the first launches after an idle run slower)
                self.step(random_actions=True)
            ev0.record()
            for _ in range(launches):
                self.step(random_actions=True)
            ev1.record()
            torch.cuda.synchronize(self.device)
            times.append(ev0.elapsed_time(ev1) / launches * 1e3)
        best = min(range(len(tried)), key=lambda i: times[i])
        self.obs = tried[best]
        for k, v in keep.items():
            getattr(self, k).copy_(v)
        self.epoch, self.turn = epoch, turn
        self.status()                           # (whatever the timing turns flagged is theirs)
        del tried
        torch.cuda.empty_cache()
        return {"candidates_us": [round(t, 2) for t in times], "picked": best}

    def scratch_obs(self) -> torch.Tensor:
        if self._scratch_obs is None:
            self._scratch_obs = torch.zeros((self.num_envs,) + self.spec.obs_shape, dtype=self.obs_dtype, device=self.device)
        return self._scratch_obs

    def step(self, actions: Optional[torch.Tensor] = None, *, random_actions: bool = False, sweep: bool = True,
             write_obs: bool = True, agent_begin: int = 0, agent_end: Optional[int] = None,
             turn: Optional[int] = None, advance_turn: bool = True, obs_out: Optional[torch.Tensor] = None,
             obs_next: bool = False, obs_next_out: Optional[torch.Tensor] = None, no_move: bool = False, agent_major: bool = False):
        """One ``Environment.take_turn`` for every env (K2).

        ``actions``: uint8 ``[E, A]`` chosen by a policy; or ``random_actions=True``
        to draw them on device from the counter RNG (they are stored to
        ``self.actions``).  ``obs_next=True`` (policy-driven stepping): instead of the stepped
        agents' own observations, write the observation of agent ``agent_end`` as it stands
        after their moves -- into slot ``agent_end`` of the observation tensor, or, with
        ``obs_next_out`` (contiguous, the engine's observation dtype, ``E * C * V * V`` elements:
        e.g. a row of that agent's replay buffer), straight into that one-window-per-env tensor.
        ``no_move=True``: nobody acts -- the sweep (if ``sweep``) and the windows of the agents from the grid after it
        (steps 1 + 2 of a policy-driven turn in one launch; ``act`` then moves the agents one by one).
        With ``set_auto_reset`` armed, the call that completes turn
        ``max_turns`` also resets every env for the next epoch (``self.epoch`` / ``self.turn``
        follow when the engine keeps the counters, i.e. ``turn`` is not passed)."""
        if advance_turn and turn is None:
            self.turn += 1
        t = self.turn if turn is None else int(turn)
        if actions is not None and actions is not self.actions:
            # the step always consumes self.actions (so it also records what was taken)
            self.actions.copy_(actions.to(device=self.device, dtype=torch.uint8).reshape(self.actions.shape))
        actions = self.actions
        flags = (N.STEP_SWEEP if sweep else 0) | (N.STEP_RANDOM_ACTIONS if random_actions else 0) | (N.STEP_NO_MOVE if no_move else 0)
        if agent_major:      # obs_out is [A, E, C*V*V] (SGW_STEP_OBS_AGENT_MAJOR: engines with CAP_OBS_AGENT_MAJOR, whole-turn calls)
            obs = self.speculation_rows(obs_out)
            flags |= N.STEP_OBS_AGENT_MAJOR
        else:
            obs = self.obs if obs_out is None else self._check_obs(obs_out, "obs_out")
        if obs_next_out is not None:
            if not obs_next:
                raise ValueError("obs_next_out is the destination of obs_next=True")
            obs = self._check_window(obs_next_out, "obs_next_out")
            flags |= N.STEP_OBS_NEXT | N.STEP_OBS_NEXT_PACKED
        elif obs_next:
            if obs is None:
                raise ValueError("obs_next needs an observation tensor")
            flags |= N.STEP_OBS_NEXT
        elif not write_obs or obs is None:
            flags |= N.STEP_NO_OBS
            obs = None
        agent_end = self.spec.num_agents if agent_end is None else agent_end
        epoch = self.epoch
        with self._on_device():
            N.check(self._lib.sgw_step(self._h, self._ptr(self.grid), self._ptr(self.agent_pos), self._ptr(actions),
                                       self._ptr(obs), self._ptr(self.rewards), self._ptr(self.total_reward),
                                       epoch, t, agent_begin, agent_end, flags, self._stream()))
        if self.max_turns and t == self.max_turns and agent_end == self.spec.num_agents and not no_move:
            self.epoch = epoch + 1        # the library has reset every env for the next epoch (sgw_set_auto_reset)
            if turn is None:
                self.turn = 0
        return obs, self.rewards

    def rollout(self, turns: int, *, random_actions: bool = True, actions: Optional[torch.Tensor] = None, sweep: bool = True,
                obs_out: Optional[torch.Tensor] = None, actions_out: Optional[torch.Tensor] = None,
                rewards_out: Optional[torch.Tensor] = None, write_obs: bool = True):
        """``turns`` whole ``take_turn``s with one call (``sgw_rollout``): where the kernel supports it they run inside one
        launch with every env's grid resident in LDS.  Without ``*_out`` tensors each turn overwrites ``self.obs`` /
        ``self.actions`` / ``self.rewards`` (they hold the last turn afterwards, as after ``turns`` calls of ``step``).
        ``obs_out [T, E, A, C, V, V]``, ``actions_out`` / ``rewards_out [T, E, A]`` (contiguous, the engine's dtypes)
        receive every turn -- e.g. ``T`` consecutive slots of a ``TurnBuffer``.  ``actions [T, E, A]``: scripted actions
        instead of on-device random ones."""
        T = int(turns)
        if T <= 0:
            return
        E, A = self.num_envs, self.spec.num_agents

        def per_turn(t, name, dtype, tail):
            """A caller-owned ``[T, *tail]`` tensor the kernels fill turn by turn through a raw pointer."""
            want = (T,) + tuple(tail)
            if not torch.is_tensor(t) or tuple(t.shape) != want or t.dtype != dtype or t.device != self.device or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous {dtype} tensor of shape {want} on {self.device}")
            return t, t[0].numel()

        flags = N.STEP_SWEEP if sweep else 0
        if actions is not None:                                  # scripted
            act, ts_act = per_turn(actions, "actions", torch.uint8, (E, A))
        elif not random_actions:
            raise ValueError("rollout needs random_actions=True or an actions tensor")
        else:
            flags |= N.STEP_RANDOM_ACTIONS
            act, ts_act = (self.actions, 0) if actions_out is None else per_turn(actions_out, "actions_out", torch.uint8, (E, A))
        rew, ts_rew = (self.rewards, 0) if rewards_out is None else per_turn(rewards_out, "rewards_out", torch.float32, (E, A))
        obs, ts_obs = (self.obs, 0) if obs_out is None else per_turn(obs_out, "obs_out", self.obs_dtype, (E,) + tuple(self.spec.obs_shape))
        if not write_obs or obs is None:
            flags |= N.STEP_NO_OBS
            obs = None
        with self._on_device():
            N.check(self._lib.sgw_rollout(self._h, self._ptr(self.grid), self._ptr(self.agent_pos), self._ptr(act), self._ptr(obs),
                                          self._ptr(rew), self._ptr(self.total_reward), self.epoch, self.turn + 1, T,
                                          ts_obs, ts_act, ts_rew, flags, self._stream()))
        # the engine's own per-turn tensors hold the LAST turn, as after T calls of step()
        if act is not self.actions:
            self.actions.copy_(act[-1])
        if rew is not self.rewards:
            self.rewards.copy_(rew[-1])
        self.turn += T
        while self.max_turns and self.turn >= self.max_turns:    # the library reset the batch at every epoch boundary it crossed
            self.turn -= self.max_turns
            self.epoch += 1

    def set_auto_reset(self, max_turns: int):
        """Arm (``max_turns > 0``) or disarm the in-stream reset at the end of turn ``max_turns``;
        ``self.episode_return`` then holds each env's ``total_reward`` of the epoch that just ended."""
        self.max_turns = int(max_turns)
        if self.max_turns and self.episode_return is None:
            self.episode_return = torch.zeros((self.num_envs,), dtype=torch.float64, device=self.device)
        with self._on_device():
            N.check(self._lib.sgw_set_auto_reset(self._h, self.max_turns,
                                                 self._ptr(self.episode_return if self.max_turns else None)))

    def random_actions(self, turn: Optional[int] = None):
        t = self.turn + 1 if turn is None else int(turn)
        with self._on_device():
            N.check(self._lib.sgw_random_actions(self._h, self._ptr(self.actions), self.epoch, t, self._stream()))
        return self.actions

    def reduce_metrics(self) -> torch.Tensor:
        """Device tensor ``[sum(total_reward), sum(total_reward**2), E, 0]`` (K4)."""
        with self._on_device():
            N.check(self._lib.sgw_reduce_metrics(self._h, self._ptr(self.total_reward), self._ptr(self.metrics),
                                                 self._stream()))
        return self.metrics

    def status(self) -> int:
        """Synchronising read-and-clear of the device status word."""
        v = C.c_int32(0)
        with self._on_device():
            N.check(self._lib.sgw_get_status(self._h, C.byref(v), self._stream()))
        return int(v.value)

    def raise_on_status(self):
        s = self.status()
        if s & N.STATUS_OOB_MOVE:
            raise IndexError("an agent moved off the grid: the agent layer's border must be impassable "
                             "(the reference has no bounds check in Gridworld.move)")
        if s & N.STATUS_BAD_ACTION:
            raise KeyError("action index outside the ActionSpec")
        if s & N.STATUS_BAD_TYPE:
            raise KeyError("grid holds an entity type id that was never registered")
        if s & N.STATUS_BAD_POS:
            raise IndexError("an agent position outside the grid was passed to the engine")

    # ------------------------------------------------------------------ timing
    def set_timing(self, enable: bool):
        N.check(self._lib.sgw_set_timing(self._h, 1 if enable else 0))

    def step_time_ms(self):
        ms, n = C.c_double(0.0), C.c_int64(0)
        N.check(self._lib.sgw_get_step_time_ms(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def step_times_ms(self, capacity: int = 1 << 16):
        """Per-launch durations (ms, HIP events on the launch stream) since the last read, oldest first."""
        buf = (C.c_float * capacity)()
        n = C.c_int64(0)
        rc = self._lib.sgw_get_step_times_ms(self._h, buf, capacity, C.byref(n))
        if rc < 0:
            N.check(rc)
        self.series_truncated = rc == 1          # launches beyond `capacity` / the library's sample cap were left out
        return [float(buf[i]) for i in range(int(n.value))]

    def set_wg_per_cu(self, n: int):
        """Launch tuning of the wave-per-env kernel: 0 automatic, 1..8 forced, -1 never capped (include/sgw.h)."""
        N.check(self._lib.sgw_set_wg_per_cu(self._h, int(n)))

    def launch_info(self) -> str:
        buf = C.create_string_buffer(1024)
        N.check(self._lib.sgw_launch_info(self._h, buf, 1024))
        return buf.value.decode()

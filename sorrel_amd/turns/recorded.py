"""A whole policy-driven turn as one graph replay (``Environment.capture_turn``; ``sgw_turn_*`` in include/sgw.h)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from sorrel_amd.agents.agent import Agent


class RecordedTurns:
    """Mixed into ``sorrel_amd.environment.Environment``."""

    # ------------------------------------------------------------------ a whole policy turn as one graph
    def _turn_protocol_body(self, eng) -> None:
        """One policy-driven take_turn through the device-counted protocol (include/sgw.h, sgw_turn_*): the same calls with the same
        arguments every turn -- what a graph can record."""
        rows = self._capture_rows
        if rows is not None:       # windows in per-agent rows at fixed addresses; the kernels write the replay rows alongside (no copy at the end)
            eng.turn_begin_rows(rows, sweep=True)
        else:                      # windows in the observation tensor; sgw_turn_end copies them into the replay rows
            rows = eng.window_rows(None)
            eng.turn_begin(sweep=True)
        self._turn_windows = [self.world.mutations, rows, 0, None]
        self._turn_capture = True
        try:
            for agent in self.agents:
                agent.transition(self.world)          # pov (a view of the window at its fixed address) -> get_action -> act -> add_memory (deferred)
        finally:
            self._turn_capture = False
            self._turn_windows = None
        eng.turn_end(commit_windows=self._capture_rows is None)

    #: a recorded turn writes every window twice (a fixed address for the policy + the replay row); above this many bytes of windows per
    #: turn that costs more than the host time a replay saves while there are few agents (measured: 32x32 / 8 agents, 65 536 envs = 617 MB:
    #: 610 us recorded against 500 eager; 16 384 envs = 154 MB: 249 against 385; config 5's 64 agents gain at any size) -- capture_turn()
    #: then declines unless forced
    capture_max_window_bytes = 384 << 20
    #: ... and where the eager loop is the fast one (``fast_policy_loop``: agents with the standard hooks) the crossover is lower (round 5, same
    #: shape: 16 384 envs = 154 MB: 245 us recorded against 283 eager; 24 576 envs = 231 MB: 331 against 297; 32 768: 367 against 312) --
    #: counted per agent, since the host time a replay saves grows with the agents as the windows do: 24 MB of windows per agent and turn
    capture_max_window_bytes_per_agent_fast = 24 << 20

    def capture_turn(self, warmup: int = 2, force: bool = False):
        """Record ONE whole policy-driven ``take_turn`` -- sweep + every agent's window, then per agent the policy's forward pass
        and its act, then the copy of the turn's windows into the agents' replay rows -- as a graph (``torch.cuda.graph``), so that
        every later ``take_turn()`` is one replay without Python in the agent loop (``sorrel/agents/agent.py:155-173`` costs
        1 + A engine launches and A policy calls from Python otherwise; below ~16 k envs the host is the bottleneck).

        What makes that possible: the turn number, the epoch and the replay row of every agent live in device memory that the
        engine's own kernels advance (``sgw_turn_begin`` / ``sgw_turn_end``), so the recorded launches carry no per-turn
        arguments; the policies read their windows from the observation tensor (a fixed address), and ``sgw_turn_end`` copies
        them into the ring rows ``Buffer.add`` would have filled.  Results are those of the eager loop, bit for bit.

        ``warmup`` real turns are played through the same protocol first (lazy initialisation must not happen inside a
        capture).  Returns the ``CapturedTurn``, or ``None`` -- and the eager loop stays in charge -- when the turn cannot be
        recorded: an agent class overrides ``transition`` / ``add_memory``, a model's memory is not a ``sorrel_amd.buffers.Buffer``
        of plain windows (appended features index the ring from the host; frame stacks -- ``n_frames > 1`` -- are recorded: ``current_state()``
        becomes a gather by the device's row count, ``sgw_turn_prev_rows``; agents that share such a ring need the "rows" layout), the engine has
        no observation tensor, or a model's forward pass does something a capture forbids (a host synchronisation)."""
        from sorrel_amd import _native as N
        from sorrel_amd.buffers import Buffer

        self._captured = None
        eng = self._ensure_engine()
        if self._mixed:
            self.capture_error = ValueError("agents with different observation / action specs step on separate engine handles: not recorded")
            return None
        if eng.obs is None or not self.patch_windows:
            return None
        per_env = 1
        for d in eng.spec.obs_shape[1:]:
            per_env *= int(d)
        window_bytes = eng.num_envs * len(self.agents) * per_env * (4 if eng.obs_dtype == torch.float32 else 1)
        limit = self.capture_max_window_bytes
        if self.fast_policy_loop and self._fast_plan(eng) is not None:
            limit = min(limit, self.capture_max_window_bytes_per_agent_fast * len(self.agents))
        if not force and len(self.agents) <= 16 and window_bytes > limit:
            self.capture_error = ValueError(f"{window_bytes >> 20} MiB of windows per turn: a recorded turn writes them twice, which costs more than the "
                                            "replay saves at this batch (capture_turn(force=True) records anyway)")
            return None
        # what pov appends behind the window (Tag's "it" flag, Cleanup's positional code) is the engine's to write (_bind_row_tail): the
        # rows the policies read and the replay rows then hold window + tail
        use_rows = self.capture_layout != "tensor" and bool(eng.capabilities() & N.CAP_OBSERVE_ROWS)
        if eng.row_tail and not use_rows:
            return None
        per_row = per_env + eng.row_tail
        sharers = {}
        for agent in self.agents:
            mem = getattr(agent.model, "memory", None)
            if type(agent).transition is not Agent.transition or type(agent).add_memory is not Agent.add_memory:
                return None
            if mem is None:
                continue
            if not isinstance(mem, Buffer) or mem.extra_data or mem.num_envs != eng.num_envs or mem.device != eng.device \
                    or mem.states.dtype != eng.obs_dtype or mem.states[0, 0].numel() != per_row or mem.n_frames - 1 > mem.capacity:
                return None
            sharers.setdefault(id(mem), [mem, []])[1].append(agent.slot)
        if not use_rows and any(v[0].n_frames > 1 and len(v[1]) > 1 for v in sharers.values()):
            # frame stacks of agents that share one ring interleave their rows: agent k's stack holds the windows of agents k-1, k-2 of THIS turn
            # (sorrel/buffers.py:143-154 with idx advanced by their adds).  The "rows" layout has them in the ring by then (every window is
            # rendered into its replay row at the start of the turn and repaired there); the "tensor" layout copies them at the end of the turn
            self.capture_error = ValueError("agents that share a frame-stacking ring need capture_layout = 'rows' (windows written into the ring as the turn goes)")
            return None
        buffers = [v[0] for v in sharers.values()]

        def rings():
            out = [None] * len(self.agents)
            for mem, slots in sharers.values():
                for k, a in enumerate(slots):
                    out[a] = (mem.states, mem.rewards, mem.actions, mem.dones if mem._dones_dirty else None,
                              (mem.idx + k) % mem.capacity, len(slots))
            return out

        # where the policies read their windows: per-agent rows the row kernels fill (and, alongside, the replay rows) where the engine
        # has them -- one-hot float32 windows --, else the observation tensor + a copy at the end of the turn
        self._capture_rows = None
        if use_rows:
            self._capture_rows = eng.window_rows([torch.zeros((eng.num_envs, per_row), dtype=torch.float32, device=eng.device) for _ in self.agents])
        cap = CapturedTurn(self, eng, buffers, [len(v[1]) for v in sharers.values()], rings)
        cap._stacked = [(v[0], v[1]) for v in sharers.values() if v[0].n_frames > 1]
        try:
            cap.record(max(1, int(warmup)))
        except Exception as exc:                                   # not capturable: leave everything consistent and say why
            cap.abort()
            self.capture_error = exc
            return None
        self._captured = cap
        return cap


class CapturedTurn:
    """One policy-driven ``take_turn`` recorded as a graph (``Environment.capture_turn``).  ``replay()`` plays the next turn;
    the host only keeps its counters (``Environment.turn``, every buffer's ``idx`` / ``size``) in step with the device's."""

    def __init__(self, env, eng, buffers, adds_per_turn, rings):
        self.env, self.eng, self.buffers, self.adds, self._rings = env, eng, buffers, adds_per_turn, rings
        self.graph = None
        self.turns_replayed = 0
        self._stacked = []                     # (buffer, [slot]) of the frame-stacking memories
        self._expect, self._at = None, None    # the rings' rows and (epoch, turn) the device's turn state stands at, as the host last knew them

    def valid(self, eng) -> bool:
        return self.graph is not None and eng is self.eng

    def resync(self) -> None:
        """After ``Environment.reset`` (or any host-side change of the counters): the device's turn state follows the host's."""
        self.eng.turn_bind(self._rings())
        self.eng.turn_set(self.env.epoch, self.env.turn)
        self._expect, self._at = [mem.idx for mem in self.buffers], (self.env.epoch, self.env.turn)

    def _host_step(self) -> None:
        env = self.env
        env.turn += 1
        self.eng.epoch, self.eng.turn = env.epoch, env.turn
        env._fresh_obs = None

    def record(self, warmup: int) -> None:
        env, eng = self.env, self.eng
        self.resync()
        for mem in self.buffers:
            mem._deferred, mem._deferred_adds = True, 0
        for mem, slots in self._stacked:
            # Buffer.current_state (frame stacks, n_frames > 1): gathered by the device's row count into a fixed tensor
            # (agents that share the ring: the j-th of them to ask in a turn stands at row idx + j -- its own slot's row count on the device)
            outs = [torch.zeros((mem.n_frames - 1,) + tuple(mem.states.shape[1:]), dtype=mem.states.dtype, device=mem.device) for _ in slots]
            mem._prev_rows = (lambda mem=mem, slots=slots, k=mem.n_frames - 1, outs=outs:
                              eng.turn_prev_rows(slots[mem._deferred_adds % len(slots)], k, outs[mem._deferred_adds % len(slots)]))
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):                       # real turns: they count
                self._host_step()
                env._turn_protocol_body(eng)
        torch.cuda.current_stream(eng.device).wait_stream(side)
        torch.cuda.synchronize(eng.device)
        for mem, n in zip(self.buffers, self.adds):
            if mem._deferred_adds != n * warmup:
                raise RuntimeError("an agent's add_memory did not run once per turn")
        before = [(mem.idx, mem.size) for mem in self.buffers]
        g = torch.cuda.CUDAGraph()
        # No garbage collection inside the capture: an unreachable engine or graph of an EARLIER environment that the collector happens to
        # free now would call hipFree / hipGraphDestroy while a stream is capturing, which HIP forbids -- the capture fails, and torch aborts
        # the process while it unwinds (seen under rocprofv3, where the timing differs; torch.cuda.graph no longer collects on entry itself)
        # The window: process-wide and NOT thread-safe (another thread that re-enables the collector, or drops the last reference to an engine /
        # graph between here and the end of the capture, still frees inside it) -- a capture is a single-threaded moment of the caller's program.
        # Reference-counted frees of THIS thread are kept out explicitly: engines whose close() is pending are closed now, before the capture.
        import gc
        from sorrel_amd.engine import GridEngine
        gc.collect()
        GridEngine.drain_pending_closes()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g):
                env._turn_protocol_body(eng)              # recorded, not run: the host-side effects are undone below
        except BaseException:
            import os, sys, traceback
            env.capture_error_trace = traceback.format_exc()      # (what failed INSIDE the capture; torch may abort while it unwinds the graph)
            if os.environ.get("SGW_DEBUG"):
                print(env.capture_error_trace, file=sys.stderr, flush=True)
            raise
        finally:
            if gc_was_on:
                gc.enable()
            # ... also when the capture fails half-way (a later agent's forward pass synchronises): the agents before it have already
            # counted an add_memory for rows that were never written -- the eager loop must not find them counted as valid
            for mem, (idx, size) in zip(self.buffers, before):
                mem.idx, mem.size = idx, size
                mem._deferred_adds = 0
        self.graph = g
        GridEngine.drain_pending_closes()
        self._expect, self._at = [mem.idx for mem in self.buffers], (env.epoch, env.turn)

    def abort(self) -> None:
        for mem in self.buffers:
            mem._deferred = False
            mem._prev_rows = None
        self.graph = None
        self.env._capture_rows = None
        try:
            self.eng.turn_bind(None)
        except Exception:
            pass

    def replay(self) -> None:
        if self._expect != [mem.idx for mem in self.buffers] or self._at != (self.env.epoch, self.env.turn):
            self.resync()                                 # host code moved a ring (Buffer.clear at the start of an epoch) or the counters
        self._host_step()
        if self.env._value_agents:
            self.env._push_epsilon(self.eng)              # a decaying epsilon reaches the recorded acts through the device's turn state
        self.graph.replay()
        for mem, n in zip(self.buffers, self.adds):
            mem.idx = (mem.idx + n) % mem.capacity
            mem.size = min(mem.size + n, mem.capacity)
        self._expect, self._at = [mem.idx for mem in self.buffers], (self.env.epoch, self.env.turn)
        self.turns_replayed += 1

    def release(self) -> None:
        """Back to the eager loop (the buffers copy for themselves again)."""
        self.abort()
        if self.env._captured is self:
            self.env._captured = None

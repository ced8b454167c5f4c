"""The eager policy-driven turn: every window once, then per agent the policy and ``sgw_act`` (``sorrel/agents/agent.py:155-173``)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from sorrel_amd.agents.agent import Agent


class PolicyTurns:
    """Mixed into ``sorrel_amd.environment.Environment``."""

    #: where the engine has the instance (``CAP_SWEEP_ROWS``), the sweep and every agent's window into its replay row are ONE launch
    #: (``sgw_sweep_observe_rows``); False = the sweep alone + ``sgw_observe_rows`` (A/B and tests)
    fuse_sweep_and_rows = True

    #: agents with the standard hooks (``Agent.speculative_ok``: pov = the flattened window, get_action = ``model.take_action``) and replay
    #: memories whose rows hold exactly one window are stepped by a loop that does per agent what ``Agent.transition`` does -- the model's
    #: forward pass, one ``sgw_act`` with pointers worked out once per turn, the ring's bookkeeping -- without the generic hooks' checks in
    #: between (``_FastPolicyTurn``; ~20 -> ~8 us of engine-side Python per agent).  False = the generic loop (A/B and tests).
    fast_policy_loop = True

    def _fast_plan(self, eng):
        key = (id(eng), self.patch_windows, self.write_obs_into_replay, eng.row_tail,
               tuple((id(a.model), id(getattr(a.model, "memory", None)), type(a)) for a in self.agents))
        cached = self.__dict__.get("_fast_plan_cache")
        if cached is None or cached[0] != key:
            cached = (key, _FastPolicyTurn.build(self, eng), (eng, [(a.model, getattr(a.model, "memory", None)) for a in self.agents]))   # (holds what its key names by id())
            self.__dict__["_fast_plan_cache"] = cached
        plan = cached[1]
        return plan if plan is not None and plan.still_valid() else None

    #: policy-driven turns render every agent's window once and let each act launch repair the cells its move changed
    #: (``sgw_observe_rows`` / ``sgw_act``); False = the older 1 + A protocol, a window rendered per launch (A/B and test switch)
    patch_windows = True

    def _begin_policy_turn(self, eng) -> bool:
        """Steps 1 and 2 of the patched-window protocol (``include/sgw.h``): the entity sweep alone, then EVERY agent's
        window, once, from the grid after the sweep -- into the row of each agent's replay buffer that its ``add_memory``
        is about to fill where that is possible (``_replay_rows``), else into the observation tensor.  Step 3 is
        ``_act``.  ``sgw_act`` has an instance for every agent rule (plain movers, Tag, Cleanup), so this returns False only
        when the protocol is switched off (``patch_windows = False``) or the engine has no observation tensor."""
        from sorrel_amd import _native as N

        self._turn_windows = None
        caps = eng.capabilities()
        if not self.patch_windows or not (caps & N.CAP_ACT) or eng.obs is None:
            return False
        dests = self._replay_rows() if caps & N.CAP_OBSERVE_ROWS else None
        slots = self._replay_slots if dests is not None else None      # (buffer, row) per agent
        if dests is None and eng.row_tail:      # tailed rows without replay buffers to put them in: the environment's own
            if self._tail_rows is None:
                per_env = int(np.prod(eng.spec.obs_shape[1:])) + eng.row_tail
                self._tail_rows = [torch.zeros((eng.num_envs, per_env), dtype=torch.float32, device=eng.device) for _ in self.agents]
            dests = self._tail_rows
        rows = eng.window_rows(dests)
        if dests is not None and self.fuse_sweep_and_rows and caps & N.CAP_SWEEP_ROWS:      # (round 6: row tails, Tag / Cleanup worlds and worlds above 4 KiB too)
            eng.sweep_observe_rows(rows, sweep=True, turn=self.turn)      # both in one launch (the grid read once, a burst per env)
        elif dests is not None:                 # the sweep alone, then every window into its agent's replay row
            eng.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=self.turn)
            eng.observe_rows(rows)
        else:                                   # the windows live in the observation tensor: sweep + all of them in ONE launch
            eng.step(sweep=True, no_move=True, turn=self.turn)
        self._turn_windows = [self.world.mutations, rows, 0, slots]
        return True

    def _replay_rows(self):
        """One destination per agent -- the row of its replay buffer that its next ``add_memory`` fills -- if EVERY agent
        has one: a ``sorrel_amd.buffers.Buffer`` of the engine's dtype and device whose rows hold exactly one window (a
        ``pov`` that appends to the window, like Cleanup's positional code, does not qualify); agents that share one
        buffer get consecutive rows, in the order their ``add_memory`` calls will arrive.  ``Buffer.add`` then finds the
        state in place and copies nothing (config 3 at 65 536 envs: 77 MB per agent and turn).

        Invariant this relies on: a ``pov`` is followed by the same agent's ``add_memory`` within the turn, which is what
        ``Agent.transition`` does; an agent class that overrides ``transition`` or ``add_memory`` could leave a
        pre-written row behind in a full ring, so such agents (and ``write_obs_into_replay = False``) get the
        observation tensor and ``Buffer.add`` copies."""
        from sorrel_amd.buffers import Buffer

        eng = self._engine
        if not self.write_obs_into_replay:
            return None
        per_env = 1
        for d in eng.spec.obs_shape[1:]:
            per_env *= int(d)
        taken, rows, self._replay_slots = {}, [], []
        for agent in self.agents:
            mem = getattr(agent.model, "memory", None)
            if not isinstance(mem, Buffer) or type(agent).transition is not Agent.transition \
                    or type(agent).add_memory is not Agent.add_memory:
                return None
            k = taken.get(id(mem), 0)
            taken[id(mem)] = k + 1
            if k >= mem.capacity:
                return None
            i = (mem.idx + k) % mem.capacity
            row = mem.states[i]
            if row.dtype != eng.obs_dtype or row.device != eng.device or not row.is_contiguous() or row.dim() < 2 \
                    or row.shape[0] != eng.num_envs or row.numel() != eng.num_envs * (per_env + eng.row_tail):
                return None
            rows.append(row)
            self._replay_slots.append((mem, i))
        return rows


class _FastPolicyTurn:
    """The eager policy-driven turn of agents with the standard hooks (``Environment.fast_policy_loop``): the same launches as the generic
    loop -- the sweep alone, every window into the replay row its agent's ``add_memory`` is about to fill, then per agent the model's forward
    pass and ``sgw_act`` (the act + the repair of the later agents' windows; reward and int64 action into the ring's rows) -- with
    everything that does not change from turn to turn (which checks an agent passes, base pointers, row sizes) worked out once."""

    @classmethod
    def build(cls, env, eng):
        from sorrel_amd import _native as N
        from sorrel_amd.agents.agent import MovingAgent
        from sorrel_amd.buffers import Buffer

        caps = eng.capabilities()
        if env._mixed or not env.patch_windows or not env.write_obs_into_replay or eng.obs is None \
                or eng.obs_dtype != torch.float32 or not (caps & N.CAP_ACT) or not (caps & N.CAP_OBSERVE_ROWS):
            return None
        per_env = int(np.prod(eng.spec.obs_shape[1:])) + eng.row_tail       # (the engine writes what pov appends behind the window: Tag, Cleanup)
        taken, agents = {}, []
        for agent in env.agents:
            mem = getattr(agent.model, "memory", None)
            if not env._standard_hooks(agent) or type(agent).act is not MovingAgent.act or not isinstance(mem, Buffer) or mem.n_frames != 1 \
                    or mem.extra_data or mem.num_envs != eng.num_envs or mem.device != eng.device or mem.states.dtype != torch.float32 \
                    or not mem.states.is_contiguous() or mem.states[0, 0].numel() != per_env or getattr(agent.model, "device_random", False):
                return None
            k = taken.get(id(mem), 0)                 # agents that share a ring fill consecutive rows, in list order
            taken[id(mem)] = k + 1
            agents.append((agent, agent.model, mem, k))
        if any(n > mem.capacity for (_a, _m, mem, _k), n in zip(agents, (taken[id(x[2])] for x in agents))):
            return None
        return cls(env, eng, agents, per_env)

    def __init__(self, env, eng, agents, per_env):
        import ctypes as C

        self.env, self.eng, self.agents, self.per_env = env, eng, agents, per_env
        self.A, self.E = len(agents), eng.num_envs
        self.arr = (C.c_void_p * self.A)()
        self.rows = (self.arr, per_env, None)
        self.views = {}                               # (id(states), row) -> the [E, N] view the policy reads
        self.lib = eng._lib
        self.kinds = eng._ACTION_KINDS
        from sorrel_amd import _native as N
        self.fused = bool(eng.capabilities() & N.CAP_SWEEP_ROWS)
        self.qf32, self.nact = N.ACT_QF32, eng.spec.num_actions

    def still_valid(self) -> bool:
        return not any(mem._deferred for _a, _m, mem, _k in self.agents)

    def run_turn(self) -> None:
        env, eng, E, N_ = self.env, self.eng, self.E, self.per_env
        env._turn_windows = None
        row_bytes = E * N_ * 4
        rows_i = []
        for a, (_agent, _model, mem, k) in enumerate(self.agents):
            i = (mem.idx + k) % mem.capacity
            rows_i.append(i)
            self.arr[a] = mem.states.data_ptr() + i * row_bytes
        if self.fused and env.fuse_sweep_and_rows:
            eng.sweep_observe_rows(self.rows, sweep=True, turn=env.turn)                        # the sweep + every agent's window into its replay row
        else:
            eng.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=env.turn)  # the sweep alone
            eng.observe_rows(self.rows)                                                     # every agent's window into its replay row
        h, stream = eng._h, eng._stream()
        grid, pos, acts, rew, tot = eng.grid.data_ptr(), eng.agent_pos.data_ptr(), eng.actions.data_ptr(), eng.rewards.data_ptr(), eng.total_reward.data_ptr()
        dev = eng.device
        world = env.world
        edits = world.mutations
        slots = [(m[2], r) for m, r in zip(self.agents, rows_i)]
        with eng._on_device():
            for a, (agent, model, mem, _k) in enumerate(self.agents):
                i = rows_i[a]
                key = self.arr[a]                     # (a cached view keeps its storage alive: the address cannot come to mean another tensor)
                state = self.views.get(key)
                if state is None:
                    if len(self.views) > 65536:
                        self.views.clear()
                    state = self.views[key] = mem.states[i].view(E, N_)
                action = model.take_action(state)
                if world.mutations != edits:          # the model edited the world: windows are rendered on demand from here on, by the generic hooks
                    reward = env._act(agent, action)
                    mem.add(state, eng.actions[:, a] if torch.is_tensor(action) and action.dim() == 2 else action, reward, agent.is_done(world))
                    for later, _m, _mem, _k in self.agents[a + 1:]:
                        later.transition(world)
                    return
                values = torch.is_tensor(action) and action.dim() == 2
                if values and a in env._value_agents and action.dtype == torch.float32 and action.device == dev and action.is_contiguous() \
                        and tuple(action.shape) == (E, self.nact):
                    # action VALUES: the act launch takes the argmax / explores (SGW_ACT_QF32); the draws are keyed by the turn in flight
                    if env._turn_state_at.get(eng.uid) != (env.epoch, env.turn):
                        eng.turn_set(env.epoch, env.turn - 1)
                        env._turn_state_at[eng.uid] = (env.epoch, env.turn)
                    env._push_epsilon(eng, (a,))
                    pa, kind = action.data_ptr(), self.qf32
                elif not torch.is_tensor(action) or values:               # a plain int, or an agent's FIRST action values (or odd ones): the generic act
                    env._turn_windows = [edits, self.rows, a, slots]      # knows how
                    reward = env._act(agent, action)
                    env._turn_windows = None
                    mem.add(state, eng.actions[:, a], reward, agent.is_done(world))
                    continue
                else:
                    kind = self.kinds.get(action.dtype)
                    if kind is None or action.device != dev or action.dim() != 1 or action.shape[0] != E or not action.is_contiguous():
                        eng.actions[:, a].copy_(action)
                        pa, kind = None, 0
                    else:
                        pa = action.data_ptr()
                rc = self.lib.sgw_act(h, grid, pos, acts, self.arr, N_, rew, tot, a, pa, kind, mem.rewards.data_ptr() + i * E * 4,
                                      mem.actions.data_ptr() + i * E * 8, stream)
                if rc:
                    from sorrel_amd import _native as N
                    N.check(rc)
                done = agent.is_done(world)
                if torch.is_tensor(done) or done:
                    mem.dones[i] = done
                    mem._dones_dirty = True
                elif mem._dones_dirty:
                    mem.dones[i] = 0
                mem.idx = (mem.idx + 1) % mem.capacity
                mem.size = min(mem.size + 1, mem.capacity)

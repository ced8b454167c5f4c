"""Speculative policy turns (``Environment.speculate_turns``): ``sorrel/agents/agent.py:155-173`` for many agents at once."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from sorrel_amd.agents.agent import Agent


class SpeculativeTurns:
    """Mixed into ``sorrel_amd.environment.Environment``."""

    # ------------------------------------------------------------------ many policy-driven agents: speculative turns
    #: Evaluate the policies of ALL agents on their pre-move windows in one batch per model, let the engine find the (env, agent) pairs
    #: whose window an earlier agent's move changed (``sgw_turn_resolve``) and re-evaluate only those, until nothing changes: the
    #: fixed point is the reference's agent-after-agent turn (``sorrel/agents/agent.py:155-173``) -- bit for bit when a policy is a
    #: function of its window -- in two or three batched passes instead of A dependent (forward, act) pairs.  Pays with many agents
    #: whose models are shared (one forward pass per model and pass); needs plain movers, ``Agent.speculative_ok`` agents, one-frame
    #: memories.  Off by default: the agents' ``pov`` / ``get_action`` / ``act`` hooks are not called one by one in such a turn.
    #: True: where it is possible AND the cost model below says it pays; "always": wherever it is possible.
    speculate_turns = False

    @staticmethod
    def _standard_hooks(agent) -> bool:
        """The agent's class declares (``speculative_ok``) that its ``pov`` is the flattened window of its own spec and its ``get_action`` is
        ``model.take_action`` of it -- and no class derived from the one that says so overrides a hook of the turn."""
        hooks = ("pov", "get_action", "act", "transition", "add_memory")
        if any(h in agent.__dict__ for h in hooks):       # (a hook patched onto the instance)
            return False
        for cls in type(agent).__mro__:
            if cls.__dict__.get("speculative_ok") is True:
                return True
            if "speculative_ok" in cls.__dict__ or any(h in cls.__dict__ for h in hooks):
                return False
        return False

    def _speculation_groups(self, eng):
        """``[(a0, a1, model)]``: runs of consecutive agents that share a model object -- or None when this turn must run agent after
        agent (the switch is off, an agent does not qualify, the engine cannot resolve this world)."""
        from sorrel_amd import _native as N
        from sorrel_amd.buffers import Buffer

        if not self.speculate_turns:
            return None
        # (the answer only changes with the engine, the agents' models and their memories: asked every turn, computed once -- with 64 agents
        # the checks below are ~100 us of Python)
        key = (id(eng), eng.row_tail, self.speculate_turns, self.speculation_cost_model,
               tuple((id(a.model), id(getattr(a.model, "memory", None)), type(a)) for a in self.agents))
        cached = self.__dict__.get("_spec_groups")
        if cached is not None and cached[0] == key:
            groups = cached[1]
            if groups is not None and any(getattr(m, "memory", None) is not None and m.memory._deferred for _a0, _a1, m in groups):
                return None
            return groups
        groups = self._speculation_groups_uncached(eng, N, Buffer)
        if groups is not None and self.speculate_turns != "always" and not self._speculation_pays(eng):
            groups = None
        # (the key names objects by id(): the entry holds them, so no id in it can be handed to a NEW engine / model / memory while it is cached)
        self.__dict__["_spec_groups"] = (key, groups, (eng, [(a.model, getattr(a.model, "memory", None)) for a in self.agents]))
        return groups

    #: (fixed us of a speculative turn, us per MB of windows, us of host time per agent of the sequential loop, its fixed us): the sequential
    #: loop costs ~33 us of host time per agent (two torch ops + ``sgw_act``), a speculative turn ~200 us of passes and read-backs plus device
    #: time that grows with the windows it renders, compares and re-evaluates.  Measured (one linear policy shared by all agents, wall us per
    #: turn, speculative / sequential): 8 agents 235 / 281 at 2 048 envs, 244 / 277 at 4 096, 306 / 304 at 8 192, 406 / 292 at 16 384;
    #: 16 agents 237 / 479 at 1 024, 459 / 552 at 8 192; config 5's 64 agents 546 / 2 190 at 2 048.
    speculation_cost_model = (200.0, 1.6, 33.0, 20.0)
    #: ... of the generic form (Tag, Cleanup, tailed rows): every pass plays a whole turn on the scratch state and reads all windows twice, and the
    #: examples' windows cover a fifth to most of their maps, so pass 2 re-evaluates ~60 % of the rows (profiles/r06_speculation_study_rules.txt:
    #: 2.3-2.6 passes per env on average, 99th percentile 4, 1.7 evaluations per agent-step; a BATCH needs the passes of its slowest env: 3-5).
    #: Measured, wall us per turn, generic speculative / eager (profiles/r06_speculative_generic.txt): Tag 5 agents 410 / 182 at 1 024 envs, Tag 16
    #: agents 858 / 566, Cleanup 10 agents 785 / 358, 96 plain movers 2 059 / 2 803 at 1 024 envs and 6 239 / 2 768 at 4 096 -- the shipped
    #: examples keep the eager loop; it pays for very many agents over small batches.
    speculation_cost_model_generic = (600.0, 24.0, 33.0, 20.0)

    def _speculation_pays(self, eng) -> bool:
        """``speculate_turns = True`` speculates where the model above says it is the faster turn (``"always"``: wherever it is possible)."""
        fixed, per_mb, per_agent, seq_fixed = self.speculation_cost_model_generic if getattr(self, "_spec_generic", False) else self.speculation_cost_model
        A = len(self.agents)
        per_env = int(np.prod(eng.spec.obs_shape[1:]))
        mb = eng.num_envs * A * per_env * 4 / 1e6
        return per_agent * A + seq_fixed > fixed + per_mb * mb

    def _speculation_groups_uncached(self, eng, N, Buffer):
        if self._mixed or eng.obs_dtype != torch.float32:
            return None
        # plain movers without row tails: the resolve kernel (sgw_turn_resolve).  Everything else -- Tag, Cleanup, agents beyond the 64 a wave
        # holds, tailed rows -- : the generic form (round 6), which plays the current actions as one sequential turn on a scratch copy of the
        # state and compares what the agents saw (sgw_verify_rows); it needs the windows in rows the row kernels can write
        self._spec_generic = not (eng.capabilities() & N.CAP_RESOLVE) or bool(eng.row_tail)
        if self._spec_generic and not (eng.capabilities() & N.CAP_OBSERVE_ROWS):
            return None
        per_env = int(np.prod(eng.spec.obs_shape[1:])) + (eng.row_tail if self._spec_generic else 0)
        groups = []
        for a, agent in enumerate(self.agents):
            model = agent.model
            if not self._standard_hooks(agent) or getattr(model, "device_random", False):
                return None
            mem = getattr(model, "memory", None)
            if mem is not None and (not isinstance(mem, Buffer) or mem.n_frames != 1 or mem.extra_data or mem.num_envs != eng.num_envs
                                    or mem.device != eng.device or mem.states[0, 0].numel() != per_env or mem._deferred):
                return None
            if groups and groups[-1][2] is model:
                groups[-1][1] = a + 1
            else:
                groups.append([a, a + 1, model])
        for a0, a1, model in groups:
            mem = getattr(model, "memory", None)
            if mem is not None and a1 - a0 > mem.capacity:
                return None
        if 3 * len(groups) > len(self.agents):         # (nearly) a model per agent: a pass is then as many forward passes as the sequential turn has
            return None                                # -- measured 17 ms against 3.3 ms for 64 agents with 64 models (profiles/r05_speculative_turn.txt)
        return groups

    def _spec_scratch_engine(self, eng):
        """A second handle over scratch copies of the state tensors (same spec, same global env ids): where a speculative pass plays its turn."""
        scr = self.__dict__.get("_spec_scratch")
        if scr is not None and scr[0] is eng:
            return scr[1]
        from sorrel_amd.engine import GridEngine

        t = dict(agent_pos=torch.zeros_like(eng.agent_pos), total_reward=torch.zeros_like(eng.total_reward))
        if eng.agent_state is not None:
            t["agent_state"] = eng.agent_state.clone()
        if eng.agent_dir is not None:
            t["agent_dir"] = eng.agent_dir.clone()
        played = GridEngine(eng.spec, eng.num_envs, device=eng.device, first_env_id=eng.first_env_id, tensors=t, obs_dtype=eng.obs_dtype)
        self.__dict__["_spec_scratch"] = (eng, played)
        self._aux_engines[("speculation scratch", eng.uid)] = played      # (raise_on_status polls it; closed with the others)
        return played

    def _take_turn_speculative_generic(self, eng, groups) -> None:
        """The speculative turn for any agent rule (``sgw_verify_rows``): pass 1 evaluates every agent on what its ``pov`` returns BEFORE anyone acts
        (window + tail, one batch per model); a pass then plays the current actions as ONE sequential turn on a scratch copy of the state -- the
        ordinary fused step kernel -- and compares what every agent really saw with what its action was computed on; the rows that differ are
        evaluated again, until none does.  The scratch state of that last pass is the reference's agent-after-agent turn, bit for bit."""
        E, A = eng.num_envs, len(self.agents)
        self._turn_windows = None
        played = self._spec_scratch_engine(eng)
        per_env = int(np.prod(eng.spec.obs_shape[1:])) + eng.row_tail
        rows = self.__dict__.get("_spec_rows_generic")
        if rows is None or tuple(rows.shape) != (A, E, per_env) or rows.device != eng.device:
            rows = self.__dict__["_spec_rows_generic"] = torch.zeros((A, E, per_env), dtype=torch.float32, device=eng.device)
            self.__dict__["_spec_rows_generic_wr"] = eng.window_rows([rows[a] for a in range(A)])
        wr = self.__dict__["_spec_rows_generic_wr"]
        flat = rows.view(A * E, per_env)
        from sorrel_amd import _native as N

        if eng.capabilities() & N.CAP_SWEEP_ROWS and self.fuse_sweep_and_rows:     # the sweep and every agent's PRE-act window (+ tail), one launch
            eng.sweep_observe_rows(wr, sweep=True, turn=self.turn)
        else:
            eng.step(sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=self.turn)
            eng.observe_rows(wr)

        def choose(model, x, idx):
            out = model.take_action(x)
            if out.dim() == 2:                                           # action values: the act launch's choice, exploration included (sgw_choose_actions)
                self._push_epsilon(eng, range(A))
                out = eng.choose_actions(out, idx, self.epoch, self.turn)
            return out.to(torch.int64)

        every = self.__dict__.get("_spec_arange")
        if every is None or every.numel() != A * E or every.device != eng.device:
            every = self.__dict__["_spec_arange"] = torch.arange(A * E, dtype=torch.int64, device=eng.device)
        fresh = torch.cat([choose(model, flat[a0 * E:a1 * E], every[a0 * E:a1 * E]) for a0, a1, model in groups]) if len(groups) > 1 \
            else choose(groups[0][2], flat, None)
        eng.apply_actions(None, fresh.contiguous(), A * E)
        state = [("grid", eng.grid, played.grid), ("agent_pos", eng.agent_pos, played.agent_pos), ("total_reward", eng.total_reward, played.total_reward)]
        if eng.agent_state is not None:
            state.append(("agent_state", eng.agent_state, played.agent_state))
        if eng.agent_dir is not None:
            state.append(("agent_dir", eng.agent_dir, played.agent_dir))
        k = 0
        while True:
            k += 1
            for _name, real, scratch in state:
                scratch.copy_(real)
            played.epoch = eng.epoch
            played.step(eng.actions, sweep=False, turn=self.turn)        # the whole turn, agent after agent, with the current actions
            eng.verify_rows(played, rows)
            n = eng.verify_count()                                       # (synchronises)
            if n == 0:
                break
            if k > A + 1:
                raise RuntimeError("speculative turn did not converge (a policy that is not a function of its observation?)")
            lst = eng._verify_list[:n]
            if len(groups) == 1:
                new = choose(groups[0][2], eng.gather_rows(flat, lst), lst)
            else:
                new = torch.empty_like(lst)
                a_i = torch.div(lst, E, rounding_mode="floor")
                for a0, a1, model in groups:
                    sel = torch.nonzero((a_i >= a0) & (a_i < a1)).squeeze(1)
                    if sel.numel():
                        new[sel] = choose(model, flat.index_select(0, lst[sel]), lst[sel].contiguous())
            eng.apply_actions(lst, new.contiguous(), n)
        for _name, real, scratch in state:                               # the last pass played the sequential turn: its state is the turn's
            real.copy_(scratch)
        eng.rewards.copy_(played.rewards)
        if eng.state_at_pov is not None:
            eng.state_at_pov.copy_(played.state_at_pov)
        self.speculation_passes = k
        self._spec_seen = (self.epoch, self.turn, rows)
        taken = eng.actions.t().to(torch.int64)                          # [A, E]
        rew = eng.rewards.t().contiguous()
        for a0, a1, model in groups:                                     # add_memory of every agent, in list order
            mem = getattr(model, "memory", None)
            if mem is None:
                continue
            dones = [self.agents[a].is_done(self.world) for a in range(a0, a1)]
            done = False if not any(torch.is_tensor(d) or d for d in dones) else \
                torch.stack([torch.as_tensor(d, dtype=torch.float32, device=eng.device).expand(E) for d in dones])
            mem.add_batch(rows[a0:a1], taken[a0:a1].contiguous(), rew[a0:a1], done)

    def _take_turn_speculative(self, eng, groups) -> None:
        if getattr(self, "_spec_generic", False):
            return self._take_turn_speculative_generic(eng, groups)
        E, A = eng.num_envs, len(self.agents)
        self._turn_windows = None
        # what the policies read, [A, E, N]: where ONE model (and so one replay ring) serves every agent and the ring's rows of this turn
        # are contiguous, those rows themselves -- add_memory then has nothing to copy (config 5: 381 MB of windows per turn)
        own = rrows = arows = None
        mem = getattr(groups[0][2], "memory", None) if len(groups) == 1 else None
        if mem is not None and mem.capacity % A == 0 and mem.idx % A == 0 and self.write_obs_into_replay:
            own = mem.states[mem.idx:mem.idx + A].view(A, E, -1)
            rrows, arows = mem.rewards[mem.idx:mem.idx + A], mem.actions[mem.idx:mem.idx + A]
        rows = eng.speculation_rows(own)
        flat = rows.view(A * E, -1)
        from sorrel_amd import _native as N

        if eng.capabilities() & N.CAP_OBS_AGENT_MAJOR:                   # (worlds above 4 KiB) the sweep AND every agent's PRE-move window in ONE launch
            eng.step(eng.actions, sweep=True, no_move=True, turn=self.turn, obs_out=rows, agent_major=True)
        else:
            eng.speculation_windows(own, sweep_turn=self.turn)           # the sweep, then every agent's PRE-move window, once (one launch with CAP_SWEEP_ROWS)

        def choose(model, x, idx):
            # idx: which (agent, env) pair each row of x belongs to (agent * E + env; None: the row's own number)
            out = model.take_action(x)
            if out.dim() == 2:
                # action values: what Agent.transition hands to sgw_act (SGW_ACT_QF32) -- the argmax, or with probability agent.epsilon the
                # engine's draw for (env, turn, agent) (iqn.py:294-309).  The draw is keyed, so the choice stays a function of the window
                # and the fixed point below is the sequential turn WITH its exploration.
                self._push_epsilon(eng, range(A))
                out = eng.choose_actions(out, idx, self.epoch, self.turn)
            return out.to(torch.int64)

        if len(groups) == 1:                                             # pass 1: one batch per model
            fresh = choose(groups[0][2], flat, None)
        else:
            every = self.__dict__.get("_spec_arange")
            if every is None or every.numel() != A * E or every.device != eng.device:
                every = self.__dict__["_spec_arange"] = torch.arange(A * E, dtype=torch.int64, device=eng.device)
            fresh = torch.cat([choose(model, flat[a0 * E:a1 * E], every[a0 * E:a1 * E]) for a0, a1, model in groups])
        def bucket(n):
            # a batch of a few sizes only (the BLAS picks its kernel per shape: a new shape every turn costs more than the padding)
            return 64 if n <= 64 else (1 << (n - 1).bit_length() if n <= 4096 else -(-n // 4096) * 4096)

        guess = self.__dict__.setdefault("_spec_guess", {})              # pass -> rows it left dirty the last time: how many to evaluate ahead
        k = 1
        eng.turn_resolve(1, own, fresh.contiguous(), rrows, arows)       # writes the actions, commits the envs that are at their fixed point already
        while True:
            # The host learns the dirty count with a synchronisation.  While it waits the GPU would idle, and after it the next batch's
            # launches would only start to arrive: so the rows this pass will PROBABLY leave dirty (as many as last turn, rounded up) are
            # gathered and evaluated before the count is read -- the list's entries beyond the count are older valid indices, harmless.
            ahead, m = None, 0
            if len(groups) == 1 and guess.get(k, 0) > 0:
                m = min(bucket(int(guess[k] * 1.2) + 1), A * E)
                ahead = choose(groups[0][2], eng.gather_rows(flat, eng._spec_list[k & 1, :m]), eng._spec_list[k & 1, :m])
            n = eng.spec_count(k)                                        # (synchronises)
            guess[k] = n
            if n == 0:
                break
            if ahead is not None and n <= m:
                fresh = ahead[:n]
            elif len(groups) == 1:
                m = min(bucket(n), A * E)
                pad = eng._spec_list[k & 1, :m]
                if m > n:
                    pad[n:m] = 0                                         # (row 0: evaluated again, the result thrown away)
                fresh = choose(groups[0][2], eng.gather_rows(flat, pad), pad)[:n]
            else:
                lst = eng._spec_list[k & 1, :n]
                fresh = torch.empty_like(lst)
                a_i = torch.div(lst, E, rounding_mode="floor")
                for a0, a1, model in groups:
                    sel = torch.nonzero((a_i >= a0) & (a_i < a1)).squeeze(1)
                    if sel.numel():
                        fresh[sel] = choose(model, flat.index_select(0, lst[sel]), lst[sel].contiguous())
            k += 1
            eng.turn_resolve(k, own, fresh.contiguous(), rrows, arows)
        self.speculation_passes = k
        self._spec_seen = (self.epoch, self.turn, rows)                  # (obs_of: the windows of this turn live here, not in the [E, A, ...] tensor)
        if own is not None:                                              # windows, rewards and actions already lie in the ring's rows
            done = [self.agents[a].is_done(self.world) for a in range(A)]
            if any(torch.is_tensor(d) or d for d in done):
                mem.dones[mem.idx:mem.idx + A] = torch.stack([torch.as_tensor(d, dtype=torch.float32, device=eng.device).expand(E) for d in done])
                mem._dones_dirty = True
            elif mem._dones_dirty:
                mem.dones[mem.idx:mem.idx + A] = 0
            mem.idx = (mem.idx + A) % mem.capacity
            mem.size = min(mem.size + A, mem.capacity)
            return
        taken = eng.actions.t().to(torch.int64)                          # [A, E]
        rew = eng.rewards.t().contiguous()
        for a0, a1, model in groups:                                     # add_memory of every agent, in list order
            mem = getattr(model, "memory", None)
            if mem is None:
                continue
            dones = [self.agents[a].is_done(self.world) for a in range(a0, a1)]
            done = False if not any(torch.is_tensor(d) or d for d in dones) else \
                torch.stack([torch.as_tensor(d, dtype=torch.float32, device=eng.device).expand(E) for d in dones])
            mem.add_batch(rows[a0:a1], taken[a0:a1].contiguous(), rew[a0:a1], done)

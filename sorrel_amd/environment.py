"""``Environment`` with the interface of ``sorrel/environment.py:18-93``: the batch driver.

Subclass it exactly like the reference: implement ``setup_agents`` (assign
``self.agents``) and ``populate_environment``.  ``take_turn`` advances ALL
``world.num_envs`` worlds:

* fused (one kernel launch): every agent's model is a ``RandomModel`` (actions drawn
  on device) or an ``actions [E, A]`` tensor is passed;
* policy-driven (2 + A launches): the entity sweep, EVERY agent's window rendered once, then
  for each agent in list order ``agent.transition(world)`` = observe -> policy -> act, where
  the act launch (``sgw_act``) moves the agent and rewrites, in the windows of the agents
  after it, the at most two cells its move changed -- so agent i+1 observes agent i's move
  exactly as in the reference (``sorrel/agents/agent.py:155-173``) without a window being
  rendered per launch.  ``sgw_act`` serves every agent rule (``MovingAgent.act``, ``TagAgent.act``,
  ``CleanupAgent.act``); ``patch_windows = False`` selects the older 1 + A form, in which the launch
  that moves agent i also renders agent i+1's observation (``SGW_STEP_OBS_NEXT``);
* recorded (``capture_turn()``): that whole policy-driven turn as ONE graph replay -- the turn number
  and every agent's replay row live in device memory the engine advances itself (``sgw_turn_*``).

The device status word (off-grid move, bad action index, unregistered type id -- where the
reference raises ``IndexError`` / ``KeyError``) is polled once per epoch by ``run_experiment`` /
``generate_memories`` and at the end of ``collect``; a bare ``take_turn`` loop should call
``env.raise_on_status()`` itself now and then (it synchronises, so not every turn).
"""
from __future__ import annotations

from abc import abstractmethod
from typing import List, Optional

import os
from pathlib import Path

import numpy as np
import torch

from sorrel_amd.agents.agent import Agent
from sorrel_amd.entities.entity import Entity
from sorrel_amd.agents.rules import CleanupRule, TagRule
from sorrel_amd.entities.rules import AgeRule, BecomeIfRule
from sorrel_amd.spec import NO_BORDER, RULE_BECOME_IF, RULE_NONE, RULE_SPAWN, WorldSpec, action_deltas

try:  # omegaconf is optional (absent in the build image)
    from omegaconf import DictConfig, OmegaConf  # type: ignore
except Exception:  # pragma: no cover
    DictConfig = None
    OmegaConf = None


class AttrDict(dict):
    """dict with attribute access: stands in for ``omegaconf.DictConfig`` when it is absent."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _normalise_config(config):
    if DictConfig is not None and isinstance(config, DictConfig):
        return config
    if isinstance(config, dict):
        return OmegaConf.create(config) if OmegaConf is not None else AttrDict(config)
    if isinstance(config, (list, tuple)):       # dotlist
        root: dict = {}
        for item in config:
            k, v = item.split("=", 1)
            cur = root
            parts = k.split(".")
            for p in parts[:-1]:
                cur = cur.setdefault(p, {})
            cur[parts[-1]] = v
        return OmegaConf.create(root) if OmegaConf is not None else AttrDict(root)
    return config


class _FillEntity(Entity):
    """Stand-in type for ``fill_entity_kind`` when no placed entity has that kind."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind


class Environment:
    world = None
    config = None
    agents: List[Agent]
    stop_if_done: bool

    _obs_dtype = torch.float32

    @property
    def obs_dtype(self):
        """Element type of the step's observation tensor: float32 (what the reference's replay stores, the default)
        or uint8 (compact one-hot counts).  Assigning rebuilds the engine handle on the next use; world state is kept."""
        return self._obs_dtype

    @obs_dtype.setter
    def obs_dtype(self, dtype):
        if dtype not in (torch.float32, torch.uint8):
            raise ValueError("obs_dtype must be torch.float32 or torch.uint8")
        if dtype != self._obs_dtype:
            self._obs_dtype = dtype
            self._engine_version = -1
            self._fresh_obs = None

    def __init__(self, world, config, stop_if_done: bool = False) -> None:
        self.config = _normalise_config(config)
        self.world = world
        world._environment = self
        self.turn = 0
        self.epoch = 0
        self._fresh_obs = None       # (slot, world mutation count): eng.obs[:, slot] was rendered by the last launch
        self._turn_windows = None    # [world mutation count, rows, first agent whose window is still current, replay slots]: this turn's windows
        self._replay_slots = None
        self._tail_rows = None
        self._capture_rows = None    # recorded turns: (pointer array, stride, per-agent [E, C*V*V] tensors) the policies read their windows from
        self._captured = None        # CapturedTurn: a whole policy turn recorded as one graph (capture_turn)
        self._turn_capture = False   # the turn protocol with device-side counters is in charge of this turn (recording or warming up)
        self._value_agents = set()   # slots whose get_action returns action values: the act launch takes the argmax / explores (SGW_ACT_QF32)
        self._eps_pushed = {}        # slot -> (engine uid, epsilon) last sent to the device's turn state
        self._turn_state_at = {}     # engine uid -> (epoch, turn) the device's turn state was last set for by the eager loop
        self._engine = None
        self._engine_version = -1
        self._aux_engines = {}
        self._group_engines = {}     # (ospec key, action names) -> the engine handle compiled from them
        self._agent_engine = []      # per agent slot: its handle
        self._mixed = False          # agents differ in their specs (or observe the whole map): they step one after another on their own handles
        self._mixed_obs = []         # per agent slot: the tensor its last window was rendered into (mixed mode)
        self._device_populated = False
        self.world.create_world()
        self.stop_if_done = stop_if_done
        self.setup_agents()
        self._attach_agents()
        self.populate_environment()

    @abstractmethod
    def setup_agents(self) -> None:
        """Create the agents and assign them to ``self.agents``."""

    @abstractmethod
    def populate_environment(self) -> None:
        """Populate the (already default-filled) world: either host-side ``world.add`` calls
        (applied to every env) or ``world.set_layout(...)`` + ``self.spawn_agents()`` for the
        on-device reset kernel."""

    # ------------------------------------------------------------------ batch plumbing
    @property
    def num_envs(self) -> int:
        return self.world.num_envs

    def _attach_agents(self):
        w = self.world
        if len(self.agents) == 0:
            raise ValueError("setup_agents() must create at least one agent")
        w.agent_slots = list(self.agents)
        w.agent_pos = torch.zeros((w.num_envs, len(self.agents), 2), dtype=torch.uint8, device=w.device)
        for slot, agent in enumerate(self.agents):
            agent.slot, agent._world = slot, w

    def spawn_agents(self) -> None:
        """Place the agents on distinct random interior cells of their layer in every env
        (``examples/treasurehunt/env.py:138-147``) -- runs the reset kernel (K3) together
        with the layout declared by ``world.set_layout``."""
        if self.world.layout is None:
            raise ValueError("spawn_agents() needs world.set_layout(...) first")
        if self.world.agent_layer is None:
            self.world.agent_layer = self.world.layers - 1
        self._device_populated = True
        eng = self._ensure_engine()
        eng.reset(epoch=self.epoch)

    def compile_spec(self, ospec=None, aspec=None) -> WorldSpec:
        """Entities, agents, one observation spec and one action spec -> the engine's tables.  Default: those of agent 0.  Every
        reference agent carries its OWN specs (``sorrel/agents/agent.py:38-48``); agents that differ are served by one engine handle per
        distinct (observation spec, action spec) over the same world tensors (``_ensure_engine``), each compiled here.  ``ospec`` alone
        compiles another observation spec than any agent's (on-demand observations)."""
        w, agents = self.world, self.agents
        if aspec is None:
            aspec = agents[0].action_spec
        if ospec is None:
            ospec = agents[0].observation_spec
        rules = {type(getattr(a, "interaction_rule", None)) for a in agents}
        if len(rules) > 1:
            raise ValueError("the agents of one batched Environment must share their interaction rule (plain movers, Tag or Cleanup)")
        rule = getattr(agents[0], "interaction_rule", None)
        extra = dict(agent_rule=0)
        if rule is None:
            agent_types = [w.registry.register(a) for a in agents]
        elif isinstance(rule, TagRule):
            it_t = w.registry.register(agents[0].as_kind(rule.it_kind))
            notit_t = w.registry.register(agents[0].as_kind(rule.notit_kind))
            agent_types = [notit_t] * len(agents)
            extra = dict(agent_rule=1, tag_it_type=it_t, tag_notit_type=notit_t, tag_reward=float(rule.reward_per_turn))
        elif isinstance(rule, CleanupRule):
            agent_types = [w.registry.register(a) for a in agents]
            clean, zap = rule.beams()
            kinds = {rule.clean_action: 1, rule.zap_action: 2}          # SGW_ACTION_CLEAN / SGW_ACTION_ZAP
            extra = dict(agent_rule=2, action_kind=[kinds.get(n, 0) for n in aspec.names], beam_radius=rule.beam_radius,
                         clean_beam_type=w.registry.register(clean), zap_beam_type=w.registry.register(zap),
                         reward_total_factor=2 if rule.count_total_twice else 1)
        else:
            raise ValueError(f"unsupported agent interaction rule {rule!r}")
        # resolve transition rules (may register the types they produce); iterate to a fixed point
        spawn, become = {}, {}
        n_seen = -1
        while n_seen != len(w.registry):
            n_seen = len(w.registry)
            for t, proto in enumerate(list(w.registry.prototypes)):
                if isinstance(proto, Agent) or not proto.has_transitions or t in spawn or t in become:
                    continue
                trule = proto.transition_rule
                if trule is None:
                    raise ValueError(
                        f"{type(proto).__name__} has has_transitions=True but no declarative transition_rule; "
                        "arbitrary Python transition() bodies cannot run on the device")
                if isinstance(trule, BecomeIfRule):
                    ent, layer, kinds_ = trule.resolve(w)
                    become[t] = (w.registry.register(ent), layer, kinds_)
                elif isinstance(trule, AgeRule):
                    for ent, r in trule.chain(proto, w):
                        nxt, layer, kinds_ = r.resolve(w)
                        become[w.registry.register(ent)] = (w.registry.register(nxt), layer, kinds_)
                else:
                    spawn[t] = w.spawn_rule_of(proto)
        fill_kind = ospec.fill_entity_kind
        fill_type = next((t for t, p in enumerate(w.registry.prototypes) if p.kind == fill_kind), None)
        if fill_type is None:
            fill_type = w.registry.register(_FillEntity(fill_kind))
        protos = w.registry.prototypes
        T, C = len(protos), ospec.num_channels
        app = np.zeros((T, C), dtype=np.float64)
        for t, p in enumerate(protos):
            if p.kind not in ospec.entity_map:
                raise KeyError(p.kind)     # the reference raises KeyError in visual_field for an unmapped kind
            app[t] = np.asarray(ospec.entity_map[p.kind], dtype=np.float64)
        dy, dx = action_deltas(aspec.names)
        if w.agent_layer is None:
            w.agent_layer = w.layers - 1
        if extra["agent_rule"] == 2:
            extra["beam_block_mask"] = sum(1 << t for t, p in enumerate(protos) if type(p).__name__ in rule.blocked)
            if w.agent_layer + 1 >= w.layers:
                raise ValueError("CleanupRule needs a beam layer above the agent layer")

        def kind_mask(kinds_):
            return sum(1 << t for t, p in enumerate(protos) if p.kind in kinds_)

        lay = w.layout or dict(fill=[w.default_type] * w.layers, border=[NO_BORDER] * w.layers, dense_prob=0.0, dense=[])
        return WorldSpec(
            height=w.height, width=w.width, layers=w.layers, num_agents=len(agents),
            vision_radius=0 if ospec.full_view else ospec.vision_radius, num_channels=C, agent_layer=w.agent_layer,
            default_type=w.default_type, fill_type=fill_type, action_dy=dy, action_dx=dx,
            agent_type=agent_types,
            type_value=[p.value for p in protos], type_passable=[1 if p.passable else 0 for p in protos],
            type_rule=[RULE_SPAWN if t in spawn else RULE_BECOME_IF if t in become else RULE_NONE for t in range(T)],
            rule_layer=[become[t][1] if t in become else 0 for t in range(T)],
            rule_mask=[kind_mask(become[t][2]) if t in become else 0 for t in range(T)],
            rule_become=[become[t][0] if t in become else 0 for t in range(T)],
            spawn_prob=[spawn[t][0] if t in spawn else 0.0 for t in range(T)],
            spawn_choices=[spawn[t][1] if t in spawn else [] for t in range(T)],
            appearance=app, seed=w.seed, layer_fill_type=lay["fill"], layer_border_type=lay["border"],
            dense_prob=lay["dense_prob"], dense_choices=lay["dense"],
            type_names=[type(p).__name__ for p in protos], obs_post=int(getattr(ospec, "obs_post", 0)), **extra,
        )

    def _agent_key(self, agent):
        """What an agent's engine handle is compiled from: its observation spec (table, radius, fill kind, whole map or window) and
        its action list."""
        return (self._ospec_key(agent.observation_spec), tuple(agent.action_spec.names))

    def _ensure_engine(self):
        """The engine handle of agent 0's specs (and of every agent that shares them) -- plus, when agents differ, one more handle per
        distinct (observation spec, action spec) over the SAME grid / position / action / reward tensors (``_group_engines``)."""
        from sorrel_amd.engine import GridEngine

        w = self.world
        if self._engine is not None and self._engine_version == w.registry.version:
            return self._engine
        keys = [self._agent_key(a) for a in self.agents]
        distinct = list(dict.fromkeys(keys))
        owner = {k: self.agents[keys.index(k)] for k in distinct}
        while True:                       # compiling a spec may register its fill kind as a new type: every table must see all of them
            n_types = len(w.registry)
            specs = {k: self.compile_spec(owner[k].observation_spec, owner[k].action_spec) for k in distinct}
            if len(w.registry) == n_types:
                break
        for eng in self._all_engines():
            eng.close()
        first = getattr(w, "first_env_id", 0)
        tensors = dict(grid=w.grid, agent_pos=w.agent_pos, total_reward=w.total_reward)
        if getattr(w, "agent_state", None) is not None:
            tensors["agent_state"] = w.agent_state          # survives engine rebuilds (and resets)
        if getattr(w, "agent_dir", None) is not None:
            tensors["agent_dir"] = w.agent_dir
        # agents that differ, or whose own spec is the whole map, step one after another on their own handles (take_turn); no handle
        # then needs the [E][A][C][V][V] tensor of a fused turn
        self._mixed = len(distinct) > 1 or any(a.observation_spec.full_view for a in self.agents)
        self._engine = GridEngine(specs[distinct[0]], w.num_envs, device=w.device, first_env_id=first, tensors=tensors,
                                  obs_dtype=self.obs_dtype, allocate_obs=not self._mixed)
        self._fresh_obs = None
        w.agent_state = self._engine.agent_state
        w.agent_dir = self._engine.agent_dir
        self._engine_version = w.registry.version
        self._group_engines = {distinct[0]: self._engine}
        shared = dict(tensors, actions=self._engine.actions, rewards=self._engine.rewards)
        if self._engine.agent_state is not None:
            shared["agent_state"] = self._engine.agent_state
        if self._engine.agent_dir is not None:
            shared["agent_dir"] = self._engine.agent_dir
        for k in distinct[1:]:
            self._group_engines[k] = GridEngine(specs[k], w.num_envs, device=w.device, first_env_id=first, tensors=shared,
                                                obs_dtype=self.obs_dtype, allocate_obs=False)
        self._agent_engine = [self._group_engines[k] for k in keys]
        self._mixed_obs = [None] * len(self.agents)
        self._aux_engines = {}
        self._bind_row_tail()
        self._validate_border()
        return self._engine

    def _all_engines(self):
        """Every handle this environment has built: the step engine, the handles of agents with other specs, on-demand ones."""
        seen, out = set(), []
        for eng in [self._engine] + list(getattr(self, "_group_engines", {}).values()) + list(self._aux_engines.values()):
            if eng is not None and id(eng) not in seen:
                seen.add(id(eng))
                out.append(eng)
        return out

    def _bind_row_tail(self):
        """What the agents' ``pov`` appends behind the flattened window (``Agent.row_tail``: Tag's "it" flag, Cleanup's positional
        code) is written by the engine behind the window in every row (``sgw_bind_row_tail``) -- when every agent declares the same
        tail and the engine renders float32 windows into rows; otherwise the agents concatenate on the host as before."""
        from sorrel_amd import _native as N

        eng = self._engine
        self._tail_rows = None
        self._turn_windows = None
        if eng is None:          # (nothing built yet: _ensure_engine binds)
            return
        if eng.row_tail:
            eng.bind_row_tail(N.TAIL_NONE)
        tails = [a.row_tail(self.world) for a in self.agents]
        if self._mixed or any(t is None for t in tails) or eng.obs_dtype != torch.float32 or not (eng.capabilities() & N.CAP_OBSERVE_ROWS) or not self.row_tails_in_kernel:
            return
        kind, table = tails[0]
        for k, t in tails[1:]:
            if k != kind or (t is None) != (table is None) or (t is not None and (t.shape != table.shape or not torch.equal(t, table))):
                return
        eng.bind_row_tail(kind, table)

    #: recorded turns: "rows" (default where the engine can: per-agent window rows, replay rows written alongside) or "tensor" (the
    #: observation tensor + a copy into the replay rows at the end of the turn) -- A/B and tests
    capture_layout = "rows"

    #: ``run_experiment``: record the policy turn once (``capture_turn``) and replay it for every later turn of every epoch; the eager
    #: loop stays in charge where a turn cannot be recorded (``capture_error`` says why) or ``stop_if_done`` is set.  Also read from
    #: ``config.experiment.capture_turns``.  Off by default: a recorded turn freezes the Python inside the agent loop (a policy
    #: that branches on host state must stay eager).
    capture_turns = False

    #: Tag / Cleanup agents: let the engine write what ``pov`` appends behind the window (False = ``torch.cat`` on the host: A/B and tests)
    row_tails_in_kernel = True

    def _pov_row(self, slot: int):
        """``[E, C*V*V + tail]``: this turn's finished row of agent ``slot`` -- window and tail, both written by the engine -- or None
        when the turn's windows were not rendered into tailed rows (then ``pov`` concatenates on the host)."""
        tw = self._turn_windows
        eng = self._engine
        if tw is None or eng is None or eng.row_tail == 0 or tw[0] != self.world.mutations or slot < tw[2] or tw[1][2] is None:
            return None
        return tw[1][2][slot].view(eng.num_envs, -1)

    def _validate_border(self):
        """Engine precondition: the reference has no bounds check in ``move`` --
        the agent layer's border must be impassable."""
        w = self.world
        if self._device_populated:
            return
        passable = torch.tensor([1 if p.passable else 0 for p in w.registry.prototypes], dtype=torch.uint8, device=w.device)
        g = w.grid[:, w.agent_layer].long()
        border = torch.cat([g[:, 0, :], g[:, -1, :], g[:, :, 0], g[:, :, -1]], dim=1)
        if bool(passable[border].any()):
            raise ValueError("the border of the agent layer must be impassable in every env "
                             "(Gridworld.move has no bounds check)")

    # ------------------------------------------------------------------ reference API
    def reset(self) -> None:
        """``turn = 0``, fresh world, re-populate, reset agents (``environment.py:72-79``)."""
        self.turn = 0
        self.epoch += 1
        self._fresh_obs = None
        self._turn_windows = None
        self.world.is_done = False
        self.world.create_world()
        self.populate_environment()
        for agent in self.agents:
            agent.reset()
        if self._captured is not None:
            self._captured.resync()

    def take_turn(self, actions: Optional[torch.Tensor] = None) -> None:
        """One full step of every env: entity transitions, then each agent in list order
        (``environment.py:81-93``)."""
        eng = self._ensure_engine()
        if actions is None and self._captured is not None:
            if self._captured.valid(eng):
                self._captured.replay()
                return
            self._captured = None            # the engine was rebuilt (new entity types, another obs dtype): back to the eager loop
        if self._mixed:
            return self._take_turn_mixed(eng, actions)
        self.turn += 1
        eng.epoch, eng.turn = self.epoch, self.turn
        self._fresh_obs = None
        if actions is not None:
            eng.step(actions, turn=self.turn)
        elif all(getattr(a.model, "device_random", False) for a in self.agents):
            eng.step(random_actions=True, turn=self.turn)
        elif self.speculate_turns and self._speculation_groups(eng) is not None:
            self._take_turn_speculative(eng, self._speculation_groups(eng))
        elif self.fast_policy_loop and self._fast_plan(eng) is not None:
            self._fast_plan(eng).run_turn()
        elif self._begin_policy_turn(eng):
            for agent in self.agents:
                agent.transition(self.world)
            self._turn_windows = None
        else:
            # entity sweep; the same launch renders agent 0's observation (nothing intervenes before its pov)
            slot = self._replay_slot(0, None)
            eng.step(sweep=True, agent_begin=0, agent_end=0, obs_next=True, obs_next_out=slot, turn=self.turn)
            self._fresh_obs = (0, self.world.mutations, slot)
            for agent in self.agents:
                agent.transition(self.world)

    def turn_plan(self) -> dict:
        """Which of the loops above will play the next ``take_turn()`` (without actions), and why the faster ones do not apply: a diagnostic --
        ``{"loop": "recorded" | "per-agent handles" | "fused" | "speculative" | "fast" | "generic", ...}``."""
        from sorrel_amd import _native as N

        eng = self._ensure_engine()
        if self._captured is not None and self._captured.valid(eng):
            return {"loop": "recorded", "turns_replayed": self._captured.turns_replayed}
        if self._mixed:
            return {"loop": "per-agent handles", "why": "the agents hold different observation / action specs", "handles": len(self._all_engines())}
        if all(getattr(a.model, "device_random", False) for a in self.agents):
            return {"loop": "fused", "why": "every agent's actions are drawn on the device: one launch per turn"}
        out = {}
        if self.speculate_turns:
            if self._speculation_groups(eng) is not None:
                return {"loop": "speculative", "models": len(self._speculation_groups(eng))}
            out["speculative"] = "not possible for these agents, or the cost model keeps the sequential loop (speculate_turns = 'always' overrides it)"
        plan = self._fast_plan(eng) if self.fast_policy_loop else None
        caps = eng.capabilities()
        if plan is not None:
            out.update(loop="fast", one_launch_windows=bool(plan.fused and self.fuse_sweep_and_rows), launches=(1 if plan.fused and self.fuse_sweep_and_rows else 2) + len(self.agents))
            return out
        out["fast"] = ("switched off" if not self.fast_policy_loop else
                       "needs agents whose class sets speculative_ok (no hook overridden), one-frame Buffer memories of exactly one window (+ tail) per row, float32 windows")
        patched = self.patch_windows and bool(caps & N.CAP_ACT) and eng.obs is not None
        out.update(loop="generic", protocol="windows once + sgw_act per agent" if patched else "a window per agent launch (1 + A)")
        return out

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

    # ------------------------------------------------------------------ agents that differ (sorrel/agents/agent.py:38-48)
    def _take_turn_mixed(self, eng, actions) -> None:
        """``take_turn`` for agents that hold different observation / action specs (or observe the whole map, ``full_view``): the
        entity sweep once, then agent after agent on the handle compiled from ITS specs -- its window (its radius, table and fill
        kind; or the whole layer-summed map) from the grid as the agents before it left it, then its act through its own action list
        (``Agent.transition``, ``agent.py:155-173``).  1 + 2 A launches; the fused one-launch turn needs agents that share their specs.
        ``actions`` ``[E, A]``: indices into each agent's OWN action list."""
        self.turn += 1
        for g in self._all_engines():
            g.epoch, g.turn = self.epoch, self.turn
        self._fresh_obs = None
        self._turn_windows = None
        if actions is not None:
            eng.actions.copy_(actions.to(device=eng.device, dtype=torch.uint8).reshape(eng.actions.shape))
        eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=self.turn)     # the sweep alone
        for a, agent in enumerate(self.agents):
            if actions is not None or getattr(agent.model, "device_random", False):
                g = self._agent_engine[a]
                self._mixed_window(a)
                g.step(g.actions, random_actions=actions is None, sweep=False, write_obs=False, agent_begin=a, agent_end=a + 1,
                       turn=self.turn)
            else:
                agent.transition(self.world)

    def _mixed_window(self, a: int) -> torch.Tensor:
        """Agent ``a``'s observation through its own spec, from the grid as it stands: ``[E, C, V, V]`` (or ``[E, C, H, W]`` with
        ``full_view``), rendered by its handle into the replay row its ``add_memory`` is about to fill where that applies, else into a
        tensor of its own."""
        g = self._agent_engine[a]
        ospec = self.agents[a].observation_spec
        if ospec.full_view:
            out = self._mixed_obs[a]
            if out is None or out.dtype != g.obs_dtype:
                out = None
            self._mixed_obs[a] = g.observe_full(out)
            return self._mixed_obs[a]
        shape = (g.num_envs,) + tuple(g.spec.obs_shape[1:])
        dest = self._replay_slot(a, None, g)
        if dest is None:
            dest = self._mixed_obs[a]
            if dest is None or dest.dtype != g.obs_dtype or tuple(dest.shape) != shape:
                dest = torch.zeros(shape, dtype=g.obs_dtype, device=g.device)
        self._mixed_obs[a] = dest
        g.step(g.actions, sweep=False, agent_begin=a, agent_end=a, obs_next=True, obs_next_out=dest, turn=self.turn)
        return dest.view(shape)

    def obs_of(self, agent) -> Optional[torch.Tensor]:
        """The window agent (or slot) last observed -- ``obs[:, slot]`` when the agents share their specs."""
        a = agent.slot if isinstance(agent, Agent) else int(agent)
        if not self._mixed:
            eng = self._ensure_engine()
            seen = getattr(self, "_spec_seen", None)
            if seen is not None and seen[:2] == (self.epoch, self.turn):      # the last turn was a speculative one
                nw = int(np.prod(eng.spec.obs_shape[1:]))                      # (rows of the generic form carry the pov's tail behind the window)
                return seen[2][a][:, :nw].reshape((eng.num_envs,) + tuple(eng.spec.obs_shape[1:]))
            return eng.obs[:, a]
        t = self._mixed_obs[a]
        g = self._agent_engine[a]
        if t is None or self.agents[a].observation_spec.full_view:
            return t
        return t.view((g.num_envs,) + tuple(g.spec.obs_shape[1:]))

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

    def rollout(self, turns: int) -> None:
        """``turns`` fused ``take_turn``s with ONE engine call (``sgw_rollout``: one launch with every env's grid resident
        in LDS where the kernel supports it) -- for agents whose actions are drawn on device (``RandomModel``).  The
        step outputs (``obs`` / ``rewards`` / ``actions``) hold the last turn afterwards, as after ``turns`` calls of
        ``take_turn``."""
        if not all(getattr(a.model, "device_random", False) for a in self.agents):
            raise ValueError("rollout() needs device-random models; policy-driven agents step turn by turn (take_turn)")
        if turns <= 0:
            return
        eng = self._ensure_engine()
        if self._mixed:                          # agents that differ: turn by turn, agent by agent
            for _ in range(int(turns)):
                self.take_turn()
            return
        eng.epoch, eng.turn = self.epoch, self.turn
        self._fresh_obs = None
        eng.rollout(int(turns))
        self.turn += int(turns)

    def raise_on_status(self) -> None:
        """Synchronising poll of the device status word: raises what the reference would have raised
        (``IndexError`` for a move off an un-walled border, ``KeyError`` for an action index outside the
        ``ActionSpec`` or an unregistered entity type)."""
        self._ensure_engine()
        err = None
        for g in self._all_engines():            # every handle has a status word of its own
            try:
                g.raise_on_status()
            except (IndexError, KeyError) as exc:
                err = err or exc
        if err is not None:
            raise err

    def collect(self, turns: int, buffer, actions=None) -> None:
        """``turns`` fused ``take_turn``s whose observations the step kernel writes straight into ``buffer``
        (a ``sorrel_amd.buffers.TurnBuffer``): no per-step copy of the observation tensor.  ``actions``: optional
        ``[turns, E, A]`` uint8 tensor of policy actions; default = on-device random actions (``RandomModel``)."""
        eng = self._ensure_engine()
        if self._mixed:
            raise ValueError("collect() writes one [E, A, C, V, V] tensor per turn: the agents must share their observation and action specs")
        want = (buffer.capacity, eng.num_envs) + tuple(eng.spec.obs_shape)
        if tuple(buffer.obs.shape) != want or buffer.obs.dtype != eng.obs_dtype or buffer.obs.device != eng.device:
            raise ValueError(f"buffer.obs must be {eng.obs_dtype} {want} on {eng.device} (the step kernel writes its "
                             f"observations straight into a slot); got {buffer.obs.dtype} {tuple(buffer.obs.shape)} on "
                             f"{buffer.obs.device}.  Set Environment.obs_dtype to match a uint8 buffer.")
        self._fresh_obs = None
        if buffer.positions is None and eng.max_turns == 0:
            # whole runs of consecutive ring slots in ONE call (sgw_rollout: where the kernel supports it the turns run
            # inside one launch with every env's grid resident in LDS): observations, actions and rewards of turn t go
            # straight to slot idx + t
            done = 0
            while done < turns:
                i = buffer.slot()
                n = min(turns - done, buffer.capacity - i)
                eng.epoch, eng.turn = self.epoch, self.turn
                eng.rollout(n, actions=None if actions is None else actions[done:done + n].to(device=eng.device, dtype=torch.uint8).contiguous(),
                            obs_out=buffer.obs[i:i + n], actions_out=None if actions is not None else buffer.actions[i:i + n],
                            rewards_out=buffer.rewards[i:i + n])
                if actions is not None:
                    buffer.actions[i:i + n].copy_(actions[done:done + n])
                buffer.advance(n)
                self.turn += n
                done += n
        else:
            for t in range(turns):
                self.turn += 1
                eng.epoch, eng.turn = self.epoch, self.turn
                out = buffer.obs[buffer.slot()]
                if actions is None:
                    eng.step(random_actions=True, turn=self.turn, obs_out=out)
                else:
                    eng.step(actions[t], turn=self.turn, obs_out=out)
                buffer.commit(eng.actions, eng.rewards, eng.agent_pos)
        eng.raise_on_status()

    # ------------------------------------------------------------------ kernels behind the agent hooks
    @staticmethod
    def _ospec_key(ospec):
        """What an observation spec compiles to, as a hashable key.  Computed once per (spec object, entity-map object,
        scalar settings) and kept on the spec: every agent's ``pov`` asks for it every turn, and serialising the
        appearance vectors each time cost ~7 us per agent phase.  Replace ``entity_map`` (or change radius / fill kind) to
        have a spec recompiled; after editing an appearance vector IN PLACE call ``ObservationSpec.invalidate()``."""
        sig = (id(ospec.entity_map), len(ospec.entity_map), ospec.vision_radius, ospec.fill_entity_kind, getattr(ospec, "obs_post", 0),
               bool(ospec.full_view))
        cached = ospec.__dict__.get("_sgw_key")
        if cached is not None and cached[0] == sig:
            return cached[1]
        key = (type(ospec).__name__, 0 if ospec.full_view else int(ospec.vision_radius), ospec.fill_entity_kind,
               int(getattr(ospec, "obs_post", 0)), bool(ospec.full_view),
               tuple((k, np.asarray(v, dtype=np.float64).tobytes()) for k, v in ospec.entity_map.items()))
        try:
            ospec.__dict__["_sgw_key"] = (sig, key, ospec.entity_map)   # (the map is kept alive so that its id cannot be reused)
        except (AttributeError, TypeError):      # a spec class with __slots__: just recompute
            pass
        return key

    def _engine_for(self, ospec):
        """The engine whose tables were compiled from ``ospec``: the step engine when it is (equal to) the
        agents' own spec, otherwise a second handle over the same grid / position tensors, built on first
        use (``observe`` with another entity map, radius or fill kind -- ``visual_field`` with arguments)."""
        eng = self._ensure_engine()
        if ospec is None or self._ospec_key(ospec) == self._ospec_key(self.agents[0].observation_spec):
            return eng
        for (okey, _names), g in self._group_engines.items():      # the handle of an agent that holds this very spec
            if okey == self._ospec_key(ospec):
                return g
        from sorrel_amd.engine import GridEngine

        key = (self._ospec_key(ospec), self.world.registry.version)
        aux = self._aux_engines.get(key)
        if aux is None:
            w = self.world
            spec = self.compile_spec(ospec)
            if len(w.registry) != eng.spec.num_types:     # the new spec registered a fill type: rebuild the step engine too
                eng = self._ensure_engine()
                key = (self._ospec_key(ospec), w.registry.version)
            aux = GridEngine(spec, w.num_envs, device=w.device, first_env_id=getattr(w, "first_env_id", 0),
                             tensors=dict(grid=w.grid, agent_pos=w.agent_pos), allocate_obs=False)
            self._aux_engines = {k: v for k, v in self._aux_engines.items() if k[1] == w.registry.version}
            self._aux_engines[key] = aux
        return aux

    def _observe(self, who, ospec=None):
        """[E, C, V, V] float32 of an agent slot (Agent / int), or from a (y, x, z) cell, as seen through
        ``ospec`` (default: the agents' own observation spec)."""
        if isinstance(who, Agent):
            who = who.slot
        if self._ensure_engine() is not None and self._mixed and isinstance(who, int) and \
                (ospec is None or ospec is self.agents[who].observation_spec):
            return self._mixed_window(who)          # the agent's own spec on the agent's own handle
        eng = self._engine_for(ospec)
        own = eng is self._engine and not self._mixed
        if isinstance(who, int):
            tw = self._turn_windows
            if own and tw is not None and tw[0] == self.world.mutations and who >= tw[2]:
                # this turn's window of an agent that has not acted yet: rendered after the sweep, kept current by the act
                # launches of the agents before it
                dests = tw[1][2]
                return eng.obs[:, who] if dests is None else dests[who].view((eng.num_envs,) + tuple(eng.spec.obs_shape[1:]))
            if own and self._fresh_obs is not None and self._fresh_obs[:2] == (who, self.world.mutations):
                # rendered by the launch that moved the previous agent (SGW_STEP_OBS_NEXT) -- into the observation tensor,
                # or straight into the row of this agent's replay buffer that add_memory is about to fill
                slot = self._fresh_obs[2]
                return eng.obs[:, who] if slot is None else slot.view((eng.num_envs,) + tuple(eng.spec.obs_shape[1:]))
            out = eng.obs if (own and eng.obs is not None) else eng.scratch_obs()
            eng.observe(who, who + 1, out=out)
            return out[:, who]
        y, x, _z = (int(v) for v in who)          # the window is layer-summed: only (y, x) matters
        pos = torch.zeros_like(eng.agent_pos)
        pos[:, 0, 0], pos[:, 0, 1] = y, x
        eng.observe(0, 1, pos=pos, out=eng.scratch_obs())
        return eng.scratch_obs()[:, 0]

    def _full_view(self, ospec, who=None):
        """Whole-map appearance summed over layers, ``[E, C, H, W]`` float32 (``visual_field.py:41-55``): the engine's
        ``sgw_observe_full`` on the handle compiled from ``ospec``'s appearance table.  The same for every agent; a
        ``full_view`` spec is still rejected as the agents' OWN spec of a fused step (it would mean A whole maps per env
        and turn), it is an on-demand observation."""
        if isinstance(who, Agent) and who.slot is not None and self._ensure_engine() is not None and self._mixed \
                and ospec is self.agents[who.slot].observation_spec:
            return self._mixed_window(who.slot)       # the agent's own whole-map spec inside a turn: its handle, its tensor
        return self._engine_for(ospec).observe_full()

    #: policy-driven turns render each agent's window straight into its replay row where that is possible (see below);
    #: False = always through the observation tensor + a copy in ``Buffer.add`` (A/B and test switch)
    write_obs_into_replay = True

    def _replay_slot(self, a: int, acting: Optional[int], eng=None):
        """The row of agent ``a``'s replay buffer that its next ``add_memory`` will fill, if the step kernel can write
        the agent's window straight into it (``SGW_STEP_OBS_NEXT_PACKED``): a ``sorrel_amd.buffers.Buffer`` of the engine's
        dtype and device whose rows hold exactly one window (a ``pov`` that appends to the window, like Cleanup's
        positional code, does not qualify), not shared with the agent that acts in between (its ``add_memory`` would
        land on the same row first).  ``Buffer.add`` then finds the state already in place and copies nothing -- at
        65 536 envs of config 3 that copy is 77 MB per agent and turn."""
        from sorrel_amd.buffers import Buffer

        eng = self._engine if eng is None else eng
        mem = getattr(self.agents[a].model, "memory", None)
        if not self.write_obs_into_replay or not isinstance(mem, Buffer) or eng is None or (eng.obs is None and not self._mixed):
            return None
        if type(self.agents[a]).transition is not Agent.transition or type(self.agents[a]).add_memory is not Agent.add_memory:
            return None        # (the row is pre-written: only safe if this agent's pov is always followed by its add_memory)
        if acting is not None and mem is getattr(self.agents[acting].model, "memory", None):
            return None
        row = mem.states[mem.idx]
        per_env = 1
        for d in eng.spec.obs_shape[1:]:
            per_env *= int(d)
        if row.dtype != eng.obs_dtype or row.device != eng.device or not row.is_contiguous() or row.dim() < 2 \
                or row.shape[0] != eng.num_envs or row.numel() != eng.num_envs * per_env:
            return None
        return row

    def _push_epsilon(self, eng, slots=None) -> None:
        """The exploration rates of the agents that act through action values, to the device's turn state when they change."""
        for a in (self._value_agents if slots is None else slots):
            eps = min(1.0, max(0.0, self.agents[a].epsilon))
            if self._eps_pushed.get(a) != (eng.uid, eps):
                eng.turn_epsilon(eps, a)
                self._eps_pushed[a] = (eng.uid, eps)

    def _act(self, agent: Agent, action) -> torch.Tensor:
        eng = self._ensure_engine()
        a = agent.slot
        if self._mixed:
            eng = self._agent_engine[a]          # the handle compiled from this agent's own action list
        if not torch.is_tensor(action):
            action = torch.full((self.num_envs,), int(action), dtype=torch.uint8, device=self.world.device)
        values = action.dim() == 2
        if values:
            # the policy's action VALUES [E, n_actions]: the act launch takes the argmax itself (one launch less per agent) and, with
            # probability agent.epsilon, the engine's own uniform draw for (env, turn, agent) instead (iqn.py:294-309)
            if tuple(action.shape) != (self.num_envs, eng.spec.num_actions) or not action.is_floating_point():
                raise ValueError(f"action values must be floating point [{self.num_envs}, {eng.spec.num_actions}]; got {action.dtype} {tuple(action.shape)}")
            action = action.to(device=eng.device, dtype=torch.float32).contiguous()
            if a not in self._value_agents:
                if self._captured is not None and self._captured.graph is not None:
                    raise RuntimeError("an agent switched to action values after its turn was recorded: capture_turn() again")
                self._value_agents.add(a)
            if not self._turn_capture and self._turn_state_at.get(eng.uid) != (self.epoch, self.turn):
                eng.turn_set(self.epoch, self.turn - 1)               # the exploration draws are keyed by the turn in flight
                self._turn_state_at[eng.uid] = (self.epoch, self.turn)
            if not (self._turn_capture and torch.cuda.is_current_stream_capturing()):
                self._push_epsilon(eng, (a,))                         # (a recorded turn gets its epsilons before each replay)
        if self._mixed:                          # windows are rendered per agent at its pov: nothing to keep current
            direct = values or (action.device == eng.device and action.dtype in eng._ACTION_KINDS and action.dim() == 1
                                and action.shape[0] == eng.num_envs and action.is_contiguous())
            if not direct:
                eng.actions[:, a].copy_(action)
            return eng.act(a, None, action=action if direct else None)
        tw = self._turn_windows
        if self._turn_capture:                   # the turn protocol with device-side counters (capture_turn): rows by the device's count
            if tw is None or tw[0] != self.world.mutations or a < tw[2]:
                raise RuntimeError("a captured policy turn cannot be recorded while host code edits the world between pov and act")
            tw[2] = a + 1
            direct = values or (action.device == eng.device and action.dtype in eng._ACTION_KINDS and action.dim() == 1
                                and action.shape[0] == eng.num_envs and action.is_contiguous())
            if not direct:
                eng.actions[:, a].copy_(action)
            if self._capture_rows is not None:
                return eng.turn_act_rows(a, tw[1], action if direct else None)
            return eng.turn_act(a, action if direct else None)
        if tw is not None:
            if tw[0] == self.world.mutations and a >= tw[2]:
                tw[2] = a + 1                    # the windows of the agents after a stay current: sgw_act repairs them
                # the policy's output goes to the kernel as it is (no narrowing copy); rewards and actions are also written
                # where the agent's add_memory would copy them to
                direct = values or (action.device == eng.device and action.dtype in eng._ACTION_KINDS and action.dim() == 1
                                    and action.shape[0] == eng.num_envs and action.is_contiguous())
                if not direct:
                    eng.actions[:, a].copy_(action)
                rr = ar = None
                if tw[3] is not None:
                    mem, i = tw[3][a]
                    if mem.idx == i:             # (still the row this agent's add_memory fills)
                        rr, ar = mem.rewards[i], mem.actions[i]
                        mem._prefilled = (i, (eng.actions[:, a] if values else action).data_ptr())     # (values: Agent.transition hands add_memory the record of what was taken)
                return eng.act(a, tw[1], action=action if direct else None, reward_row=rr, action_row=ar)
            self._turn_windows = None            # host code changed the world mid-turn: render on demand from here on
        if values:                               # (the older per-launch protocol has no action-value input: greedy, on the host)
            if agent.epsilon > 0.0:
                raise ValueError("in-kernel exploration (action values with epsilon > 0) needs the patched-window turn protocol "
                                 "(Environment.patch_windows = True and an observation tensor)")
            action = action.argmax(dim=1)
        eng.actions[:, a].copy_(action)          # one strided copy that also narrows int64 -> uint8
        nxt = a + 1 < len(self.agents) and eng.obs is not None
        slot = self._replay_slot(a + 1, a) if nxt else None
        eng.step(eng.actions, sweep=False, write_obs=False, agent_begin=a, agent_end=a + 1, turn=self.turn, obs_next=nxt,
                 obs_next_out=slot)
        self._fresh_obs = (a + 1, self.world.mutations, slot) if nxt else None
        return eng.rewards[:, a]

    # ------------------------------------------------------------------ step outputs (batched additions)
    @property
    def obs(self):
        """``[E, A, C, V, V]`` of the last turn -- or, for agents that differ in their specs, the list of each agent's own window
        tensor (``obs_of``)."""
        eng = self._ensure_engine()
        if self._mixed:
            return [self.obs_of(a) for a in range(len(self.agents))]
        return eng.obs

    @property
    def rewards(self):
        return self._ensure_engine().rewards

    @property
    def actions(self):
        return self._ensure_engine().actions

    @property
    def dones(self):
        """All-zero inside an epoch: ``world.is_done`` only flips after the turn loop
        (``environment.py:171``)."""
        return torch.zeros_like(self._ensure_engine().rewards)

    @property
    def total_reward(self):
        return self.world.total_reward

    # ------------------------------------------------------------------ model hooks (overridable, environment.py:95-105)
    def _model_start_epoch_action(self, agent: Agent, epoch: int):
        agent.model.start_epoch_action(epoch=epoch)

    def _model_end_epoch_action(self, agent: Agent, epoch: int):
        agent.model.end_epoch_action(epoch=epoch)

    def _model_train_step(self, agent: Agent):
        return agent.model.train_step()

    # ------------------------------------------------------------------ epoch loops (environment.py:108-300)
    def _output_dir(self, output_dir) -> Path:
        if output_dir is None:
            exp = self.config.experiment
            output_dir = Path(exp.output_dir) if hasattr(exp, "output_dir") or "output_dir" in exp else Path("./data/")
        output_dir = Path(output_dir)
        os.makedirs(output_dir, exist_ok=True)
        return output_dir

    def _cfg_model(self, key, default=None):
        model = getattr(self.config, "model", None) if not isinstance(self.config, dict) else self.config.get("model")
        if model is None:
            return default
        try:
            return model[key] if key in model else default
        except TypeError:
            return getattr(model, key, default)

    def run_experiment(self, animate: bool = False, logging: bool = True, logger=None, output_dir=None,
                       epochs: Optional[int] = None, max_turns: Optional[int] = None, all_reduce: bool = True):
        """``for epoch in range(epochs + 1)``: reset -> start-of-epoch hooks -> ``max_turns`` x take_turn ->
        ``world.is_done = True`` -> end-of-epoch hooks -> ``train_step`` per agent (the loss logged is the LAST
        agent's, as in the reference: assignment, not a sum) -> ``logger.record_turn(epoch, loss, reward, epsilon)``
        -> epsilon decay -> model checkpoint every ``record_period`` epochs when ``config.model.save_weights``
        (``sorrel/environment.py:148-211``).  The reward logged is the mean of ``world.total_reward`` over ALL
        envs of ALL ranks (the one RCCL all-reduce); the per-epoch metric dicts are returned.  ``animate`` is
        accepted for signature compatibility; sprite rendering is outside this engine."""
        from sorrel_amd import distributed as D

        exp = self.config.experiment
        epochs = int(exp.epochs) if epochs is None else epochs
        max_turns = int(exp.max_turns) if max_turns is None else max_turns
        record_period = int(exp.record_period) if (hasattr(exp, "record_period") or "record_period" in exp) else 1
        save_weights = bool(self._cfg_model("save_weights", False))
        decay = self._cfg_model("epsilon_decay", None)
        out_dir = self._output_dir(output_dir) if save_weights else None
        capture = bool(self.capture_turns or (exp.get("capture_turns", False) if hasattr(exp, "get") else getattr(exp, "capture_turns", False)))
        history = []
        for epoch in range(epochs + 1):
            self.reset()
            for agent in self.agents:
                self._model_start_epoch_action(agent, epoch)
            if all(getattr(a.model, "device_random", False) for a in self.agents) and not self.stop_if_done \
                    and type(self).take_turn is Environment.take_turn:
                self.rollout(max_turns - self.turn)        # the whole epoch in one engine call (a subclass that overrides
                                                           # take_turn gets its per-turn loop below, as in the reference)
            elif capture and self._captured is None and not self.stop_if_done and type(self).take_turn is Environment.take_turn \
                    and max_turns - self.turn > 2:
                capture = self.capture_turn(warmup=2) is not None      # (two real turns of this epoch; not tried again when it fails)
            while self.turn < max_turns:
                self.take_turn()
                if self.world.is_done and self.stop_if_done:
                    break
            self.world.is_done = True
            self.raise_on_status()
            m = D.rollout_metrics(self._ensure_engine(), all_reduce=all_reduce)
            for agent in self.agents:
                self._model_end_epoch_action(agent, epoch)
            total_loss = 0
            for agent in self.agents:
                total_loss = self._model_train_step(agent)
            m["loss"] = float(total_loss) if total_loss is not None else 0.0
            m["epsilon"] = float(getattr(self.agents[0].model, "epsilon", 0.0))
            history.append(m)
            if logging and logger is not None:
                logger.record_turn(epoch, total_loss, m["mean_total_reward"], m["epsilon"])
            for i, agent in enumerate(self.agents):
                if decay is not None:
                    agent.model.epsilon_decay(float(decay))
                if epoch % record_period == 0 and save_weights and hasattr(agent.model, "save"):
                    os.makedirs(out_dir / "checkpoints", exist_ok=True)
                    agent.model.save(out_dir / "checkpoints" / f"epoch{epoch}-agent-{i}.pkl")
        return history

    def generate_memories(self, num_games: int = 1000, animate: bool = False, output_dir=None,
                          record_positions: bool = False):
        """Play ``num_games`` games of ``max_turns`` turns with the existing models and write one replay file per
        agent, ``<output_dir>/memories/agent{i}.npz`` (``sorrel/environment.py:213-300``).

        File format = the reference's ``SavedGames.save`` (``sorrel/buffers.py:361-379``): ``states`` float32
        ``[N, *obs_shape]``, ``actions`` int64 ``[N]``, ``rewards`` / ``dones`` float32 ``[N]``, ``positions`` int64
        ``[N, 2]``, ``n_frames``, ``idx`` -- the reference's ``Buffer.load`` reads it.  The batch is laid out env
        by env: rows ``[e * G * T, (e + 1) * G * T)`` are env ``e``'s ``G`` games of ``T`` turns in play order, i.e.
        what the reference would have saved for that one world.

        Policy-driven agents (phased turns): after every game the agent's whole ``model.memory`` is appended with
        ``add_from_buffer``, the reference's own call -- so, as there, ``positions`` are stored only if that memory
        carries them, and a memory the model does not clear per game is appended again from its start
        (``sorrel/environment.py:297``, ``buffers.py:71-99``).  Device-random models (``RandomModel``): the turns run
        fused, the step kernel writes the observations straight into a device ring (``collect``) and every game is
        appended once; ``positions`` stay zero unless ``record_positions`` (then: each agent's cell after its move)."""
        from sorrel_amd.buffers import SavedGames, TurnBuffer

        out_dir = self._output_dir(output_dir)
        T = int(self.config.experiment.max_turns)
        E, A = self.num_envs, len(self.agents)
        saved = []
        for agent in self.agents:
            n_frames = getattr(agent.model, "n_frames", 1)
            obs_shape = tuple(agent.observation_spec.input_size)
            saved.append(SavedGames(capacity=num_games * T, obs_shape=obs_shape, n_frames=n_frames, num_envs=E,
                                    device="cpu", positions=(2,)))
            if hasattr(agent.model, "eval"):
                agent.model.eval()
        device_random = all(getattr(a.model, "device_random", False) for a in self.agents)
        # one engine call per game -- unless a subclass overrides take_turn: the reference's loop goes through take_turn
        # every turn (sorrel/environment.py:266-282), so an override (per-turn logging, extra world logic) must be called
        one_call = device_random and type(self).take_turn is Environment.take_turn
        exp = self.config.experiment
        capture = bool(self.capture_turns or (exp.get("capture_turns", False) if hasattr(exp, "get") else getattr(exp, "capture_turns", False)))
        ring = None
        for game in range(num_games):
            self.reset()
            for agent in self.agents:
                self._model_start_epoch_action(agent, game)
            eng = self._ensure_engine()
            if device_random:
                if ring is None:
                    ring = TurnBuffer(T, E, eng.spec.obs_shape, device=eng.device, obs_dtype=eng.obs_dtype,
                                      positions=record_positions)
                ring.clear()
                if one_call:
                    self.collect(T, ring)
                else:
                    while self.turn < T:
                        self.take_turn()
                        eng = self._ensure_engine()
                        ring.obs[ring.slot()].copy_(eng.obs)
                        ring.commit(eng.actions, eng.rewards, eng.agent_pos)
                        if self.world.is_done and self.stop_if_done:
                            break
                n = len(ring)
                for a, sg in enumerate(saved):
                    st, ac, rw, dn = ring.agent_view(a)
                    sg.add_turns(st[:n], ac[:n], rw[:n], dn[:n], positions=None if ring.positions is None else ring.positions[:n, :, a])
            else:
                if capture and self._captured is None and not self.stop_if_done and type(self).take_turn is Environment.take_turn and T - self.turn > 2:
                    capture = self.capture_turn(warmup=2) is not None      # (capture_turns: as in run_experiment)
                while self.turn < T:
                    self.take_turn()
                    if self.world.is_done and self.stop_if_done:
                        break
            self.world.is_done = True
            self.raise_on_status()
            for agent, sg in zip(self.agents, saved):
                self._model_end_epoch_action(agent, game)
                if not device_random:
                    sg.add_from_buffer(agent.model.memory)
        os.makedirs(out_dir / "memories", exist_ok=True)
        paths = []
        for i, sg in enumerate(saved):
            paths.append(out_dir / "memories" / f"agent{i}.npz")
            sg.save(paths[-1])
        return paths

    # ------------------------------------------------------------------ world-state checkpoint (the reference leaves
    # "# TODO: ability to save/load?" at sorrel/environment.py:107)
    def state_dict(self) -> dict:
        """Everything a rollout needs to continue bit-exactly: the grid, agent positions, ``total_reward``, the
        per-agent state / facing tensors, the epoch / turn counters, the RNG seed and the first global env id."""
        w = self.world
        eng = self._ensure_engine()
        sd = dict(version=1, grid=w.grid.cpu().clone(), agent_pos=w.agent_pos.cpu().clone(),
                  total_reward=w.total_reward.cpu().clone(), epoch=int(self.epoch), turn=int(self.turn),
                  seed=int(w.seed), first_env_id=int(getattr(w, "first_env_id", 0)), num_envs=int(w.num_envs),
                  shape=(w.layers, w.height, w.width), is_done=bool(w.is_done),
                  type_names=[type(p).__name__ for p in w.registry.prototypes])
        if eng.agent_state is not None:
            sd["agent_state"] = eng.agent_state.cpu().clone()
        if eng.agent_dir is not None:
            sd["agent_dir"] = eng.agent_dir.cpu().clone()
        return sd

    def load_state_dict(self, sd: dict) -> None:
        w = self.world
        eng = self._ensure_engine()
        if tuple(sd["shape"]) != (w.layers, w.height, w.width) or int(sd["num_envs"]) != w.num_envs:
            raise ValueError("checkpoint was taken from a world of another shape or batch size")
        if int(sd["seed"]) != int(w.seed) or int(sd["first_env_id"]) != int(getattr(w, "first_env_id", 0)):
            raise ValueError("checkpoint was taken with another seed / first global env id: the rollout would not continue bit-exactly")
        if list(sd["type_names"]) != [type(p).__name__ for p in w.registry.prototypes]:
            raise ValueError("checkpoint was taken with another entity type table")
        w.grid.copy_(sd["grid"].to(w.device))
        w.agent_pos.copy_(sd["agent_pos"].to(w.device))
        w.total_reward.copy_(sd["total_reward"].to(w.device))
        if "agent_state" in sd and eng.agent_state is not None:
            eng.agent_state.copy_(sd["agent_state"].to(w.device))
        if "agent_dir" in sd and eng.agent_dir is not None:
            eng.agent_dir.copy_(sd["agent_dir"].to(w.device))
        self.epoch, self.turn = int(sd["epoch"]), int(sd["turn"])
        eng.epoch, eng.turn = self.epoch, self.turn
        w.is_done = bool(sd.get("is_done", False))
        w.mutations += 1
        self._fresh_obs = None

    def save_checkpoint(self, path) -> None:
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path) -> None:
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))   # tensors and plain values only


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
            if not Environment._standard_hooks(agent) or type(agent).act is not MovingAgent.act or not isinstance(mem, Buffer) or mem.n_frames != 1 \
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

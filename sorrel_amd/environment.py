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
from sorrel_amd.epochs import EpochLoops
from sorrel_amd.turns import CapturedTurn, MixedSpecTurns, PolicyTurns, RecordedTurns, SpeculativeTurns, _FastPolicyTurn  # noqa: F401  (re-exported)

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


class Environment(SpeculativeTurns, MixedSpecTurns, PolicyTurns, RecordedTurns, EpochLoops):
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
                return {"loop": "speculative", "models": len(self._speculation_groups(eng)),
                        "resolve": "sgw_verify_rows (any agent rule: a whole turn per pass on a scratch state)" if getattr(self, "_spec_generic", False)
                        else "sgw_turn_resolve (plain movers)"}
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
        todo = {}
        for a in (self._value_agents if slots is None else slots):
            eps = min(1.0, max(0.0, self.agents[a].epsilon))
            if self._eps_pushed.get(a) != (eng.uid, eps):
                todo[a] = eps
        if not todo:
            return
        if len(todo) == len(self.agents) and len(set(todo.values())) == 1:      # every agent, one rate (agents that share a model): ONE launch, not A
            eng.turn_epsilon(next(iter(todo.values())), -1)
        else:
            for a, eps in todo.items():
                eng.turn_epsilon(eps, a)
        for a, eps in todo.items():
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

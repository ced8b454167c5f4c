"""Shared test plumbing: golden fixtures, spec conversions, the C oracle binding."""
from __future__ import annotations

import ctypes as C
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gridstep_oracle as O  # noqa: E402  (tests may use the oracle; the product may not)
from sorrel_amd import _native as N  # noqa: E402
from sorrel_amd.spec import WorldSpec  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


NON_WORLD_FIXTURES = {"buffer_ring", "buffer_saved_by_reference", "savedgames_by_reference",
                      "full_view_treasurehunt", "mixed_specs_treasurehunt"}   # fixtures that are not step-loop traces in the common format
INJECTED_FIXTURES = {"cleanup_15x16", "cleanup_21x31_default", "cleanup_13x12_r2"}  # worlds populated by host code: runs start from the stored grid0 / pos0


def golden_names():
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if n not in NON_WORLD_FIXTURES]


def load_golden(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    spec = oracle_spec_from_json(str(d["spec_json"]))
    return d, spec


def oracle_spec_from_json(text: str) -> O.Spec:
    j = json.loads(text)
    j["appearance"] = np.asarray(j["appearance"], dtype=np.float64)
    return O.Spec(**j)


def world_spec(spec: O.Spec) -> WorldSpec:
    """oracle Spec -> product WorldSpec (same fields, independent classes)."""
    return WorldSpec(
        height=spec.height, width=spec.width, layers=spec.layers, num_agents=spec.num_agents,
        vision_radius=spec.vision_radius, num_channels=spec.num_channels, agent_layer=spec.agent_layer,
        default_type=spec.default_type, fill_type=spec.fill_type,
        action_dy=list(spec.action_dy), action_dx=list(spec.action_dx), agent_type=list(spec.agent_type),
        type_value=list(spec.type_value), type_passable=list(spec.type_passable), type_rule=list(spec.type_rule),
        spawn_prob=list(spec.spawn_prob), spawn_choices=[list(c) for c in spec.spawn_choices],
        appearance=np.asarray(spec.appearance, dtype=np.float64), seed=spec.seed,
        layer_fill_type=list(spec.layer_fill_type), layer_border_type=list(spec.layer_border_type),
        dense_prob=spec.dense_prob, dense_choices=list(spec.dense_choices), obs_post=getattr(spec, "obs_post", 0),
        agent_rule=getattr(spec, "agent_rule", 0), tag_it_type=getattr(spec, "tag_it_type", 0),
        tag_notit_type=getattr(spec, "tag_notit_type", 0), tag_reward=getattr(spec, "tag_reward", 0.0),
        rule_layer=list(getattr(spec, "rule_layer", [])), rule_mask=list(getattr(spec, "rule_mask", [])),
        rule_become=list(getattr(spec, "rule_become", [])), action_kind=list(getattr(spec, "action_kind", [])),
        beam_radius=getattr(spec, "beam_radius", 0), clean_beam_type=getattr(spec, "clean_beam_type", 0),
        zap_beam_type=getattr(spec, "zap_beam_type", 0), beam_block_mask=getattr(spec, "beam_block_mask", 0),
        reward_total_factor=getattr(spec, "reward_total_factor", 1),
    )


# ----------------------------------------------------------------------------- C oracle
_olib = None


def oracle_lib():
    global _olib
    if _olib is None:
        import __graft_entry__ as g

        # the library is built by the session hook in conftest.py, BEFORE any test touches the GPU (a compiler child process started
        # from a process that holds the GPU is what this avoids); here it is only loaded -- or built when nothing has initialised HIP yet
        path = g.ORACLE_LIB
        if not g._up_to_date(g.ORACLE_LIB, g.ORACLE_SRC, os.path.join(ROOT, "include", "sgw.h")):
            import torch

            if torch.cuda.is_initialized():
                raise RuntimeError("oracle/libgridstep_oracle.so is missing or stale and this process already holds the GPU: "
                                   "run `python -c 'import __graft_entry__ as g; g.build()'` first")
            path = g.build_oracle()
        lib = C.CDLL(path)
        lib.sgo_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.sgo_rng_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.sgo_rng_u32.restype = C.c_uint32
        vp = C.c_void_p
        cfgp = C.POINTER(N.SgwConfig)
        lib.sgo_reset.argtypes = [cfgp, vp, vp, vp, C.c_uint32, C.c_int, vp]
        lib.sgo_observe.argtypes = [cfgp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int]
        lib.sgo_step.argtypes = [cfgp, vp, vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32,
                                 C.c_uint32, C.c_int, vp, vp, vp]
        lib.sgo_init_agent_state.argtypes = [cfgp, vp]
        lib.sgo_random_actions.argtypes = [cfgp, vp, C.c_uint32, C.c_uint32]
        lib.sgo_reduce_metrics.argtypes = [cfgp, vp, vp]
        lib.sgo_threads.argtypes = [C.c_int]
        lib.sgo_rollout.argtypes = [cfgp, vp, vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp]
        _olib = lib
    return _olib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class COracle:
    """Batch state + the C restatement, with the engine's tensor layouts."""

    def __init__(self, wspec: WorldSpec, num_envs: int, first_env_id: int = 0, threads: int = 0):
        self.lib = oracle_lib()
        self.spec = wspec
        self.cfg = wspec.to_config(num_envs, first_env_id)
        self.threads = threads
        E, A = num_envs, wspec.num_agents
        self.grid = np.zeros((E, wspec.layers, wspec.height, wspec.width), np.uint8)
        self.pos = np.zeros((E, A, 2), np.uint8)
        self.actions = np.zeros((E, A), np.uint8)
        self.obs = np.zeros((E,) + wspec.obs_shape, np.float32)
        self.rewards = np.zeros((E, A), np.float32)
        self.total = np.zeros((E,), np.float64)
        self.agent_state = np.zeros((E, A), np.uint8)
        self.state_at_pov = np.zeros((E, A), np.uint8)
        self.agent_dir = np.full((E, A), 2, np.uint8)       # CleanupAgent starts facing down
        self.lib.sgo_init_agent_state(C.byref(self.cfg), _p(self.agent_state))

    def reset(self, epoch=0):
        self.lib.sgo_reset(C.byref(self.cfg), _p(self.grid), _p(self.pos), _p(self.total), epoch, self.threads,
                           _p(self.agent_state))

    def observe(self, a0=0, a1=None):
        a1 = self.spec.num_agents if a1 is None else a1
        self.lib.sgo_observe(C.byref(self.cfg), _p(self.grid), _p(self.pos), _p(self.obs), a0, a1, self.threads)
        return self.obs

    def step(self, epoch, turn, actions=None, random_actions=False, sweep=True, write_obs=True, a0=0, a1=None):
        a1 = self.spec.num_agents if a1 is None else a1
        if actions is not None:
            self.actions[...] = actions
        flags = (1 if sweep else 0) | (2 if random_actions else 0) | (0 if write_obs else 4)
        return self.lib.sgo_step(C.byref(self.cfg), _p(self.grid), _p(self.pos), _p(self.actions),
                                 _p(self.obs) if write_obs else None, _p(self.rewards), _p(self.total),
                                 epoch, turn, a0, a1, flags, self.threads, _p(self.agent_state), _p(self.state_at_pov),
                                 _p(self.agent_dir))

    def metrics(self):
        out = np.zeros(4, np.float64)
        self.lib.sgo_reduce_metrics(C.byref(self.cfg), _p(self.total), _p(out))
        return out


def oracle_spec(ws: WorldSpec) -> O.Spec:
    """product WorldSpec -> oracle Spec (the inverse of world_spec)."""
    return O.Spec(
        height=ws.height, width=ws.width, layers=ws.layers, num_agents=ws.num_agents,
        vision_radius=ws.vision_radius, num_types=ws.num_types, num_channels=ws.num_channels,
        agent_layer=ws.agent_layer, default_type=ws.default_type, fill_type=ws.fill_type,
        action_dy=list(ws.action_dy), action_dx=list(ws.action_dx), agent_type=list(ws.agent_type),
        type_value=list(ws.type_value), type_passable=list(ws.type_passable), type_rule=list(ws.type_rule),
        spawn_prob=list(ws.spawn_prob), spawn_choices=[list(c) for c in ws.spawn_choices],
        appearance=np.asarray(ws.appearance, dtype=np.float64), seed=ws.seed,
        layer_fill_type=list(ws.layer_fill_type), layer_border_type=list(ws.layer_border_type),
        dense_prob=ws.dense_prob, dense_choices=list(ws.dense_choices), obs_post=ws.obs_post,
        agent_rule=ws.agent_rule, tag_it_type=ws.tag_it_type, tag_notit_type=ws.tag_notit_type, tag_reward=ws.tag_reward,
        rule_layer=list(ws.rule_layer), rule_mask=list(ws.rule_mask), rule_become=list(ws.rule_become),
        action_kind=list(ws.action_kind), beam_radius=ws.beam_radius, clean_beam_type=ws.clean_beam_type,
        zap_beam_type=ws.zap_beam_type, beam_block_mask=ws.beam_block_mask, reward_total_factor=ws.reward_total_factor,
    )


def random_rule_world(rng):
    """A random layered world for the widened rule set: 2-4 layers, spawners and BECOME_IF rules with random
    layer / mask / successor tables, timers (unconditional become chains), Cleanup or Tag agents, random map."""
    from sorrel_amd.spec import WorldSpec, action_deltas, NO_BORDER
    from sorrel_amd import _native as N

    L = int(rng.integers(2, 5))
    h, w = int(rng.integers(6, 26)), int(rng.integers(6, 26))
    r = int(rng.integers(1, min((min(h, w) - 1) // 2, 4) + 1))
    a = int(rng.integers(1, min(10, (h - 2) * (w - 2) // 2) + 1))
    zA = int(rng.integers(0, L - 1))                      # a layer above the agents always exists
    mode = rng.choice(["cleanup", "move", "tag"], p=[0.5, 0.3, 0.2])
    cleanup = mode == "cleanup"
    # types: 0 empty (passable), 1 wall, 2..T-2 random things, T-1 agent (+ T-2 second agent type for Tag)
    T = int(rng.integers(6, 14))
    agent_t = T - 1
    C = int(rng.integers(2, 9))
    app = np.zeros((T, C))
    for t in range(1, T):
        app[t, int(rng.integers(0, C))] = 1.0 if rng.random() < 0.8 else float(rng.integers(2, 5))
    value = [0.0, -1.0] + [float(rng.choice([0, 1, -1, 5, 0.5])) for _ in range(T - 3)] + [0.0]
    passable = [1, 0] + [int(rng.random() < 0.5) for _ in range(T - 3)] + [0]
    rule, sp, sc = [0] * T, [0.0] * T, [[] for _ in range(T)]
    rl, rm, rb = [0] * T, [0] * T, [0] * T
    for t in range(2, T - 1):
        u = rng.random()
        if u < 0.3:
            rule[t] = N.RULE_SPAWN
            sp[t] = float(rng.choice([0.02, 0.2, 1.0]))
            sc[t] = [int(x) for x in rng.integers(0, T - 1, size=int(rng.integers(1, 4)))]
        elif u < 0.65:
            rule[t] = N.RULE_BECOME_IF
            rl[t] = int(rng.integers(-1, L))
            rm[t] = int(rng.integers(0, 1 << T))
            rb[t] = int(rng.integers(0, T - 1))
    if mode == "tag":               # two impassable agent types: T-2 = "it", T-1 = not "it"
        rule[T - 2], sp[T - 2], sc[T - 2], value[T - 2], passable[T - 2] = 0, 0.0, [], 0.0, 0
    names = ["up", "down", "left", "right"] + (["clean", "zap"] if cleanup else ["stay"])
    dy, dx = action_deltas(names)
    kw = {}
    if cleanup:
        kw = dict(agent_rule=2, action_kind=[0, 0, 0, 0, 1, 2], beam_radius=int(rng.integers(0, 5)),
                  clean_beam_type=int(rng.integers(2, T - 1)), zap_beam_type=int(rng.integers(2, T - 1)),
                  beam_block_mask=int(rng.integers(0, 1 << T)) | 2, reward_total_factor=int(rng.integers(1, 3)))
    if mode == "tag":
        kw = dict(agent_rule=1, tag_it_type=T - 2, tag_notit_type=T - 1, tag_reward=float(rng.choice([10, 1, 0.5])))
    ws = WorldSpec(height=h, width=w, layers=L, num_agents=a, vision_radius=r, num_channels=C, agent_layer=zA,
                   default_type=0, fill_type=1, action_dy=dy, action_dx=dx, agent_type=[agent_t] * a,
                   type_value=value, type_passable=passable, type_rule=rule, spawn_prob=sp, spawn_choices=sc,
                   appearance=app, seed=int(rng.integers(0, 2**40)), layer_fill_type=[0] * L,
                   layer_border_type=[1] * L, rule_layer=rl, rule_mask=rm, rule_become=rb, **kw)
    # the map: walls around every layer, random things inside, agents on distinct interior cells of their layer
    g = np.zeros((L, h, w), np.uint8)
    g[:, 0, :] = g[:, -1, :] = g[:, :, 0] = g[:, :, -1] = 1
    inner = rng.random((L, h - 2, w - 2)) < 0.35
    g[:, 1:-1, 1:-1] = np.where(inner, rng.integers(2, T - 2, size=inner.shape), 0).astype(np.uint8)
    cells = rng.permutation((h - 2) * (w - 2))[:a]
    pos = np.stack([cells // (w - 2) + 1, cells % (w - 2) + 1], axis=1).astype(np.uint8)
    for y, x in pos:
        g[zA, y, x] = agent_t
    return ws, g, pos


# ----------------------------------------------------------------------------- agents that differ (round 5)
TH_KINDS = ["EmptyEntity", "EmptyEntity", "Wall", "Gem", "Bone", "Food", "TreasurehuntAgent"]   # kind of each Treasurehunt type id
_MOVES = {"up": (-1, 0), "down": (1, 0), "left": (0, -1), "right": (0, 1)}


def load_mixed(name="mixed_specs_treasurehunt"):
    """(arrays, base oracle Spec, [per-agent view Spec], [full_view flag], [agent definition dict]) of a fixture whose agents
    each hold their own observation / action specs (oracle/make_golden.py round5)."""
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    base = oracle_spec_from_json(str(d["spec_json"]))
    defs = json.loads(str(d["agents_json"]))
    views, full = [], []
    for a in defs:
        app = np.stack([np.asarray(a["entity_map"][k], dtype=np.float64) for k in TH_KINDS])
        views.append(O.agent_view(base, a["radius"], app, TH_KINDS.index(a["fill"]),
                                  [_MOVES.get(n, (0, 0))[0] for n in a["actions"]], [_MOVES.get(n, (0, 0))[1] for n in a["actions"]]))
        full.append(bool(a["full_view"]))
    return d, base, views, full, defs

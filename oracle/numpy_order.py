"""TEST INFRASTRUCTURE ONLY -- "numpy-order" mode of the CPU restatement.

The stock Treasurehunt example draws from the global legacy ``np.random``
stream, consumed in sweep / agent order:

* one ``np.random.random()`` per spawning cell per turn, in ``ndenumerate``
  order (y, x, z)                       examples/treasurehunt/entities.py:73
* ``np.random.choice`` over 3 objects on success                  ...:75-84
* ``np.random.randint(0, n_actions)`` per agent, between its observation and
  its move                                   sorrel/models/base_model.py:111
* ``np.random.choice(n, size=A, replace=False)`` for agent placement
                                      examples/treasurehunt/env.py:140-142

That stream cannot be batched, so it is kept CPU-only (BASELINE config 1
plumbing): this module replays the same calls in the same order on the
object-free world of ``gridstep_oracle`` so that a seeded *unmodified*
Treasurehunt run of the reference is matched draw for draw
(``tests/golden/stock_np_random.npz``).
"""
from __future__ import annotations

import numpy as np

from oracle import gridstep_oracle as O


def reset_numpy_order(spec: O.Spec) -> O.EnvState:
    H, W, L = spec.height, spec.width, spec.layers
    grid = np.zeros((L, H, W), dtype=np.uint8)
    for z in range(L):
        grid[z] = spec.layer_fill_type[z]
        b = spec.layer_border_type[z]
        if b != O.NO_BORDER:
            grid[z, 0, :] = b
            grid[z, H - 1, :] = b
            grid[z, :, 0] = b
            grid[z, :, W - 1] = b
    valid = [(y, x) for y in range(1, H - 1) for x in range(1, W - 1)]
    chosen = np.random.choice(len(valid), size=spec.num_agents, replace=False)
    pos = np.array([valid[i] for i in chosen], dtype=np.int64)
    for a in range(spec.num_agents):
        grid[spec.agent_layer, pos[a, 0], pos[a, 1]] = spec.agent_type[a]
    return O.EnvState(grid=grid, pos=pos, total_reward=0.0)


def step_numpy_order(spec: O.Spec, st: O.EnvState):
    H, W, L = spec.height, spec.width, spec.layers
    agent_types = set(int(t) for t in spec.agent_type)
    for y in range(H):
        for x in range(W):
            for z in range(L):
                t = int(st.grid[z, y, x])
                if t in agent_types or spec.type_rule[t] != O.RULE_SPAWN:
                    continue
                if np.random.random() < spec.spawn_prob[t]:
                    ch = spec.spawn_choices[t]
                    st.grid[z, y, x] = ch[int(np.random.choice(len(ch)))]
    A, C, V = spec.num_agents, spec.num_channels, spec.window
    obs = np.zeros((A, C, V, V), dtype=np.float32)
    actions = np.zeros(A, dtype=np.int64)
    rewards = np.zeros(A, dtype=np.float32)
    z = spec.agent_layer
    for a in range(A):
        y, x = int(st.pos[a, 0]), int(st.pos[a, 1])
        obs[a] = O.visual_field(spec, st.grid, y, x).astype(np.float32)
        act = int(np.random.randint(0, len(spec.action_dy)))
        ny, nx = y + int(spec.action_dy[act]), x + int(spec.action_dx[act])
        target = int(st.grid[z, ny, nx])
        reward = spec.type_value[target]
        if spec.type_passable[target]:
            st.grid[z, ny, nx] = spec.agent_type[a]
            st.grid[z, y, x] = spec.default_type
            st.pos[a] = (ny, nx)
        st.total_reward += reward
        actions[a], rewards[a] = act, reward
    return obs, actions, rewards


def rollout_numpy_order(spec: O.Spec, turns: int, np_seed: int):
    np.random.seed(np_seed)
    st = reset_numpy_order(spec)
    A, C, V = spec.num_agents, spec.num_channels, spec.window
    out = dict(
        grid0=st.grid[None].copy(), pos0=st.pos[None].astype(np.uint8),
        obs=np.zeros((turns, 1, A, C, V, V), np.float32), actions=np.zeros((turns, 1, A), np.uint8),
        rewards=np.zeros((turns, 1, A), np.float32), total_reward=np.zeros((turns, 1), np.float64),
        grid=np.zeros((turns, 1, L_(spec), spec.height, spec.width), np.uint8),
        pos=np.zeros((turns, 1, A, 2), np.uint8),
    )
    for t in range(turns):
        o, a, r = step_numpy_order(spec, st)
        out["obs"][t, 0], out["actions"][t, 0], out["rewards"][t, 0] = o, a, r
        out["total_reward"][t, 0] = st.total_reward
        out["grid"][t, 0] = st.grid
        out["pos"][t, 0] = st.pos
    return out


def L_(spec: O.Spec) -> int:
    return spec.layers

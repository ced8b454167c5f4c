#!/usr/bin/env python3
"""CPU study (round-4 review, item 4): can the agent-after-agent policy turn be evaluated SPECULATIVELY?

The reference steps its agents strictly one after another (sorrel/agents/agent.py:155-173): agent j's window shows the moves of agents
< j.  With a deterministic policy the sequential turn is the fixed point of

    pass 1   every agent's action from its window of the grid BEFORE anyone moves (one batched policy evaluation);
    resolve  apply the current actions in agent order; agent j is DIRTY if the window it really has when its turn comes differs from
             the one its action was computed on (an earlier agent's move changed a cell inside it);
    pass k   re-evaluate the dirty agents on their true windows, resolve again ... until nobody is dirty.

The first dirty agent of an env moves to a higher index every pass, so at most A passes are needed and the result is exact.  This script
measures, on the oracle (oracle/gridstep_oracle.c through tests/helpers.COracle; windows by the closed form, checked against
oracle/gridstep_oracle.py), how many passes envs need and how many (env, agent) pairs are re-evaluated per pass -- for a linear
argmax policy over the window and for a uniformly random one -- on BASELINE config 3's and config 5's shapes.  No GPU.
usage: tools/speculation_study.py [envs=1024] [turns=8]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gridstep_oracle as O          # noqa: E402  (a study tool, like the tests: the product never imports the oracle)
from tests import helpers as H                   # noqa: E402
from sorrel_amd.spec import treasurehunt_spec    # noqa: E402


def windows_of(grid_pad, app, pos, r):
    """[A, C, V, V] float32 windows of one env from its padded type grid [L, H + 2r, W + 2r] (pad = 255 -> fill row of `app`)."""
    V = 2 * r + 1
    out = np.empty((len(pos), app.shape[1], V, V), np.float32)
    for a, (y, x) in enumerate(pos):
        t = grid_pad[:, y:y + V, x:x + V]                       # [L, V, V] type ids (255 outside the map)
        w = app[t[0]]
        for z in range(1, t.shape[0]):
            w = w + np.where((t[0] == 255)[..., None], 0.0, app[t[z]])   # out of bounds: the fill appearance ONCE (visual_field.py:89-94)
        out[a] = np.transpose(w, (2, 0, 1))
    return out


def study(name, h, w, A, r, E, turns, policy_kind, seed=0, p=0.005, dense=0.0, check=4):
    ws = treasurehunt_spec(h, w, A, r, spawn_prob=p, seed=seed, dense_prob=dense)
    sp = H.oracle_spec(ws)
    co = H.COracle(ws, E, first_env_id=0)
    co.reset(0)
    for t in range(1, 21):                                      # a played-in world (spawns accumulated, agents spread out)
        co.step(0, t, random_actions=True)
    V, C = 2 * r + 1, ws.num_channels
    app = np.zeros((256, C), np.float64)
    app[:ws.num_types] = np.asarray(ws.appearance)
    app[255] = app[ws.fill_type]
    rng = np.random.default_rng(7)
    Wt = rng.standard_normal((A, C * V * V, len(ws.action_dy))).astype(np.float32)     # a linear policy per agent
    dy, dx = np.asarray(ws.action_dy), np.asarray(ws.action_dx)
    passable = np.asarray(ws.type_passable, bool)
    zA, agent_t, dflt = ws.agent_layer, ws.agent_type[0], ws.default_type

    def policy(win, a, env, turn):
        if policy_kind == "random":                             # (what a stochastic policy keyed by (env, turn, agent) does: the window does not matter)
            return int(O.categorical(O.rng_u32(ws.seed, env, 0, turn, O.STREAM_ACTION, a), len(dy)))
        return int(np.argmax(win.reshape(-1) @ Wt[a]))

    passes_hist = np.zeros(A + 2, np.int64)
    reeval = np.zeros(A + 2, np.int64)                          # (env, agent) evaluations in pass k (k = 1: all of them)
    changed_actions = 0
    t0 = time.time()
    for turn in range(21, 21 + turns):
        co.step(0, turn, actions=np.zeros((E, A), np.uint8), sweep=True, a0=0, a1=0, write_obs=False)     # the sweep alone
        acts_all = np.zeros((E, A), np.uint8)
        for e in range(E):
            g0 = co.grid[e]
            pos0 = co.pos[e].astype(np.int64)
            pad = np.full((ws.layers, h + 2 * r, w + 2 * r), 255, np.uint8)
            pad[:, r:r + h, r:r + w] = g0
            based = windows_of(pad, app, pos0, r)               # pass 1: everybody on the pre-move grid
            act = np.array([policy(based[a], a, e, turn) for a in range(A)])
            reeval[1] += A
            npass = 1
            while True:
                # resolve: apply the current actions in order on a scratch copy; find the agents whose true window differs
                g = pad.copy()
                pos = pos0.copy()
                dirty = []
                for a in range(A):
                    y, x = pos[a]
                    true_w = None
                    # cheap test first: did any earlier mover touch this window?  (the exact comparison decides)
                    if a and touched[:a].any() and ((np.abs(cells[:2 * a, 0] - y) <= r) & (np.abs(cells[:2 * a, 1] - x) <= r) & live[:2 * a]).any():
                        true_w = windows_of(g, app, pos[a:a + 1], r)[0]
                        if not np.array_equal(true_w, based[a]):
                            dirty.append((a, true_w))
                    if a == 0:
                        cells = np.zeros((2 * A, 2), np.int64)
                        live = np.zeros(2 * A, bool)
                        touched = np.zeros(A, bool)
                    ny, nx = y + dy[act[a]], x + dx[act[a]]
                    tt = g[zA, ny + r, nx + r]
                    if passable[tt]:
                        g[zA, ny + r, nx + r] = agent_t
                        g[zA, y + r, x + r] = dflt
                        pos[a] = (ny, nx)
                        cells[2 * a], cells[2 * a + 1] = (y, x), (ny, nx)
                        live[2 * a] = live[2 * a + 1] = True
                        touched[a] = True
                if not dirty:
                    break
                npass += 1
                reeval[npass] += len(dirty)
                for a, tw in dirty:
                    based[a] = tw
                    na = policy(tw, a, e, turn)
                    changed_actions += int(na != act[a])
                    act[a] = na
            passes_hist[npass] += 1
            acts_all[e] = act
        if check and turn == 21:                               # exactness: the fixed point IS the sequential turn (Python restatement, a few envs)
            for e in range(min(check, E)):
                st = O.EnvState(grid=co.grid[e].copy(), pos=co.pos[e].astype(np.int64).copy(), total_reward=0.0)
                st.agent_state = np.asarray(sp.agent_type, np.uint8).copy()
                got = []
                for a in range(A):
                    y, x = int(st.pos[a, 0]), int(st.pos[a, 1])
                    win = O.visual_field(sp, st.grid, y, x).astype(np.float32)
                    k = policy(win, a, e, turn)
                    got.append(k)
                    O.act_agent(sp, st, a, k)
                assert got == [int(v) for v in acts_all[e]], (name, e, got, acts_all[e].tolist())
        co.step(0, turn, actions=acts_all, sweep=False)         # the agents act (sequentially, in the oracle) with the fixed-point actions
    n = passes_hist.sum()
    cum = np.cumsum(passes_hist) / n
    p99 = int(np.searchsorted(cum, 0.99) )
    print(f"{name:28s} {policy_kind:7s} envs x turns = {E} x {turns}: passes mean {np.dot(np.arange(A + 2), passes_hist) / n:.2f}, "
          f"median {int(np.searchsorted(cum, 0.5))}, 99th pct {p99}, max {int(np.nonzero(passes_hist)[0].max())}   "
          f"[{time.time() - t0:.0f} s]")
    print("    envs needing exactly k passes, k = 1..: " + " ".join(f"{v / n:.3f}" for v in passes_hist[1:p99 + 3]))
    print("    (env, agent) pairs evaluated in pass k / (E x A): " + " ".join(f"{v / (n * A):.4f}" for v in reeval[1:p99 + 3])
          + f"   total {reeval.sum() / (n * A):.3f} evaluations per agent-step; re-evaluations that changed the action: "
          f"{changed_actions / max(1, reeval[2:].sum()):.2f}")
    return passes_hist, reeval


if __name__ == "__main__":
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    for kind in ("linear", "random"):
        study("config 3: 32x32, 8 agents, 7x7", 32, 32, 8, 3, E, T, kind)
        study("config 5: 128x128, 64 agents, 11x11", 128, 128, 64, 5, E, max(2, T // 4), kind, p=0.05, dense=0.25)

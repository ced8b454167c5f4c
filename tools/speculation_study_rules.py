#!/usr/bin/env python3
"""CPU study (round-5 review, item 5): would the SPECULATIVE policy turn (tools/speculation_study.py, sgw_turn_resolve) pay for Tag and Cleanup agents?

Same protocol, any agent rule: pass 1 evaluates every agent on what it would observe BEFORE anyone acts (window + what its pov appends: Tag's "it"
flag, Cleanup's positional code); a resolve plays the current actions in agent order with the reference's act (oracle/gridstep_oracle.py: act_agent /
act_cleanup) and marks agent j DIRTY if what it really observes when its turn comes differs from what its action was computed on; dirty agents are
re-evaluated, until nobody is dirty.  What makes these rule sets harder than plain movers: a Tag window (9x9 on an 11x11 map) shows nearly the
whole world and a tag also flips the victim's flag; a Cleanup act writes up to 3 R beam cells on the layer above and beams age in the next sweep.
Worlds: the examples as shipped -- Tag 11x11 / 5 agents / 9x9 windows (tests/golden/tag_11x11_default.npz's spec), Cleanup 21x31x3 / 10 agents /
11x11 windows (cleanup_21x31_default.npz's spec and populated start grid) -- and larger variants.  Policies: a linear argmax over the observation,
and a random one keyed by (env, turn, agent) (what a fully exploring policy does: never dirty).  No GPU.
usage: tools/speculation_study_rules.py [envs=96] [turns=6]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gridstep_oracle as O          # noqa: E402  (a study tool, like the tests: the product never imports the oracle)
from tests import helpers as H                   # noqa: E402


def observe(sp, st, a, code):
    """What agent a's pov returns now: the flattened window + its tail."""
    y, x = int(st.pos[a, 0]), int(st.pos[a, 1])
    w = O.visual_field_closed_form(sp, st.grid, y, x).astype(np.float32).reshape(-1)
    if sp.agent_rule == O.AGENT_RULE_TAG:
        return np.concatenate([w, [1.0 if st.agent_state[a] == sp.tag_it_type else 0.0]]).astype(np.float32)
    if code is not None:
        return np.concatenate([w, code[y, x]]).astype(np.float32)
    return w


def clone(st):
    return O.EnvState(grid=st.grid.copy(), pos=st.pos.copy(), total_reward=st.total_reward,
                      agent_state=None if st.agent_state is None else st.agent_state.copy(),
                      agent_dir=None if st.agent_dir is None else st.agent_dir.copy())


def study(name, sp, start, E, turns, policy_kind, code_len=0):
    A, nact = sp.num_agents, len(sp.action_dy)
    rng = np.random.default_rng(7)
    code = rng.standard_normal((sp.height, sp.width, code_len)).astype(np.float32) if code_len else None
    n_obs = sp.num_channels * sp.window ** 2 + (1 if sp.agent_rule == O.AGENT_RULE_TAG else code_len)
    Wt = rng.standard_normal((A, n_obs, nact)).astype(np.float32)
    sweep = O.sweep if O._has_become_rules(sp) else O.sweep_vectorised

    def policy(obs, a, env, turn):
        if policy_kind == "random":
            return int(O.categorical(O.rng_u32(sp.seed, env, 0, turn, O.STREAM_ACTION, a), nact))
        return int(np.argmax(obs @ Wt[a]))

    passes_hist = np.zeros(A + 3, np.int64)
    reeval = np.zeros(A + 3, np.int64)
    t0 = time.time()
    envs = [start(e) for e in range(E)]
    for e, st in enumerate(envs):                               # played-in worlds: 10 random turns
        for t in range(1, 11):
            O.step_env(sp, st, e, 0, t)
    for turn in range(11, 11 + turns):
        for e, st in enumerate(envs):
            sweep(sp, st, e, 0, turn)
            based = [observe(sp, st, a, code) for a in range(A)]
            act = [policy(based[a], a, e, turn) for a in range(A)]
            reeval[1] += A
            npass = 1
            while True:
                trial = clone(st)
                dirty = []
                for a in range(A):
                    true = observe(sp, trial, a, code)
                    if not np.array_equal(true, based[a]):
                        dirty.append((a, true))
                    trial.total_reward += O.act_agent(sp, trial, a, act[a])
                if not dirty:
                    break
                npass += 1
                reeval[npass] += len(dirty)
                for a, tw in dirty:
                    based[a] = tw
                    act[a] = policy(tw, a, e, turn)
            passes_hist[npass] += 1
            # exactness: the fixed point is the sequential turn
            seq = clone(st)
            for a in range(A):
                k = policy(observe(sp, seq, a, code), a, e, turn)
                assert k == act[a], (name, e, turn, a)
                seq.total_reward += O.act_agent(sp, seq, a, k)
            assert np.array_equal(seq.grid, trial.grid) and np.array_equal(seq.pos, trial.pos)
            envs[e] = seq
    n = passes_hist.sum()
    cum = np.cumsum(passes_hist) / n
    p99 = int(np.searchsorted(cum, 0.99))
    print(f"{name:44s} {policy_kind:7s} envs x turns = {E} x {turns}: passes mean {np.dot(np.arange(A + 3), passes_hist) / n:.2f}, "
          f"median {int(np.searchsorted(cum, 0.5))}, 99th pct {p99}, max {int(np.nonzero(passes_hist)[0].max())}   [{time.time() - t0:.0f} s]", flush=True)
    print("    envs needing exactly k passes, k = 1..: " + " ".join(f"{v / n:.3f}" for v in passes_hist[1:p99 + 2]))
    print("    (env, agent) pairs evaluated in pass k / (E x A): " + " ".join(f"{v / (n * A):.4f}" for v in reeval[1:p99 + 2])
          + f"   total {reeval.sum() / (n * A):.3f} evaluations per agent-step (the sequential turn: 1.000, in A = {A} dependent batches)", flush=True)


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    turns = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    d, tag = H.load_golden("tag_11x11_default")

    def tag_start(sp):
        def start(e):
            st = O.reset_env(sp, e, 0)
            if st.agent_state is None:
                st.agent_state = O.init_agent_state(sp, e)
            return st
        return start

    def tag_variant(h, w, a, r):
        import copy
        sp = copy.deepcopy(tag)
        sp.height, sp.width, sp.num_agents, sp.vision_radius, sp.agent_type = h, w, a, r, [tag.agent_type[0]] * a
        return sp

    dc, cl = H.load_golden("cleanup_21x31_default")
    g0, p0 = np.asarray(dc["grid0"]), np.asarray(dc["pos0"])
    g0, p0 = (g0[0] if g0.ndim == 4 else g0), (p0[0] if p0.ndim == 3 else p0)

    def cleanup_start(e):
        return O.EnvState(grid=g0.copy(), pos=p0.astype(np.int64).copy(), total_reward=0.0,
                          agent_state=np.asarray(cl.agent_type, np.uint8).copy(), agent_dir=np.full(cl.num_agents, 2, np.uint8))

    print(f"# tools/speculation_study_rules.py {E} {turns}")
    for kind in ("linear", "random"):
        study("Tag 11x11, 5 agents, 9x9 (as shipped)", tag, tag_start(tag), E, turns, kind)
        big = tag_variant(32, 32, 8, 3)
        study("Tag 32x32, 8 agents, 7x7", big, tag_start(big), E, turns, kind)
        study("Cleanup 21x31x3, 10 agents, 11x11 (as shipped)", cl, cleanup_start, max(8, E // 3), turns, kind, code_len=12)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Round 6 (review item 7): what does ONE sgw_act launch cost behind what?  Config 3's shape, windows in per-agent rows (the replay-row protocol of
Environment.take_turn).  Every variant plays whole turns -- sweep + rows (one launch), then per agent [something], sgw_act -- and reads the engine's
own per-launch HIP-event timers (sgw_set_timing: an event pair around every engine launch on the launch stream; the time of a launch runs from the end
of whatever preceded it on the stream to the end of its kernel), averaged per act over the eight agents and all turns:
  back_to_back        nothing between the acts
  empty_kernel        a one-element torch op before every act (a dependent-launch boundary and nothing else)
  gemm_unused         the policy's matmul over the agent's [E, 294] rows + argmax before every act, the act takes PRECOMPUTED actions
  gemm_used           ... and the act takes the argmax's output (what Environment.take_turn does)
  gemm_values         the act takes the matmul's [E, 4] action values itself (SGW_ACT_QF32: no argmax launch)
  gemm_other_tensor   the same matmul + argmax over ANOTHER tensor of the rows' size (same cache footprint, no data dependence on this turn's rows)
  fill_77mb           a fill_ of a scratch tensor of the rows' size before every act (cache pressure from stores instead of loads)
  read_77mb           a sum over a scratch tensor of the rows' size (a streaming read, like the matmul's, without the matmul)
usage: tools/act_after_probe.py [envs ...]   (default 65536 16384 1024)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec


def run(E, turns=40):
    ws = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=0)
    A = ws.num_agents
    eng = GridEngine(ws, E, device="cuda:0", allocate_obs=False)
    eng.reset(0)
    Nw = int(np.prod(ws.obs_shape[1:]))
    dests = [torch.zeros((E, Nw), device="cuda:0") for _ in range(A)]
    rows = eng.window_rows(dests)
    W = torch.randn((Nw, ws.num_actions), device="cuda:0")
    other = torch.randn((E, Nw), device="cuda:0")
    scratch = torch.empty((E, Nw), device="cuda:0")
    one = torch.zeros((1,), device="cuda:0")
    pre = torch.randint(0, ws.num_actions, (A, E), device="cuda:0", dtype=torch.int64)

    def between(kind, a):
        """What runs on the stream before agent a's act; returns the act's action argument."""
        if kind == "back_to_back":
            return pre[a]
        if kind == "empty_kernel":
            one.add_(1.0)
            return pre[a]
        if kind in ("gemm_unused", "gemm_used", "gemm_values"):
            q = dests[a] @ W
            if kind == "gemm_values":
                return q
            k = q.argmax(dim=1)
            return k if kind == "gemm_used" else pre[a]
        if kind == "gemm_other_tensor":
            (other @ W).argmax(dim=1)
            return pre[a]
        if kind == "fill_77mb":
            scratch.fill_(1.0)
            return pre[a]
        if kind == "read_77mb":
            other.sum()
            return pre[a]
        raise ValueError(kind)

    out = {}
    for kind in ("back_to_back", "empty_kernel", "gemm_unused", "gemm_used", "gemm_values", "gemm_other_tensor", "fill_77mb", "read_77mb"):
        def turn():
            eng.turn += 1                                    # (epsilon is 0: the value acts need no turn state)
            eng.sweep_observe_rows(rows, sweep=True, turn=eng.turn)
            for a in range(A):
                eng.act(a, rows, action=between(kind, a))
        for _ in range(60):
            turn()
        torch.cuda.synchronize()
        eng.set_timing(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(turns):
            turn()
        e1.record()
        torch.cuda.synchronize()
        ms = eng.step_times_ms()
        eng.set_timing(False)
        n_per = 1 + A
        launches = [ms[i::n_per] for i in range(n_per)]
        acts = launches[-A:]
        mean = lambda v: sum(v) / max(1, len(v)) * 1000.0
        out[kind] = (mean([x for v in acts for x in v]), mean(acts[0]), mean(acts[-1]), e0.elapsed_time(e1) / turns * 1000.0)
    assert eng.status() == 0
    print(f"config 3's shape, {E} envs: us per sgw_act launch (engine event pairs), first / last agent, and the whole turn (HIP events)")
    for k, (m, f, l, t) in out.items():
        print(f"  {k:20s} act {m:7.2f} us   (agent 0 {f:6.2f}, agent 7 {l:6.2f})   turn {t:8.1f} us", flush=True)
    del eng
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for E in [int(v) for v in sys.argv[1:]] or [65536, 16384, 1024]:
        run(E)

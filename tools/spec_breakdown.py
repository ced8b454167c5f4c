#!/usr/bin/env python3
"""Where a speculative policy turn spends its time (config 5's shape by default): every segment of Environment._take_turn_speculative
between two synchronisations, mean over the turns.  GPU only.  usage: tools/spec_breakdown.py [h w agents radius envs]"""
import os, sys, time
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sorrel_amd.buffers import Buffer
from sorrel_amd.examples.treasurehunt.entities import EmptyEntity
from sorrel_amd.examples.treasurehunt.env import TreasurehuntEnv
from sorrel_amd.examples.treasurehunt.main import make_config
from sorrel_amd.examples.treasurehunt.world import TreasurehuntWorld
from sorrel_amd.models import BaseModel

h, w, A, r, E = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (128, 128, 64, 5, 2048)
one = [None]


class Shared(BaseModel):
    def __init__(self, input_size, action_space):
        super().__init__(input_size, action_space, memory_size=0, num_envs=E, device="cuda:0")
        self.memory = Buffer(capacity=4 * A, obs_shape=tuple(input_size), num_envs=E, device="cuda:0")
        self.w = torch.randn(int(input_size[0]), action_space, generator=torch.Generator().manual_seed(1)).cuda()

    def take_action(self, state):
        return (state.reshape(state.shape[0], -1) @ self.w).argmax(dim=1)


def factory(i, a):
    if one[0] is None:
        one[0] = Shared(i, a)
    return one[0]


cfg = make_config(h, w, A, r, spawn_prob=0.05 if h > 64 else 0.005)
env = TreasurehuntEnv(TreasurehuntWorld(cfg, EmptyEntity(), num_envs=E, device="cuda:0", seed=0), cfg, model_factory=factory)
env.speculate_turns = "always"
for _ in range(10):
    env.take_turn()
eng = env._engine
model = one[0]
acc, cnt = defaultdict(float), defaultdict(int)
mark = [0.0]


def seg(name):
    torch.cuda.synchronize()
    now = time.perf_counter()
    acc[name] += now - mark[0]
    cnt[name] += 1
    mark[0] = time.perf_counter()


T = int(os.environ.get("SPEC_TURNS", "100"))
dirty_counts = defaultdict(list)
for turn in range(T):
    env.turn += 1
    eng.epoch, eng.turn = env.epoch, env.turn
    mem = model.memory
    own = mem.states[mem.idx:mem.idx + A].view(A, E, -1)
    rows = eng.speculation_rows(own)
    flat = rows.view(A * E, -1)
    torch.cuda.synchronize()
    mark[0] = time.perf_counter()
    eng.step(eng.actions, sweep=True, agent_begin=0, agent_end=0, write_obs=False, turn=env.turn)
    seg("1 sweep")
    eng.speculation_windows(own)
    seg("2 pre-move windows of every agent")
    fresh = model.take_action(flat)
    seg("3 forward pass, all rows")
    rr, ar = mem.rewards[mem.idx:mem.idx + A], mem.actions[mem.idx:mem.idx + A]
    k = 1
    while True:
        eng.turn_resolve(k, own, fresh.contiguous(), rr, ar)
        seg(f"5.{k} resolve (writes the actions first)")
        lst = eng.spec_dirty(k)
        n = int(lst.numel())
        seg(f"6.{k} the host reads the count")
        dirty_counts[k].append(n)
        if n == 0:
            break
        k += 1
        m = 64 if n <= 64 else (1 << (n - 1).bit_length() if n <= 4096 else -(-n // 4096) * 4096)
        pad = eng._spec_list[(k - 1) & 1, :m]
        pad[n:m] = 0
        x = eng.gather_rows(flat, pad)
        seg(f"8.{k - 1} gather of the dirty rows (padded)")
        fresh = model.take_action(x)[:n]
        seg(f"9.{k - 1} forward pass, dirty rows")
    mem.idx = (mem.idx + A) % mem.capacity
    mem.size = min(mem.size + A, mem.capacity)
    seg("b ring bookkeeping (nothing to copy)")
total = 0.0
for name in sorted(acc):
    us = acc[name] / T * 1e6
    total += us
    print(f"{name:44s} {us:8.1f} us per turn   ({cnt[name] / T:.2f} per turn)")
print(f"{'sum (with a synchronisation per segment)':44s} {total:8.1f} us")
for k in sorted(dirty_counts):
    v = dirty_counts[k]
    print(f"dirty (env, agent) pairs after resolve {k}: mean {sum(v) / len(v):9.1f} of {E * A}  ({sum(v) / len(v) / (E * A) * 100:.2f} %)   turns that got there: {len(v)}")

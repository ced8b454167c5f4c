#!/usr/bin/env python3
"""Probe: how much would overlapping the store-bound and the compute-bound phases of config 5 buy?  Two engines of 1 024
envs each stepped on two HIP streams (their kernels run concurrently and drift out of phase) against one engine of 2 048
envs on one stream.  A diagnostic, not a bench line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

spec = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=0, dense_prob=0.25)
K = 300
one = GridEngine(spec, 2048, device="cuda:0"); one.reset(0)
for _ in range(200): one.step(random_actions=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(K): one.step(random_actions=True)
b.record(); torch.cuda.synchronize()
print(f"one stream, 2048 envs: {a.elapsed_time(b) / K * 1000:.1f} us per batch step")
del one
for parts in (2, 4):
    n = 2048 // parts
    engs = [GridEngine(spec, n, device="cuda:0", first_env_id=i * n) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for e in engs: e.reset(0)
    torch.cuda.synchronize()
    def run(k):
        for _ in range(k):
            for e, s in zip(engs, streams):
                with torch.cuda.stream(s):
                    e.step(random_actions=True)
    run(200)
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    print(f"{parts} streams x {n} envs: {(time.perf_counter() - t) / K * 1e6:.1f} us per 2048-env step (wall)")
    del engs

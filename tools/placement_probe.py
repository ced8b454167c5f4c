#!/usr/bin/env python3
"""Does WHERE the tensors live change a launch's duration?  (r05: two engines of the same config in one process ran the same kernel
83.0 and 90.7 us apart, steadily.)  Config 5's shape (or `--c3`: config 3), one process:
  A. six engines created one after another, timed round-robin: does the creation order matter?
  B. ONE engine whose observation tensor is re-pointed into one big allocation at different offsets;
  C. ... and into freshly hipMalloc'ed buffers of its own (bypassing torch's caching allocator: torch.cuda.memory.CUDAPluggableAllocator
     is not needed -- a new segment per tensor is forced by sizes above the allocator's split limit).
GPU only.  usage: tools/placement_probe.py [--c3] [envs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from _warm import timed_us
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

c3 = "--c3" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
E = int(args[0]) if args else (65536 if c3 else 2048)
spec = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=3) if c3 else treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=3, dense_prob=0.25)


def t(eng, n=200):
    return timed_us(lambda: eng.step(random_actions=True), n, ms=60.0)


def addr(x):
    return f"{x.data_ptr():#x}"


print(torch.cuda.get_device_name(0), "E =", E, flush=True)
print("A. engines created one after another (obs / grid addresses), three rounds of timing")
engs = []
for i in range(6):
    e = GridEngine(spec, E, device="cuda:0")
    e.reset(0)
    for _ in range(20):
        e.step(random_actions=True)
    engs.append(e)
res = [[] for _ in engs]
for rnd in range(3):
    for i, e in enumerate(engs):
        res[i].append(t(e))
for i, e in enumerate(engs):
    print(f"  engine {i}: obs {addr(e.obs)} (mod 2 MiB = {e.obs.data_ptr() % (2 << 20):#x})  grid {addr(e.grid)}  "
          + "  ".join(f"{u:7.1f}" for u in res[i]) + " us", flush=True)

print("B. engine 0, observation tensor re-pointed into ONE 2 x-sized allocation at offsets")
e = engs[0]
n = e.obs.numel()
big = torch.zeros(2 * n + (8 << 20), dtype=torch.float32, device="cuda:0")
base_off = (-big.data_ptr()) % (2 << 20) // 4            # elements up to the next 2 MiB boundary
own = e.obs
for off_bytes in (0, 128, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 128 * 37, 3 << 20):
    o = base_off + off_bytes // 4
    e.obs = big[o:o + n].view(own.shape)
    print(f"  offset 2MiB-aligned + {off_bytes:8d} B: {t(e):7.1f} {t(e):7.1f} us   ({addr(e.obs)})", flush=True)
e.obs = own
print(f"  its own tensor again: {t(e):7.1f} us")

print("C. engine 0, observation tensor in fresh allocations (each its own segment), then the engines of A again")
keep = []
for i in range(5):
    x = torch.zeros(n + i * 524288, dtype=torch.float32, device="cuda:0")    # (different sizes: no reuse of a cached block)
    keep.append(x)
    e.obs = x[:n].view(own.shape)
    print(f"  fresh allocation {i}: {t(e):7.1f} {t(e):7.1f} us   ({addr(e.obs)})", flush=True)
e.obs = own
for i, g in enumerate(engs):
    print(f"  engine {i} again: {t(g):7.1f} us", flush=True)
print(torch.cuda.memory_summary(abbreviated=True)[:1500])

if "--coarse" in sys.argv:
    print("D. engine 0's observation tensor in allocations of power-of-two sizes (one buddy block each?), four of each")
    nbytes = n * 4
    for size in (nbytes, 1 << 29, 1 << 30, 1 << 31):
        if size < nbytes:
            continue
        row = []
        for i in range(4):
            x = torch.empty(size // 4, dtype=torch.float32, device="cuda:0")
            keep.append(x)
            e.obs = x[:n].view(own.shape)
            row.append(t(e))
        print(f"  allocation of {size >> 20:5d} MiB: " + "  ".join(f"{u:7.1f}" for u in row) + " us", flush=True)
    print("E. one 8 GiB allocation, the observation tensor at coarse offsets inside it")
    huge = torch.empty((8 << 30) // 4, dtype=torch.float32, device="cuda:0")
    for off_mb in (0, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 7168):
        o = off_mb * (1 << 20) // 4
        e.obs = huge[o:o + n].view(own.shape)
        print(f"  offset {off_mb:5d} MiB: {t(e):7.1f} us", flush=True)
    e.obs = own

"""Diagnostic: per-segment share of a step_fast wave's lifetime, slot relaunch gaps and per-SIMD occupancy.
Needs a -DSGW_STAMPS build of the library (stamps exist in the PREBUILT instances only: jit = 0; group = 64 keeps small worlds on the
wave-per-env kernel): SGW_LIB=<that .so> SGW_OPTIONS="jit=0;group=64" PYTHONPATH=. python tools/stamps.py [E [h w a r]]."""
import ctypes as C
import sys

import torch

from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
h, w, a, r = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (32, 32, 8, 3)
spec = treasurehunt_spec(h, w, a, r, spawn_prob=0.005, seed=0)
eng = GridEngine(spec, E, device="cuda:0")
eng.reset(0)
print(eng.launch_info())
lib = N.load()
import numpy as np

buf = np.zeros((65536, 8), np.uint64)
for _ in range(3000 if E <= 8192 else 10):
    eng.step(random_actions=True)
torch.cuda.synchronize()
lib.sgw_debug_stamps(buf.ctypes.data_as(C.c_void_p))
b = buf[:E, :6].astype(np.float64)
names = ["launch->loads arrived", "grid->LDS + sweep", "move inputs (action draw)", "agent loop (obs + moves)", "write-back issue", "store drain"]
tot = b.sum(axis=1)
for i, n in enumerate(names):
    print(f"{n:28s} mean {b[:, i].mean():9.0f}  median {np.median(b[:, i]):9.0f}  p90 {np.percentile(b[:, i], 90):9.0f} ticks  {100.0 * b[:, i].sum() / tot.sum():5.1f} %")
print(f"{'total':28s} mean {tot.mean():9.0f}  median {np.median(tot):9.0f} ticks per wave")
# ---- timeline: per SIMD, how long is a wave slot empty between two waves?
start = buf[:E, 6].astype(np.int64)
end = start + buf[:E, :6].sum(axis=1).astype(np.int64)
hw = buf[:E, 7]
wave_id = (hw & 0xF).astype(np.int64)
simd = ((hw >> 4) & 3).astype(np.int64)
cu = ((hw >> 8) & 0xF).astype(np.int64)
sh = ((hw >> 12) & 1).astype(np.int64)
se = ((hw >> 13) & 7).astype(np.int64)
xcc = ((hw >> 32) & 0xF).astype(np.int64)
print("distinct xcc", len(np.unique(xcc)), "se", len(np.unique(se)), "sh", len(np.unique(sh)), "cu", len(np.unique(cu)), "simd", len(np.unique(simd)), "wave ids", np.unique(wave_id))
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
slot = key * 16 + wave_id
print("distinct SIMDs used", len(np.unique(key)), "distinct slots", len(np.unique(slot)))
gaps, lives = [], []
for x in np.unique(xcc):
    m = xcc == x
    t0 = start[m].min()
    print(f"xcc {x}: waves {m.sum()}  first start 0  last end {(end[m].max() - t0)} ticks")
order = np.lexsort((start, slot))
ss, st, en = slot[order], start[order], end[order]
same = ss[1:] == ss[:-1]
gap = (st[1:] - en[:-1])[same]
if gap.size == 0:       # one round of waves (a small batch): no slot is reused, nothing more to say
    print("every wave ran in its own slot (one round): no relaunch gaps, no convoys to look for")
    sys.exit(0)
print(f"slot relaunch gap: mean {gap.mean():.0f}  median {np.median(gap):.0f}  p90 {np.percentile(gap, 90):.0f}  max {gap.max()} ticks  (wave life mean {np.mean(en - st):.0f})")
# per-SIMD occupancy over the kernel: sum(life) / (8 slots * span)
occ = []
for k in np.unique(key)[:2000]:
    m = key == k
    span = end[m].max() - start[m].min()
    occ.append((end[m] - start[m]).sum() / (8.0 * span))
print(f"per-SIMD occupancy over its own span: mean {np.mean(occ):.3f}  min {np.min(occ):.3f}")
per_simd = np.bincount(np.unique(key, return_inverse=True)[1])
print("waves per SIMD: min", per_simd.min(), "max", per_simd.max(), "mean", per_simd.mean())

# ---- phase mix over time on one XCD: do the waves run in convoys (all loading, then all storing)?
x = xcc == np.unique(xcc)[0]
t0 = start[x].min()
seg = np.cumsum(np.concatenate([np.zeros((x.sum(), 1)), buf[:E][x][:, :6].astype(np.float64)], axis=1), axis=1) + (start[x] - t0)[:, None]
T = seg[:, -1].max()
bins = np.arange(0, T, 100.0)   # 1 us
names2 = ["loadwait", "sweep", "action", "agents", "wb+drain"]
rows = []
for tb in bins:
    c = [int(((seg[:, i] <= tb) & (seg[:, i + 1] > tb)).sum()) for i in range(4)]
    c.append(int(((seg[:, 4] <= tb) & (seg[:, 6] > tb)).sum()))
    rows.append(c)
rows = np.array(rows)
print("time(us)   " + " ".join(f"{n:>9s}" for n in names2) + "   resident")
for tb, r in list(zip(bins, rows))[:: max(1, len(bins) // 120)]:
    print(f"{tb / 100:9.1f}  " + " ".join(f"{v:9d}" for v in r) + f"   {r.sum():7d}")

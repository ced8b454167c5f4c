import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
E = 2048
spec = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=0, dense_prob=0.25)
eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
lib = N.load()
for _ in range(10): eng.step(random_actions=True)
torch.cuda.synchronize()
buf = np.zeros((65536, 8), np.uint64)
lib.sgw_debug_stamps(buf.ctypes.data_as(C.c_void_p))
b = buf[:E, :5].astype(np.float64) / 100.0   # us
names = ["load + sweep", "phase M (moves)", "phase R (observations)", "write-back issue", "store drain (wave 0)"]
for i, n in enumerate(names):
    print(f"{n:26s} mean {b[:, i].mean():7.2f} us  median {np.median(b[:, i]):7.2f}  p90 {np.percentile(b[:, i], 90):7.2f}")
start = buf[:E, 6].astype(np.int64); start -= start.min()
end = start + (buf[:E, :5].sum(axis=1)).astype(np.int64)
print("workgroup life mean %.1f us; kernel span %.1f us" % (b.sum(axis=1).mean(), end.max() / 100.0))
order = np.argsort(start)
print("start times (us) of workgroups, every 128th:", (start[order][::128] / 100.0).round(1).tolist())
# phase mix over time
seg = np.cumsum(np.concatenate([np.zeros((E, 1)), buf[:E, :5].astype(np.float64)], axis=1), axis=1) + start[:, None]
for tb in np.arange(0, end.max(), 500.0):
    c = [int(((seg[:, i] <= tb) & (seg[:, i + 1] > tb)).sum()) for i in range(5)]
    print(f"t={tb/100:6.1f} us  load+sweep {c[0]:5d}  M {c[1]:5d}  R {c[2]:5d}  wb {c[3]:5d}  drain {c[4]:5d}")

import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spec = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=0, dense_prob=0.25)
eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
print(eng.launch_info())
lib = N.load()
for _ in range(800): eng.step(random_actions=True)
torch.cuda.synchronize()
buf = np.zeros((65536, 8), np.uint64)
lib.sgw_debug_stamps(buf.ctypes.data_as(C.c_void_p))
b = buf[:E, :5].astype(np.float64) / 100.0   # us
names = ["load + sweep", "phase M (moves)", "phase R (observations)", "write-back issue", "store drain (wave 0)"]
for i, n in enumerate(names):
    print(f"{n:26s} mean {b[:, i].mean():7.2f} us  median {np.median(b[:, i]):7.2f}  p90 {np.percentile(b[:, i], 90):7.2f}")
start = buf[:E, 6].astype(np.int64)
stamped = start != 0            # (the walking variant: only a workgroup's FIRST env carries a start stamp)
start = np.where(stamped, start - start[stamped].min(), 0)
end = start + (buf[:E, :5].sum(axis=1)).astype(np.int64)
print("workgroup life mean %.1f us; kernel span %.1f us" % (b.sum(axis=1).mean(), end.max() / 100.0))
order = np.argsort(start)
print("start times (us) of workgroups, every 128th:", (start[order][::128] / 100.0).round(1).tolist())
# phase mix over time
seg = np.cumsum(np.concatenate([np.zeros((E, 1)), buf[:E, :5].astype(np.float64)], axis=1), axis=1) + start[:, None]
for tb in np.arange(0, min(end.max(), 100000), 500.0):
    c = [int(((seg[:, i] <= tb) & (seg[:, i + 1] > tb)).sum()) for i in range(5)]
    print(f"t={tb/100:6.1f} us  load+sweep {c[0]:5d}  M {c[1]:5d}  R {c[2]:5d}  wb {c[3]:5d}  drain {c[4]:5d}")

# where and when each workgroup ran: per XCD (the walking variant: a workgroup's first env carries its hardware id and start time;
# its later envs blockIdx + k * gridDim follow on the same CU)
hw = buf[:E, 7]
xcc = ((hw >> 32) & 0xF).astype(np.int64)
se = ((hw >> 13) & 7).astype(np.int64)
cu = ((hw >> 8) & 0xF).astype(np.int64)
have = stamped
print("workgroups with a start stamp:", int(have.sum()), "of", E, "envs")
for x in np.unique(xcc[have]):
    m = have & (xcc == x)
    cus = len(np.unique((se[m] * 16 + cu[m])))
    print(f"xcc {x}: {int(m.sum()):4d} workgroups on {cus:3d} CUs  first start {start[m].min() / 100.0:6.1f} us  last start {start[m].max() / 100.0:6.1f}  last end {end[m].max() / 100.0:6.1f} us"
          f"  mean life of their first env {b[m].sum(axis=1).mean():6.1f} us")

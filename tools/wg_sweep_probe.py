import sys; sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from _warm import timed_us
from generic_tables_probe_worlds import move_world
E = 65536
for name, spec in (("32x32x1 C4 A8 r3", move_world(32, 32, 1, 4, 8, 3)), ("40x40x2 C12 A8 r2", move_world(40, 40, 2, 12, 8, 2))):
    for opts in ({}, {"burst": 2}, {"burst": 1}, {"pack3": 0}):
        with N.options(**opts):
            eng = GridEngine(spec, E, device="cuda:0")
        eng.reset(0)
        res = []
        for cap in (0, 8, 7, 6, 5, 4):
            eng.set_wg_per_cu(cap if cap else 0)
            for _ in range(100): eng.step(random_actions=True)
            res.append("%d:%.1f" % (cap, timed_us(lambda: eng.step(random_actions=True), 100)))
        print(name, opts, " ".join(res), eng.launch_info().split(" group")[0], flush=True)
        del eng; torch.cuda.empty_cache()

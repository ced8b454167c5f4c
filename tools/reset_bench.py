import sys, os
sys.path.insert(0, os.getcwd())
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
for (h, w, a, r, E, dp) in ((32, 32, 8, 3, 65536, 0.0), (128, 128, 64, 5, 2048, 0.25), (21, 21, 2, 2, 65536, 0.0)):
    eng = GridEngine(treasurehunt_spec(h, w, a, r, dense_prob=dp), E, device="cuda:0")
    for _ in range(5): eng.reset(1)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for k in range(50): eng.reset(k)
    t1.record(); torch.cuda.synchronize()
    print(f"reset {h}x{w} A{a} E={E}: {t0.elapsed_time(t1) / 50 * 1000:.1f} us")

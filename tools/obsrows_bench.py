import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
h, w, A, r, E = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 32, 8, 3, 65536)
spec = treasurehunt_spec(h, w, A, r, spawn_prob=0.005, seed=0)
eng = GridEngine(spec, E, device="cuda:0")
eng.reset(0)
for _ in range(30): eng.step(random_actions=True)
rows = eng.window_rows(None)
def t(fn, K=50):
    from _warm import warm
    warm(fn, 80.0, probe=5)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K * 1000
print(os.environ.get("SGW_ROWS_ABLATE", "-"), f"observe_rows {t(lambda: eng.observe_rows(rows)):7.1f} us   observe (step kernel) {t(lambda: eng.observe()):7.1f} us")

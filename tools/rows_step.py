"""sgw_observe_rows of one world N times (for a profiler: rocprofv3 --pmc ... -- python3 tools/rows_step.py c3 60).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sorrel_amd.engine import GridEngine
from world_step import WORLDS

name, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
E = int(os.environ.get("E", 65536))
spec = WORLDS[name]
eng = GridEngine(spec, E, device="cuda:0")
eng.reset(0)
for _ in range(20):
    eng.step(random_actions=True)
per_env = 1
for d in spec.obs_shape[1:]:
    per_env *= int(d)
dests = [torch.empty((E, per_env), device="cuda:0") for _ in range(spec.num_agents)]
rows = eng.window_rows(dests)
for _ in range(steps):
    eng.observe_rows(rows)
torch.cuda.synchronize()
print(name, "observe_rows", eng.launch_info().split("phase=")[-1].split(" big_stage")[0], "bytes/launch", E * spec.num_agents * per_env * 4)

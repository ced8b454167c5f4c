"""One world of tools/jit_probe.py's list stepped N times with on-device random actions (for a profiler: rocprofv3 --pmc ... -- python3 tools/world_step.py
"32x32x1 C4 A8 r3" 60).  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
from generic_tables_probe_worlds import move_world

WORLDS = {"c3": treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005), "32x32x1 C4 A8 r3": move_world(32, 32, 1, 4, 8, 3),
          "40x40x2 C12 A8 r2": move_world(40, 40, 2, 12, 8, 2), "32x32x2 C5 A8 r3": move_world(32, 32, 2, 5, 8, 3),
          "32x32x2 C8 A8 r3": move_world(32, 32, 2, 8, 8, 3),
          "th 21x21 A2 r2": treasurehunt_spec(21, 21, 2, 2, spawn_prob=0.005), "th 10x10 A2 r2": treasurehunt_spec(10, 10, 2, 2, spawn_prob=0.005),
          "th 16x16 A4 r2": treasurehunt_spec(16, 16, 4, 2, spawn_prob=0.005),
          "24x24x2 C8 A6 r4": move_world(24, 24, 2, 8, 6, 4), "th 30x26 A7 r5": treasurehunt_spec(30, 26, 7, 5, spawn_prob=0.005)}
if __name__ == "__main__":
    name, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
    E = int(os.environ.get("E", 65536))
    eng = GridEngine(WORLDS[name], E, device="cuda:0")
    eng.reset(0)
    for _ in range(steps):
        eng.step(random_actions=True)
    torch.cuda.synchronize()
    print(name, eng.launch_info().split(" group")[0], "bytes/launch", WORLDS[name].algorithmic_bytes_per_env_step() * E)

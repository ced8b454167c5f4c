"""After how long an idle does the chip ramp again?  Config 3 at 65 536 envs: N ms of sleep after a settled run, then the next 20 and the following 200 launches (HIP events).  GPU only."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
spec = treasurehunt_spec(32, 32, 8, 3, spawn_prob=0.005, seed=0)
eng = GridEngine(spec, 65536, device="cuda:0"); eng.reset(0)
for _ in range(1500): eng.step(random_actions=True)
torch.cuda.synchronize()
def region(K):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K): eng.step(random_actions=True)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K * 1000
for rep in range(2):
    for idle_ms in (0, 1, 3, 10, 30, 100):
        for _ in range(300): eng.step(random_actions=True)
        torch.cuda.synchronize()
        time.sleep(idle_ms / 1000)
        for _ in range(5): eng.step(random_actions=True)
        torch.cuda.synchronize()
        k20 = region(20)
        k200 = region(200)
        print(f"idle {idle_ms:4d} ms: next 20 launches {k20:.1f} us, the 200 after {k200:.1f} us", flush=True)

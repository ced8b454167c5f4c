"""Cleanup 21x31x3 (10 agents, 11x11 windows, 9 channels): the window pipeline alone (sgw_observe of every agent: grid
read, gather, staged bursts) next to the whole turn, per stage_agents option (SGW_OPTIONS, read by sorrel_amd._native).  GPU only."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, %r); os.chdir(%r)
sys.argv = ["x"]; os.environ["MISC_ONLY"] = "none"
import importlib.util
spec_ = importlib.util.spec_from_file_location("bm", "tools/bench_misc.py"); bm = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(bm)
from sorrel_amd.engine import GridEngine
E = 65536
spec = bm.cleanup_spec(21, 31, 10, 5)
eng = GridEngine(spec, E, device="cuda:0")
g = np.zeros((3, 21, 31), np.uint8)
g[:, 0, :] = g[:, -1, :] = 2; g[:, :, 0] = g[:, :, -1] = 2
g[0, 1:7, 1:-1] = 3; g[0, 14:20, 1:-1] = 5; g[0, 7:14, 1:-1] = 1
pos = np.array([[8 + (i // 5) * 2, 3 + (i %% 5) * 5] for i in range(10)], np.uint8)
for (y, x) in pos: g[1, y, x] = 11
eng.grid.copy_(torch.from_numpy(np.broadcast_to(g, (E,) + g.shape).copy()))
eng.agent_pos.copy_(torch.from_numpy(np.broadcast_to(pos, (E,) + pos.shape).copy()))
for _ in range(60): eng.step(random_actions=True)
def t(f, K=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(K): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / K * 1000
print("observe %%.1f us   turn %%.1f us   %%s" %% (t(lambda: eng.observe()), t(lambda: eng.step(random_actions=True)), eng.launch_info().split(" group")[0]))
''' % (ROOT, ROOT)
for sa in sys.argv[1:] or ["3"]:
    env = dict(os.environ)
    for kv in sa.split(","):
        if "=" in kv:
            k, v = kv.split("="); env[k] = v
        else:
            env["SGW_OPTIONS"] = "stage_agents=" + kv
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if "observe" in l]
    print(sa, line[-1] if line else out.stderr[-400:], flush=True)

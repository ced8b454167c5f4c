"""Worlds above 4 KiB (step_big): direct per-channel dword stores against LDS-staged, line-aligned 16-byte streaming stores
(the default where a compile-time instance exists; option big_stage=0: off), us per turn; every variant's tensors compared with the first one's.  GPU only."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch, hashlib
sys.path.insert(0, %r)
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
h, w, a, r, E = (int(v) for v in sys.argv[1:6])
spec = treasurehunt_spec(h, w, a, r, spawn_prob=0.002, seed=3)
eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
for _ in range(30): eng.step(random_actions=True)
torch.cuda.synchronize()
dig = hashlib.sha256(eng.obs.cpu().numpy().tobytes() + eng.grid.cpu().numpy().tobytes() + eng.total_reward.cpu().numpy().tobytes()).hexdigest()[:12]
for _ in range(100): eng.step(random_actions=True)
sys.path.insert(0, os.path.join(%r, "tools"))
from _warm import timed_us
us = timed_us(lambda: eng.step(random_actions=True), 100)
print("RESULT %%7.1f us  %%s  %%s" %% (us, dig, eng.launch_info().split(" threads")[0] + " " + " ".join(x for x in eng.launch_info().split() if x.startswith(("lds=", "big_stage=")))))
''' % (ROOT, ROOT)
shapes = [(128, 128, 64, 5, 2048), (128, 128, 64, 5, 4096), (128, 128, 64, 5, 8192), (48, 48, 8, 5, 16384), (72, 72, 16, 5, 8192)]
variants = [{}, {"SGW_OPTIONS": "big_stage=0"}]       # (the unpadded-rows variant, option big_pad, was retired in round 5)
if os.environ.get("PROBE_WALK"):   # where does the walking variant pay?
    shapes = [(128, 128, 64, 5, e) for e in (1024, 1280, 1536, 2048, 2560, 3072, 4096)]
    variants = [{}, {"SGW_OPTIONS": "big_walk=0"}, {"SGW_OPTIONS": "big_walk=0,big_stage=0"}]
for sh in shapes:
    for env in variants:
        out = subprocess.run([sys.executable, "-c", CODE, *map(str, sh)], env={**os.environ, **env}, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(sh, str(env).ljust(75), line[-1][7:] if line else out.stderr[-300:], flush=True)

"""Plain and Tag worlds between 4 and 8 KiB per env: the workgroup-per-env kernel (option fast_8k=0) against a wave per env
(fast_8k=1) per batch size -- the data behind the batch threshold in sgw_create.  GPU only."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools")); os.chdir(%r)
os.environ["MISC_ONLY"] = "none"
import bench_misc as bm
from _warm import timed_us
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
kind, h, w, a, r, E = sys.argv[1], *(int(v) for v in sys.argv[2:7])
spec = bm.tag_spec(h, w, a, r) if kind == "tag" else treasurehunt_spec(h, w, a, r, spawn_prob=0.003, seed=2)
eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
for _ in range(50): eng.step(random_actions=True)
us = timed_us(lambda: eng.step(random_actions=True), 60)
print("RESULT %%8.1f us  %%.2f  %%s" %% (us, spec.algorithmic_bytes_per_env_step() * E / us / 1e3 / 8000, eng.launch_info().split(" threads")[0]))
''' % (ROOT, ROOT, ROOT)
SHAPES = [("th", 48, 48, 8, 5), ("th", 64, 64, 16, 3), ("th", 50, 50, 8, 3), ("tag", 72, 72, 16, 4), ("tag", 90, 90, 12, 3)]
for sh in SHAPES:
    for E in (2048, 4096, 8192, 16384, 65536):
        row = []
        for v in ("0", "1"):
            out = subprocess.run([sys.executable, "-c", CODE, *map(str, sh), str(E)], env={**os.environ, "SGW_OPTIONS": "fast_8k=" + v}, capture_output=True, text=True)
            l = [x for x in out.stdout.splitlines() if x.startswith("RESULT")]
            row.append(l[-1][7:] if l else out.stderr[-200:])
        print(sh, E, "| workgroup per env:", row[0].split("  step_")[0], "| wave per env:", row[1], flush=True)

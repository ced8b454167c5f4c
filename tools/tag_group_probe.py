"""Tag worlds up to 4 KiB: wave-per-env (option group=64) against two envs per wave (the default rule), 65 536 envs.  GPU only."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r); os.chdir(%r)
os.environ["MISC_ONLY"] = "none"
import importlib.util
sp = importlib.util.spec_from_file_location("bm", "tools/bench_misc.py"); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
from sorrel_amd.engine import GridEngine
h, w, a, r = (int(v) for v in sys.argv[1:5])
spec = bm.tag_spec(h, w, a, r)
E = 65536
eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
for _ in range(200): eng.step(random_actions=True)
sys.path.insert(0, "tools")
from _warm import timed_us
us = timed_us(lambda: eng.step(random_actions=True), 100)
print("RESULT %%7.1f us  %%.2f  %%s" %% (us, spec.algorithmic_bytes_per_env_step() * E / us / 1e3 / 8000, eng.launch_info().split(" threads")[0]))
''' % (ROOT, ROOT)
for shape in ((11, 11, 5, 4), (24, 24, 6, 3), (28, 28, 6, 3), (30, 30, 6, 4), (32, 32, 8, 4), (40, 40, 8, 3), (48, 48, 10, 4), (64, 64, 8, 3)):
    for env in ({}, {"SGW_OPTIONS": "group=64"}, {"SGW_OPTIONS": "group=32"}):
        out = subprocess.run([sys.executable, "-c", CODE, *map(str, shape)], env={**os.environ, **env}, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(shape, str(env).ljust(22), line[-1][7:] if line else out.stderr[-300:], flush=True)

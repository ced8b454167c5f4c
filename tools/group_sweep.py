#!/usr/bin/env python3
"""Which kernel for which small world?  Times Treasurehunt-shaped worlds on the wave-per-env kernel and on the packed
generic kernel (option group = 16 / 32 / 64, passed as SGW_OPTIONS to the child processes) -- the data behind the dispatch rule in sgw_create.  Run on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
h, w, a, r, E = (int(v) for v in sys.argv[1:6])
spec = treasurehunt_spec(h, w, a, r)
eng = GridEngine(spec, E, device="cuda:0"); eng.reset(0)
for _ in range(200): eng.step(random_actions=True)
sys.path.insert(0, os.path.join(%r, "tools"))
from _warm import timed_us
us = timed_us(lambda: eng.step(random_actions=True), 200)
print("%%7.1f us  %%.3f  %%s" %% (us, spec.algorithmic_bytes_per_env_step() * E / us / 1e3 / 8000, eng.launch_info().split(" threads")[0]))
''' % (ROOT, ROOT)
shapes = [(10, 10, 2, 2), (16, 16, 4, 2), (21, 21, 2, 2), (21, 21, 8, 2), (24, 24, 4, 3), (28, 28, 8, 3), (32, 32, 2, 2), (32, 32, 8, 2), (32, 32, 4, 3), (20, 20, 4, 4), (32, 32, 8, 3), (30, 30, 8, 4), (32, 32, 16, 4)]
E = 65536
for h, w, a, r in shapes:
    for env in ({"SGW_OPTIONS": "group=64"}, {"SGW_OPTIONS": "group=16"}, {"SGW_OPTIONS": "group=32"}, {"SGW_OPTIONS": "group=64,force_generic=1"}):
        out = subprocess.run([sys.executable, "-c", CODE, str(h), str(w), str(a), str(r), str(E)], env={**os.environ, **env},
                             capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if " us " in l]
        print(f"{h}x{w} A{a} r{r} {str(env):52s} {line[-1] if line else out.stderr[-200:]}", flush=True)

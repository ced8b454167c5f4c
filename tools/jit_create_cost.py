"""What sgw_create costs with specialised instances: cold (hipRTC compiles), from the disk cache (a fresh process: load only), from
memory (a second engine of the same world in one process), and with jit = 0 (the prebuilt instances).  GPU only."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
from generic_tables_probe_worlds import move_world
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
N.load()
worlds = [("th 32x33 A8 r3", treasurehunt_spec(32, 33, 8, 3)), ("own 32x32x2 C5", move_world(32, 32, 2, 5, 8, 3)), ("own 24x24x3 C7 (packed)", move_world(24, 24, 3, 7, 4, 2)),
          ("th 100x100 A8 r5 (step_big)", treasurehunt_spec(100, 100, 8, 5))]
for name, spec in worlds:
    for rep in ("first", "again"):
        t0 = time.perf_counter(); eng = GridEngine(spec, 4096, device="cuda:0"); dt = (time.perf_counter() - t0) * 1e3
        s = N.jit_stats()
        print(f"  {name:30s} {rep:6s} sgw_create + tensors {dt:8.1f} ms   {eng.launch_info().split(' group')[0]}")
s = N.jit_stats()
print(f"  process totals: compiled {s['compiled']} ({s['compile_ms']:.0f} ms), loaded from disk {s['disk_hits']}, reused in memory {s['mem_hits']}, load {s['load_ms']:.1f} ms")
''' % (ROOT, os.path.join(ROOT, "tools"))

with tempfile.TemporaryDirectory() as cache:
    for label, opts in (("cold (empty disk cache: hipRTC compiles)", f"jit_cache_dir={cache}"), ("a fresh process, code objects on disk", f"jit_cache_dir={cache}"),
                        ("jit = 0 (prebuilt instances only)", "jit=0")):
        print(label, flush=True)
        out = subprocess.run([sys.executable, "-c", CODE], env={**os.environ, "SGW_OPTIONS": opts}, capture_output=True, text=True)
        print(out.stdout.rstrip() or out.stderr[-800:], flush=True)

"""A/B timing on the GPU box: python tools/ab.py [--rounds N] [--args "<bench args>"] name=ENV1=V1,ENV2=V2 ...
Each variant is a set of environment variables (e.g. SGW_LIB=..., SGW_NO_FUSED=1); variants are run
interleaved for N rounds; prints min / median ms_per_step of each."""
import json
import os
import subprocess
import sys

rounds, bargs, variants = 4, "", []
it = iter(sys.argv[1:])
for a in it:
    if a == "--rounds":
        rounds = int(next(it))
    elif a == "--args":
        bargs = next(it)
    else:
        name, _, envs = a.partition("=")
        variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-series", "--prewarm-steps", "300", "--steps", "300"] + bargs.split(),
                             env={**os.environ, **env}, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[name].append(json.loads(out)["roofline"]["kernel_ms"] * 1000)
for name, v in res.items():
    v2 = sorted(v)
    print(f"{name:12s} min {v2[0]:7.1f}  med {v2[len(v2) // 2]:7.1f}  max {v2[-1]:7.1f} us   {['%.1f' % x for x in v]}")

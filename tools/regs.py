#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of sgw.hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/regs.py [extra hipcc flags]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-I", os.path.join(ROOT, "include"),
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:], "-o", "/dev/null", os.path.join(ROOT, "sorrel_amd", "csrc", "sgw.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
pats = (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("spill_s", r"SGPRs Spill: (\d+)"), ("spill_v", r"VGPRs Spill: (\d+)"))
for line in out.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key, pat in pats:
        m = re.search(pat, line)
        if m:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':64s} vgpr sgpr scratch occ spillS spillV")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("(Params)", "").replace("void ", "")
    print(f"{n[:64]:64s} {r.get('vgpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('scratch', -1):7d} {r.get('occ', -1):3d} {r.get('spill_s', -1):6d} {r.get('spill_v', -1):6d}")
if not rows:
    print(out[-3000:])

#!/usr/bin/env python3
"""Registers / scratch / occupancy of single kernel instances WITHOUT rebuilding the library: the hipRTC translation unit
(`__graft_entry__.jit_source()`) + explicit instantiations, compiled with hipcc for gfx950 in a temporary directory.
usage: tools/quick_regs.py [-DNAME=VALUE ...] 'step_big<true, 2, 6, 5>' ...      (no device needed)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

flags = [a for a in sys.argv[1:] if a.startswith("-")]
ids = [a for a in sys.argv[1:] if not a.startswith("-")]
csrc = os.path.join(ROOT, "sorrel_amd", "csrc")
src = "#include <hip/hip_runtime.h>\n#include <stdint.h>\n#include <stddef.h>\n#include \"" + os.path.join(ROOT, "include", "sgw.h") + "\"\n"
src += "".join(f'#include "{os.path.join(csrc, n)}"\n' for n in G.JIT_PARTS + ("small_kernels.h", "resolve.h"))
for i, inst in enumerate(ids):
    rows = inst.startswith(("phase_rows", "observe_rows", "act_patch"))
    if inst.startswith("turn_resolve"):
        src += f"template __global__ void {inst}(const Params, const ResolveArgs);\n"
        continue
    src += f"template __global__ void {inst}(const Params{', const RowPtrs' if rows else ''});\n"
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "tu.hip")
    with open(path, "w") as fh:
        fh.write(src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Rpass-analysis=kernel-resource-usage",
           *flags, "-o", os.path.join(d, "tu.o"), path]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
pats = (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r"[^ ]SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
        ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("spill_s", r"SGPRs Spill: (\d+)"), ("spill_v", r"VGPRs Spill: (\d+)"),
        ("lds", r"LDS Size \[bytes/block\]: (\d+)"))
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
print(f"{'kernel':72s} vgpr sgpr scratch occ spillS spillV")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("(Params)", "").replace("void ", "")
    print(f"{n[:72]:72s} {r.get('vgpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('scratch', -1):7d} {r.get('occ', -1):3d} {r.get('spill_s', -1):6d} {r.get('spill_v', -1):6d}")
if not rows:
    print(out[-4000:])

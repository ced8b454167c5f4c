#!/usr/bin/env python3
"""Registers / scratch / LDS of SPECIALISED instances (the code objects the library compiles in-process, hipRTC): compiles -- or takes
from the cache -- the `kernel`, `kernel_plain`, `kernel_rollout`, `kernel_walk` of every plan in tools/plan_cases.py (or the template-ids
given on the command line) and reads the code objects' metadata.  No device needed.  Exit code 1 if any instance uses scratch."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sorrel_amd import _native as N

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
ids = sys.argv[1:]
if not ids:
    import plan_cases

    seen = []
    for name, plan in plan_cases.plans().items():
        if not plan["specialised"]:
            continue
        for key in ("kernel", "kernel_plain", "kernel_rollout", "kernel_walk", "kernel_phase", "kernel_observe_rows"):
            k = plan[key]
            if "<" in k and not k.startswith("phase_kernel") and k not in seen:
                seen.append(k)
    ids = seen
bad = 0
print(f"{'instance':82s} vgpr sgpr scratch spillS spillV  lds(static)")
for inst in ids:
    try:
        path = N.jit_compile(inst)
    except Exception as exc:
        print(f"{inst:82s} COMPILE FAILED: {str(exc)[:200]}")
        bad += 1
        continue
    lowered, code = N.jit_code_object(path)
    with tempfile.NamedTemporaryFile(suffix=".hsaco") as tmp:
        tmp.write(code)
        tmp.flush()
        notes = subprocess.run([READELF, "--notes", tmp.name], capture_output=True, text=True).stdout
    blocks = notes.split("- .agpr_count")
    meta = next((b for b in blocks if f".name:           {lowered}" in b), "")
    get = lambda key: int((re.search(rf"\.{key}:\s+(\d+)", meta) or [0, -1])[1])
    scratch = get("private_segment_fixed_size")
    print(f"{inst:82s} {get('vgpr_count'):4d} {get('sgpr_count'):4d} {scratch:7d} {get('sgpr_spill_count'):6d} {get('vgpr_spill_count'):6d}  {get('group_segment_fixed_size'):6d}")
    bad += scratch > 0
print(f"{len(ids)} instances, {bad} with scratch or failed; cache: {N.jit_stats()}")
sys.exit(1 if bad else 0)

#!/bin/bash
# Run ON THE GPU BOX (round 6): L2 (TCC) write-path counters of config 5's store patterns -- step_big's own (mode 0, streaming stores), a block per
# env front to back (mode 6), four / sixteen pieces per env (mode 8), the fill-shaped kernel (mode 5) -- and of torch's fill_ over the same 380 MB.
# One rocprofv3 --pmc pass per counter group and pattern (never combined with tracing beyond --kernel-trace).  Output: gpurun_out/c5_emit_pmc.txt
cd "$(dirname "$0")"
[ -x ./c5_emit ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o c5_emit c5_emit.hip || exit 1
HERE=$(pwd)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c5_pmc
mkdir -p $OUT
export C5_ITERS=20
cd /tmp && export TMPDIR=/tmp
G1="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum"
G2="TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
G3="TCC_WRITE_sum TCC_WRITEBACK_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
G4="TCC_CYCLE_sum TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum TCC_LATENCY_FIFO_FULL_sum"
run() {   # tag, command...
  tag=$1; shift
  i=0
  for grp in "$G1" "$G2" "$G3" "$G4"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag.g$i -- "$@" > $OUT/$tag.g$i.log 2>&1 || { echo "$tag group $i FAILED"; tail -3 $OUT/$tag.g$i.log; return 1; }
    echo "$tag g$i done" >> $REPO/gpurun_out/c5_emit_pmc.progress
  done
}
run m0_nt1  $HERE/c5_emit 0 2048 0 39936 1 0 || exit 1
run m6_nt0  $HERE/c5_emit 6 2048 0 0 0 0 || exit 1
run m8k4    $HERE/c5_emit 8 2048 4 0 0 0 || exit 1
run m8k16   $HERE/c5_emit 8 2048 16 0 0 0 || exit 1
run m5      $HERE/c5_emit 5 2048 0 0 0 0 || exit 1
run fill    python3 $REPO/tools/fill_trace.py || exit 1
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
tags = ["m0_nt1", "m6_nt0", "m8k4", "m8k16", "m5", "fill"]
table = {}
for tag in tags:
    acc = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(out, tag + ".g*", "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        # the last 20 dispatches of the kernel under test (the warm-up launches before them are the same kernel)
        by = defaultdict(list)
        for r in rows:
            name = r.get("Kernel_Name", "")
            if tag == "fill" and "fill" not in name.lower() and "Fill" not in name:
                continue
            by[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in by.items():
            acc[k] = v[-20:]
    table[tag] = {k: sum(v) / len(v) for k, v in acc.items() if v}
names = sorted({k for t in table.values() for k in t})
print("%-40s" % "counter (average per launch)" + "".join("%14s" % t for t in tags))
for n in names:
    print("%-40s" % n + "".join("%14.4g" % table[t].get(n, float("nan")) for t in tags))
PY

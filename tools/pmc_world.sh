#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --pmc passes (one per counter group) over tools/world_step.py for worlds that are not BASELINE configs; per-dispatch averages of the step kernel.
# usage: tools/pmc_world.sh "<world>" ["<world>" ...]
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_world
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"
G3="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES"
w=0
for world in "$@"; do
  w=$((w+1))
  python3 $REPO/tools/${PMC_SCRIPT:-world_step.py} "$world" 200 > /dev/null 2>&1   # (fills the specialiser's cache: nothing compiles under the profiler)
  i=0
  for grp in "$G1" "$G2"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/w${w}g$i -- python3 $REPO/tools/${PMC_SCRIPT:-world_step.py} "$world" 40 > $OUT/w${w}g$i.out 2> $OUT/w${w}g$i.err || { echo "pass failed: $world group $i" >&2; tail -3 $OUT/w${w}g$i.err >&2; exit 1; }
  done
  echo "== $world: $(cat $OUT/w${w}g1.out | tail -1)"
  python3 - $OUT/w${w}g1 $OUT/w${w}g2 <<'PY'
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            if os.environ.get("PMC_KERNEL", "step_") in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 2:]          # the settled half
            print("  %-28s n=%d avg=%.6g" % (k, len(v), sum(v) / len(v)))
PY
  rm -rf $OUT/w${w}g1 $OUT/w${w}g2
done

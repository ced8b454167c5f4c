#!/bin/bash
# Run ON THE GPU BOX: instruction counters of the compile-time-shape and the run-time-shape wave-per-env kernels on
# near-identical worlds (tools/rt_shape_probe.py), per-dispatch averages.  usage: tools/pmc_rt_shapes.sh
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_rt_shapes
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH"; do
  i=$((i+1))
  RT_SHAPES=${RT_SHAPES:-1} timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/tools/rt_shape_probe.py > $OUT/g$i.out 2> $OUT/g$i.err || { echo "group $i failed" >&2; tail -5 $OUT/g$i.err >&2; break; }
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "step_" in row.get("Kernel_Name", ""):
            acc[(row["Kernel_Name"][32:90], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-60s %-24s n=%d avg=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY

#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes over tools/bench_misc.py (one shape selected with MISC_ONLY), per-dispatch averages
# of the step kernel.  usage: MISC_ONLY=cleanup CLEANUP_E=65536 tools/pmc_misc.sh <tag>
set -o pipefail
TAG=${1:-misc}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' || exit 1   # never compile under the profiler
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/tools/bench_misc.py > $OUT/g$i.out 2> $OUT/g$i.err
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "pmc_misc: group $i ($grp) FAILED rc=$rc; no further pass is started" >&2
    tail -5 $OUT/g$i.err >&2
    break
  fi
done
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "step_" in row.get("Kernel_Name", ""):
            acc[(row["Kernel_Name"][:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-72s %-28s n=%d avg=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY

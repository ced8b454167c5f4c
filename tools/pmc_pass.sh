#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass per counter group for the step kernel; prints per-dispatch averages.
# usage: tools/pmc_pass.sh <tag> "<bench args>" "<counters group 1>" ["<counters group 2>" ...]
set -o pipefail
TAG=$1; BARGS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' || exit 1   # never compile under the profiler
cd /tmp && export TMPDIR=/tmp
i=0
failed=0
for grp in "$@"; do
  i=$((i+1))
  # A group that does not finish is reported and fails the script (its counters are simply missing from the averages
  # otherwise).  Known on this pool: the TA_* groups never finish under rocprofv3 on gfx950 -- do not pass them.
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/bench.py --steps 30 --warmup 3 --prewarm-steps 0 --no-series --turns-per-launch 0 --no-cpu-baseline $BARGS > /dev/null 2> $OUT/g$i.err
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "pmc_pass: group $i ($grp) FAILED rc=$rc (124/137 = timed out); no further pass is started" >&2
    tail -5 $OUT/g$i.err >&2
    failed=1
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
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print("%-40s n=%d avg=%.6g" % (k, len(v), sum(v) / len(v)))
PY
exit $failed

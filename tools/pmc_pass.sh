#!/bin/bash
# Run ON THE GPU BOX: one rocprofv3 --pmc pass per counter group for the step kernel; prints per-dispatch averages.
# usage: tools/pmc_pass.sh <tag> "<bench args>" "<counters group 1>" ["<counters group 2>" ...]
set -o pipefail
TAG=$1; BARGS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/bench.py --steps 30 --warmup 3 --no-cpu-baseline $BARGS > /dev/null 2> $OUT/g$i.err || { tail -5 $OUT/g$i.err; }
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

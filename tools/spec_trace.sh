#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a few speculative policy turns (config 5's shape): every kernel of a turn with its duration.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_spec
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPEC_TURNS=20 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/spec_breakdown.py > $OUT/run.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last three turns: from a step_big (the sweep) to the next
starts = [i for i, r in enumerate(rows) if "step_big" in r["Kernel_Name"]]
lo = starts[-4]
for r in rows[lo:starts[-1]]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {name}")
PY
rm -rf $OUT

#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a few speculative policy turns (config 5's shape): every kernel of a turn with its duration.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_spec
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LAT_TURNS=60 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/latency_bench.py one 128 128 64 5 2048 6 > $OUT/run.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last three turns: from a step_big (the sweep) to the next
starts = [i for i, r in enumerate(rows) if "step_big" in r["Kernel_Name"]]
lo = starts[-4]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for i, r in enumerate(rows[lo:starts[-1]]):
    if lo + i in starts:
        t0 = int(r["Start_Timestamp"])
        print("---- turn")
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"at {(b - t0) / 1e3:8.1f}  gap {(b - prev_end) / 1e3:6.1f}  runs {(e - b) / 1e3:7.1f} us  {name}")
    prev_end = e
PY
rm -rf $OUT

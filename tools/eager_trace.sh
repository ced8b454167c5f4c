#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the eager policy turn at the headline batch (32x32, 8 agents, 65 536 envs, a linear policy per agent):
# device time per turn by kernel, and the idle time between kernels -- is the turn host-bound or device-bound?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_eager
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LAT_TURNS=80 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/latency_bench.py one 32 32 8 3 ${1:-65536} 1 > $OUT/run.txt 2>&1
grep "us/turn" $OUT/run.txt
python3 - $OUT <<'PY'
import csv, glob, sys, os, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a turn starts at the sweep-only step launch that is followed by observe_rows (or at step_fast_rows, which is both); the last 40 turns
fused = any("step_fast_rows" in r["Kernel_Name"] for r in rows)
starts = [i + (1 if fused else 0) for i, r in enumerate(rows) if ("step_fast_rows" if fused else "observe_rows") in r["Kernel_Name"]]
lo, hi = starts[-41], starts[-1]
turns = 40
busy = collections.Counter(); count = collections.Counter()
idle = 0
prev_end = int(rows[lo - 1]["End_Timestamp"])
for r in rows[lo - 1:hi - 1]:
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    busy[name] += e - b; count[name] += 1
    idle += max(0, b - prev_end); prev_end = max(prev_end, e)
span = int(rows[hi - 2]["End_Timestamp"]) - int(rows[lo - 1]["Start_Timestamp"])
print(f"per turn over the last {turns} turns (under the tracer): span {span / turns / 1e3:.1f} us, device busy {sum(busy.values()) / turns / 1e3:.1f} us, idle between kernels {idle / turns / 1e3:.1f} us")
for name, ns in busy.most_common():
    print(f"  {ns / turns / 1e3:8.1f} us  {count[name] / turns:5.1f} launches  {name}")
PY
rm -rf $OUT

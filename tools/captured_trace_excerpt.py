"""The last replay of a recorded policy turn out of a rocprofv3 kernel trace (csv): start and duration of every node.
usage: python tools/captured_trace_excerpt.py <..._kernel_trace.csv> [nodes per replay]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 27
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
print("#   start(us)  duration(us)  kernel")
for r in last:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:10.1f}   {r['Kernel_Name'][:120]}")

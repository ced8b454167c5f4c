#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats; the headline workload alone) ==")
for f in find("trace", "*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("  %-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"),
                                                                      row.get("TotalDurationNs"), row.get("Percentage")))
for f in find("trace_full", "*kernel_stats.csv"):
    print("== kernel stats of the FULL line (side configs + policy-turn leg; bench.py --steps 100) ==")
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("  %-60s calls=%s avg_ns=%s total_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"),
                                                                      row.get("TotalDurationNs"), row.get("Percentage")))
for f in find("trace", "*kernel_trace.csv"):
    durs = defaultdict(list)
    meta = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            n = row["Kernel_Name"]
            durs[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            meta[n] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                       row.get("Grid_Size"), row.get("Workgroup_Size"))
    print("== per-kernel durations from the trace ==")
    for n, d in durs.items():
        d2 = sorted(d)
        print("  %-60s n=%d avg=%.1fus med=%.1fus min=%.1fus vgpr/agpr/sgpr/lds/grid/wg=%s" % (n[:60], len(d), sum(d) / len(d) / 1e3,
                                                                                 d2[len(d2) // 2] / 1e3, d2[0] / 1e3, meta[n]))
print("== PMC (per dispatch averages for step_kernel) ==")
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "step_" in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print("  %-24s n=%d avg=%.6g" % (k, len(v), sum(v) / len(v)))

#!/usr/bin/env python3
"""Summarise a tools/collect_pmc.sh collection: per kernel, the kernel-stats row and the median per-launch value of
every counter.    python tools/summarize_pmc.py gpurun_out/<tag> [kernel substring ...] > profiles/<name>.txt"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
subs = sys.argv[2:] or [""]


def short(k):
    return k.split("(")[0][:70]


print("# collection:", root)
for p in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)):
    print("## rocprofv3 --kernel-trace --stats:", os.path.relpath(p, root))
    for r in csv.DictReader(open(p)):
        if any(s in r["Name"] for s in subs):
            print("  %-70s calls %4s  avg %12.1f ns  min %12s  max %12s  total %5s%%" % (
                short(r["Name"]), r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r.get("Percentage", "")))
vals = defaultdict(lambda: defaultdict(list))
for p in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(p)):
        if any(s in r["Kernel_Name"] for s in subs):
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(vals):
    print("## PMC (median per launch):", k)
    for c in sorted(vals[k]):
        v = sorted(vals[k][c])
        print("  %-44s %18.1f   (%d launches, min %.1f max %.1f)" % (c, v[len(v) // 2], len(v), v[0], v[-1]))

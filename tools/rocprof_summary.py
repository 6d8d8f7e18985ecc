#!/usr/bin/env python3
"""Summarises the rocprofv3 runs of tools/gpu/profile_r02.sh: per-kernel duration statistics, FETCH_SIZE (x2: on gfx950 the
counter reports half the bytes of a wide coalesced stream, MI355X_MICROARCH.md HBM section) and SQ counters per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
csv.field_size_limit(1 << 30)


def short(n):
    return n.replace("void ", "").split("(")[0][:110]


for tag in sorted(t for t in os.listdir(root) if t.startswith("trace_") and os.path.isdir(os.path.join(root, t))):
    fs = glob.glob(os.path.join(root, tag, "**", "*kernel_stats.csv"), recursive=True)
    print(f"== {tag}: rocprofv3 --kernel-trace --stats (durations in us)")
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "tmac::" in r["Name"] and "retile" not in r["Name"]:
                print(f"  {short(r['Name']):110s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:10.2f}  min {float(r['MinNs']) / 1e3:10.2f}  max {float(r['MaxNs']) / 1e3:10.2f}  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")

for tag in sorted(t for t in os.listdir(root) if t.startswith(("fetch_", "sq1_", "sq2_", "lds_", "sq_")) and os.path.isdir(os.path.join(root, t))):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "tmac::" in r["Kernel_Name"] and "retile" not in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {tag}: rocprofv3 --kernel-trace --pmc (mean per launch)")
    for k, cs in acc.items():
        n = max(len(v) for v in cs.values())
        parts = []
        for c, v in sorted(cs.items()):
            m = sum(v) / len(v)
            if c == "FETCH_SIZE":
                parts.append(f"FETCH_SIZE {m:.1f} KiB -> x2 = {m * 2 * 1024 / 1e6:.2f} MB ({int(m * 2 * 1024)} B)")
            else:
                parts.append(f"{c} {m:.4g}")
        print(f"  {k:110s} launches {n:5d}  " + "  ".join(parts))

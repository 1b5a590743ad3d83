#!/usr/bin/env python3
"""min / median / average duration per tip:: kernel from a rocprofv3 kernel_trace.csv, first fifth of the calls dropped (warm-up,
clock ramp, first-use code-object loads).  usage: python tools/kstats_table.py <kernel_trace.csv>"""
import collections
import csv
import statistics
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "tip::" not in k:
        continue
    rows[k.split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':70s} {'calls':>6s} {'min_us':>9s} {'median_us':>9s} {'avg_us':>9s}")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v = v[len(v) // 5:]
    print(f"{k[:70]:70s} {len(v):6d} {min(v):9.2f} {statistics.median(v):9.2f} {sum(v) / len(v):9.2f}")

#!/usr/bin/env python3
"""Print the kernel timeline of one forward from a rocprofv3 --kernel-trace CSV (start, end, duration, gap)."""
import csv
import sys

path, first = sys.argv[1], sys.argv[2]           # trace csv, substring of the first kernel of a forward
t = [r for r in csv.DictReader(open(path)) if "tip::" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]]
t.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(t) if first in r["Kernel_Name"]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(t[a]["Start_Timestamp"])
prev = None
for r in t[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = (s - prev) / 1000 if prev is not None else 0.0
    print(f"{s/1000:8.2f} {e/1000:8.2f} dur {(e-s)/1000:7.2f} gap {gap:6.2f}  {r['Kernel_Name'][:60]}")
    prev = e
print(f"forward-to-forward period {(int(t[b]['Start_Timestamp']) - t0)/1000:.2f} us")

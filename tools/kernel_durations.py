#!/usr/bin/env python3
"""Per-launch durations (us) of the kernels whose name contains a pattern, in launch order, from a rocprofv3
kernel-trace CSV:  python tools/kernel_durations.py trace.csv pattern [last_n]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if sys.argv[2] in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                         r["Kernel_Name"][:60], r.get("Grid_Size_X", "") or r.get("Grid_Size", "")))
rows.sort()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
for t, d, name, grid in rows[-n:]:
    print(f"{d:8.1f} us  grid {grid:>8s}  {name}")

#!/usr/bin/env python3
"""Workgroup dispatch pressure per kernel from a rocprofv3 --kernel-trace CSV: for the last training step (cut at the
adam_kernel launches) lists launches, workgroups and nanoseconds per workgroup.  A launch is bound by the dispatch rate
(about 6 ns per workgroup on MI355X, tools/reduce_breakdown.py) rather than by its work when that figure nears 6."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        gx, gy, gz = int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
        wx, wy, wz = int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"])
        wgs = -(-gx // wx) * -(-gy // wy) * -(-gz // wz)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], wgs, wx * wy * wz))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
step = rows[ends[-2] + 1:ends[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0, 0, 0])
for _, dur, name, wgs, wsz in step:
    key = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    a = agg[key]
    a[0] += 1; a[1] += dur; a[2] += wgs; a[3] = wsz
print(f"{'kernel':70s} {'n':>4s} {'us':>8s} {'wgs':>8s} {'thr':>4s} {'ns/wg':>7s}")
for key, (n, dur, wgs, wsz) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 50]:
    print(f"{key:70s} {n:4d} {dur / 1e3:8.1f} {wgs:8d} {wsz:4d} {dur / wgs:7.1f}")

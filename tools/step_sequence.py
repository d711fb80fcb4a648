#!/usr/bin/env python3
"""The kernels of ONE training step in launch order, from a rocprofv3 --kernel-trace CSV (cut at the fused Adam launch as
tools/trace_steps.py does): index, start offset (us), duration (us), gap to the previous kernel's end (us), grid, name.
Usage: tools/step_sequence.py <kernel_trace.csv> [--step -2] > sequence.txt"""
import argparse
import csv
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from trace_steps import short  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--step", type=int, default=-2)
args = ap.parse_args()
rows = []
with open(args.trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
first, last = ends[args.step - 1] + 1, ends[args.step]
sel = rows[first:last + 1]
t0 = sel[0][0]
prev_end = t0
gaps = 0.0
for i, (s, e, n, g, wg) in enumerate(sel):
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0.0)
    print(f"{i:4d} {((s - t0) / 1e3):9.1f} {((e - s) / 1e3):8.1f} {gap:7.1f} {g:>10} {wg:>5}  {short(n)}")
    prev_end = max(prev_end, e)
print(f"# {len(sel)} kernels, wall {(sel[-1][1] - t0) / 1e3:.1f} us, sum of gaps {gaps:.1f} us")

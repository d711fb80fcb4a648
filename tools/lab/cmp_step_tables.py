#!/usr/bin/env python3
"""Per-kernel difference of two step tables (tools/trace_steps.py --md):  python tools/lab/cmp_step_tables.py A.md B.md [N]"""
import re
import sys


def load(p):
    d = {}
    for line in open(p):
        m = re.match(r'\| `(.+)` \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|', line)
        if m:
            d[m.group(1)] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = sorted((b.get(k, (0, 0, 0))[1] - a.get(k, (0, 0, 0))[1], k) for k in set(a) | set(b))
print("delta ms/step (B - A) | kernel | A launches, ms, avg us | B")
for d, k in rows[:n] + [(None, "...")] + rows[-n:]:
    if d is None:
        print("...")
        continue
    x, y = a.get(k, (0, 0, 0)), b.get(k, (0, 0, 0))
    print(f"{d:+.3f} {k[:60]:60s} {x[0]:4.0f} {x[1]:.3f} {x[2]:5.1f} | {y[0]:4.0f} {y[1]:.3f} {y[2]:5.1f}")
print(f"total {sum(r[0] for r in rows):+.3f}")

#!/usr/bin/env python3
"""Per-training-step kernel statistics from a rocprofv3 --kernel-trace CSV.

The fused Adam launch (`adam_kernel`) ends every training step, so the trace is cut at those launches and only the
LAST `--steps` steps (bench.py's timed region; warm-up, MIOpen find and hipGraph capture runs are dropped) are
aggregated.  Usage: tools/trace_steps.py <kernel_trace.csv> --steps 5 [--md out.md] [--title "..."]
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]{0,60})?", name)
    base = m.group(1) if m else name
    if base.startswith("_ZN2ck"):
        k = re.search(r"kernel_[a-z_0-9]+", name)
        base = "ck::" + (k.group(0) if k else "kernel")
    if base.startswith("at::native::") and m and m.group(2):
        f = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|\w+_kernel_impl\w*|launch_\w+|batch_norm\w+)", name)
        base += "<" + (f.group(1) if f else "") + ">"
    if m and m.group(2) and not base.startswith(("at::", "ck::")):
        base += m.group(2)[:40]
    return base[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--md", default=None)
    ap.add_argument("--title", default="")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    if len(ends) < args.steps + 1:
        raise SystemExit(f"only {len(ends)} steps in the trace")
    first = ends[-args.steps - 1] + 1
    last = ends[-1]
    sel = rows[first:last + 1]
    wall = (sel[-1][1] - sel[0][0]) / 1e6
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
    busy = sum(v[1] for v in agg.values()) / 1e6
    lines = []
    lines.append(f"steps analysed: {args.steps}; launches/step: {len(sel) / args.steps:.0f}; "
                 f"GPU wall/step: {wall / args.steps:.3f} ms; sum of kernel durations/step: {busy / args.steps:.3f} ms")
    lines.append("")
    lines.append("| kernel | launches/step | ms/step | avg us | % of kernel time |")
    lines.append("|---|---|---|---|---|")
    for name, (cnt, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        lines.append(f"| `{name}` | {cnt / args.steps:.1f} | {ns / 1e6 / args.steps:.3f} | {ns / cnt / 1e3:.1f} | "
                     f"{100.0 * ns / (busy * 1e6):.2f} |")
    text = "\n".join(lines)
    print(text)
    if args.md:
        with open(args.md, "w") as f:
            f.write(f"# {args.title}\n\n{text}\n")


if __name__ == "__main__":
    main()

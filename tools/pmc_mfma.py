#!/usr/bin/env python3
"""Matrix-core counters of the training step from one rocprofv3 --pmc pass over bench.py
(SQ_INSTS_VALU_MFMA_MOPS_BF16 / _F32, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE):
per kernel and for the whole step -- MFMA FLOPs (MOPS x 512, the gfx9 MfmaFlops formula), the share of SIMD time the
matrix pipe was busy (MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)), and FLOP/s against the
2.5 PFLOP/s dense bf16 peak.  Steps are cut at the fused Adam launch; the last --steps steps are used.

    tools/pmc_mfma.py <run_counter_collection.csv> --steps 3 --md profiles/r02_pmc_mfma.md
"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--md", default=None)
ap.add_argument("--simds", type=int, default=1024)
args = ap.parse_args()

disp = collections.OrderedDict()           # dispatch id -> dict(name, counters, start, end)
with open(args.csv) as f:
    for row in csv.DictReader(f):
        d = disp.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"], "c": {}, "t0": int(row["Start_Timestamp"]),
                                                       "t1": int(row["End_Timestamp"])})
        d["c"][row["Counter_Name"]] = d["c"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
order = sorted(disp)
ends = [i for i, k in enumerate(order) if "adam_kernel" in disp[k]["name"]]
if len(ends) <= args.steps:
    raise SystemExit(f"only {len(ends)} adam launches in the trace")
first, last = ends[-args.steps - 1] + 1, ends[-1]
sel = [disp[order[i]] for i in range(first, last + 1)]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:70]


agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, 0.0])      # launches, bf16 flops, f32 flops, mfma busy, gui active, ns
for d in sel:
    a = agg[short(d["name"])]
    c = d["c"]
    a[0] += 1
    a[1] += 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    a[2] += 512.0 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
    a[3] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a[4] += c.get("GRBM_GUI_ACTIVE", 0.0)
    a[5] += d["t1"] - d["t0"]
tot = [sum(v[i] for v in agg.values()) for i in range(6)]
n = args.steps
lines = [f"steps analysed: {n}; dispatches/step: {tot[0] / n:.0f}; MFMA FLOPs/step: bf16 {tot[1] / n / 1e9:.2f} G, f32 {tot[2] / n / 1e9:.2f} G; "
         f"sum of kernel time/step (profiled, serialised): {tot[5] / n / 1e6:.2f} ms",
         f"whole-step MfmaUtil (busy cycles / (GRBM_GUI_ACTIVE x {args.simds} SIMDs)): {100.0 * tot[3] / max(tot[4] * args.simds / 8, 1):.3f} % "
         f"(GRBM_GUI_ACTIVE is summed over the 8 XCDs: divided by 8)", "",
         "| kernel | launches/step | bf16 MFMA GFLOP/step | f32 MFMA GFLOP/step | MfmaUtil % | TFLOP/s while running |", "|---|---|---|---|---|---|"]
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    if v[1] + v[2] == 0:
        continue
    util = 100.0 * v[3] / max(v[4] * args.simds / 8, 1)
    lines.append(f"| `{k}` | {v[0] / n:.0f} | {v[1] / n / 1e9:.3f} | {v[2] / n / 1e9:.3f} | {util:.2f} | {(v[1] + v[2]) / max(v[5], 1) / 1e3:.1f} |")
text = "\n".join(lines)
print(text)
if args.md:
    with open(args.md, "w") as f:
        f.write(text + "\n")

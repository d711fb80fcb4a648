#!/bin/bash
# Per-kernel instruction counts and wave-cycle splits of the captured training step (bench.py, 3 timed steps): two rocprofv3
# --pmc passes with --kernel-trace; the table lists the kernels by their total vector-instruction count per step.
R=$PWD
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc/step$i -- python3 $R/bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-roofline --no-host-fed --sustained-seconds 0 > $R/gpurun_out/pmc/step$i.log 2>&1 || echo "group $i failed"
done
python3 - $R/gpurun_out/pmc <<'PY'
import csv, glob, sys, collections, re
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for i in (1, 2):
    fs = glob.glob(f"{root}/step{i}/*/*counter_collection.csv")
    if not fs:
        continue
    seen = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = re.sub(r"^void ", "", k)
        k = re.sub(r"\(.*", "", k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES"):
            seen[k] += 1
    if i == 1:
        cnt = seen
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))
tot = sum(v.get("SQ_INSTS_VALU", 0) for _, v in rows)
print(f"# whole run (warm-up + captures + 3 timed steps): {tot/1e6:.1f} M vector wave-instructions")
print("| kernel | dispatches | VALU M | VALU per wave | SALU per wave | LDS per wave | issuing | wait_inst | waitcnt |")
print("|---|---|---|---|---|---|---|---|---|")
for k, v in rows[:45]:
    w = max(v.get("SQ_WAVES", 0), 1)
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"| `{k[:70]}` | {cnt[k]} | {v.get('SQ_INSTS_VALU',0)/1e6:.2f} | {v.get('SQ_INSTS_VALU',0)/w:.0f} | {v.get('SQ_INSTS_SALU',0)/w:.0f} | "
          f"{v.get('SQ_INSTS_LDS',0)/w:.0f} | {v.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} | {v.get('SQ_WAIT_INST_ANY',0)/wc:.2f} | {v.get('SQ_WAIT_ANY',0)/wc:.2f} |")
PY
rm -rf $R/gpurun_out/pmc/step1 $R/gpurun_out/pmc/step2

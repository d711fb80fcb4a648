#!/bin/bash
# L2 hit rate per kernel of the captured training step with the image-to-XCD numbering on and off:
# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace over bench.py (3 timed steps), XPT_XCD_AFFINITY=1 / 0.
R=$PWD
mkdir -p gpurun_out/pmc
trap 'rm -rf $R/gpurun_out/pmc/xcd1 $R/gpurun_out/pmc/xcd0' EXIT
cd /tmp && export TMPDIR=/tmp
for a in 1 0; do
  export XPT_XCD_AFFINITY=$a
  timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc/xcd$a -- python3 $R/bench.py --steps 3 --warmup 4 --no-cpu-baseline --no-roofline --no-host-fed --sustained-seconds 0 > $R/gpurun_out/pmc/xcd$a.log 2>&1 || echo "pass $a failed"
done
unset XPT_XCD_AFFINITY
python3 - $R/gpurun_out/pmc <<'PY'
import csv, glob, sys, collections, re
root = sys.argv[1]
tab = {}
for a in (1, 0):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    fs = glob.glob(f"{root}/xcd{a}/*/*counter_collection.csv")
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        k = re.sub(r"^void ", "", k)
        k = re.sub(r"\(.*", "", k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    tab[a] = agg
print("# L2 (TCC) hit rate per kernel over the whole run (warm-up + captures + 3 timed steps): image-to-XCD numbering on / off")
print("| kernel | requests M (on) | hit rate on | hit rate off |")
print("|---|---|---|---|")
def rate(v):
    t = v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)
    return (v.get("TCC_HIT_sum", 0) / t if t else 0.0), t
rows = sorted(tab.get(1, {}).items(), key=lambda kv: -rate(kv[1])[1])
tot = {a: [sum(v.get("TCC_HIT_sum", 0) for v in tab.get(a, {}).values()), sum(v.get("TCC_MISS_sum", 0) for v in tab.get(a, {}).values())] for a in (1, 0)}
for k, v in rows[:40]:
    r1, t1 = rate(v)
    r0, _ = rate(tab.get(0, {}).get(k, {}))
    print(f"| `{k[:72]}` | {t1/1e6:.2f} | {r1:.3f} | {r0:.3f} |")
for a in (1, 0):
    h, m = tot[a]
    print(f"\nwhole run, numbering {'on' if a else 'off'}: {h/1e6:.1f} M hits, {m/1e6:.1f} M misses, hit rate {h/max(h+m,1):.3f}")
PY

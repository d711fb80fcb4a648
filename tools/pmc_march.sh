#!/bin/bash
# PMC passes over the march kernels (csrc/xpt_march.hip) with the standalone lab client (tools/lab/fused_lab.hip: no torch):
# one rocprofv3 --pmc <group> --kernel-trace pass per counter group, B=8 (the training step's shape) and B=128, plus
# calibration launches of known byte counts (FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950).
R=$PWD
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  for B in 8 128; do
    LAB_CALIB=1 timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc/g${i}_$B -- $R/tools/lab/bin/fused_lab $B 128 416 5 new > $R/gpurun_out/pmc/g${i}_$B.log 2>&1 || echo "group $i B=$B failed"
    f=$(ls $R/gpurun_out/pmc/g${i}_$B/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then
      python3 - "$f" $B <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    m = re.search(r"(march_fwd_ms_kernel|march_bwd_ms_kernel<\d>|march_bwd_ms_p_kernel<\d>|march_bwd_finish_kernel|calib_copy16|calib_read12)", k)
    if m:
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(f"B={sys.argv[2]}", key, {c: round(sum(v) / len(v)) for c, v in cs.items()}, flush=True)
PY
      rm -rf $R/gpurun_out/pmc/g${i}_$B
    fi
  done
done

#!/bin/bash
# PMC passes over the k x k weight-gradient kernels of one training step (tools/hot_replay.py replays every launch hot):
# LDS and MFMA activity per kernel instantiation.  One rocprofv3 --pmc group per run, --kernel-trace only.
R=$PWD
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc/w$i -- python3 $R/tools/hot_replay.py --repeats 2 > $R/gpurun_out/pmc/w$i.log 2>&1 || echo "group $i failed"
  f=$(ls $R/gpurun_out/pmc/w$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(conv_wgrad_kernel<\d>|conv_halo_kernel<\d, \w+>|dw_multi_bwd_vec_kernel<[^>]*>|conv1x1_wgrad_kernel<64, 64, 8, 8, true, 8>)", r["Kernel_Name"])
    if m:
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key, {c: (round(sum(v) / len(v)), len(v)) for c, v in cs.items()}, flush=True)
PY
    rm -rf $R/gpurun_out/pmc/w$i
  fi
done

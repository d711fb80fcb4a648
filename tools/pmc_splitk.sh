#!/bin/bash
# PMC passes over the split-K tile kernel (csrc/xpt_conv_splitk.hip) on dp_up4_conv1 (batch 8): one rocprofv3 --pmc <group>
# --kernel-trace pass per counter group over `python tools/bench_splitk.py 8 one`.
R=$PWD
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc/sk$i -- python3 $R/tools/${PMC_TOOL:-bench_splitk.py} 8 ${PMC_ARG:-one} > $R/gpurun_out/pmc/sk$i.log 2>&1 || echo "group $i failed"
  f=$(ls $R/gpurun_out/pmc/sk$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    m = re.search(r"(conv_splitk_kernel<\d>|conv_stream_kernel<\d>|conv_wgrad_kernel<\d>|conv_splitk_finish_kernel<\d+>|conv_lds_kernel<\d>|conv_igemm_kernel<[\d, ]+>|conv_halo_kernel<[\w, ]+>)", k)
    if m:
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))), flush=True)
PY
    rm -rf $R/gpurun_out/pmc/sk$i
  fi
done

#!/bin/bash
# PMC passes over the fused kernels (B=8 and B=128 probe); counters in small groups, kernel-trace only
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_TA_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  if [ -n "$PMC_GROUPS" ] && [ $i -gt $PMC_GROUPS ]; then break; fi
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc/g$i -- python tools/roofline_probe.py > gpurun_out/pmc/g$i.log 2>&1 || echo "group $i failed"
  f=$(ls gpurun_out/pmc/g$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "fused_fwd" in k or "fused_bwd_kernel" in k or "affine_act_fwd_kernel<float>" in k:
        import re
        m = re.search(r"(fused_fwd_ms_kernel|fused_bwd_ms_kernel<\d>|fused_fwd_kernel<[\w, ]+>|fused_bwd_kernel<\d>|affine_act_fwd_kernel<float>)", k)
        key = (m.group(1) if m else k[:40], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key, {c: round(sum(v) / len(v)) for c, v in cs.items()}, flush=True)
PY
    rm -rf gpurun_out/pmc/g$i/*/
  fi
done

#!/bin/bash
# round 4: the one-pass march launch under row-step variants / rows-per-chunk plans (tools/lab/fused_lab.hip, LAB_PLAN)
mkdir -p gpurun_out/lab
out=gpurun_out/lab/march_sweep.txt
: > $out
run() { echo "== LAB_PLAN=$1 $2 $3 $4" >> $out; LAB_PLAN=$1 timeout -k 10 120 ./tools/lab/bin/fused_lab $2 $3 $4 ${5:-30} ${6:-new} >> $out 2>&1 || exit 1; }
run 1,0,0,0,0 8 128 416 30 both &&
run 0,0,0,0,0 8 128 416 30 new &&
for plan in 1,16,16,8,8 1,16,16,16,16 1,16,32,16,16 1,32,32,16,16 1,16,8,8,8 1,8,8,8,8 0,16,16,8,8 0,16,16,16,16; do run $plan 8 128 416 30 new || exit 1; done &&
run 1,0,0,0,0 128 128 416 10 both &&
run 0,0,0,0,0 128 128 416 10 new &&
run 1,0,0,0,0 32 256 832 10 both &&
run 0,0,0,0,0 32 256 832 10 new &&
run 1,0,0,0,0 4 256 832 30 both &&
run 0,0,0,0,0 4 256 832 30 new
cat $out

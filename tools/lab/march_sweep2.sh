#!/bin/bash
mkdir -p gpurun_out/lab
out=gpurun_out/lab/march_sweep2.txt
: > $out
run() { echo "== LAB_PLAN=$1 $2 $3 $4" >> $out; LAB_PLAN=$1 timeout -k 10 120 ./tools/lab/bin/fused_lab $2 $3 $4 ${5:-30} ${6:-new} >> $out 2>&1 || exit 1; }
for plan in 1,0,0,0,0 2,0,0,0,0 1,0,0,0,0 2,0,0,0,0 1,16,16,8,8 1,16,16,16,16 1,13,16,16,8 1,12,8,8,8 1,11,11,8,8 2,13,16,16,8 1,8,8,8,8; do run $plan 8 128 416 30 new || exit 1; done
run 1,0,0,0,0 128 128 416 10 new
run 2,0,0,0,0 128 128 416 10 new
run 1,0,0,0,0 4 256 832 30 new
run 2,0,0,0,0 4 256 832 30 new
grep -v "^  " $out

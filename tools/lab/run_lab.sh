#!/bin/bash
# the standard lab sweep on the GPU box: in-step shape, beyond-cache batch, high resolution
mkdir -p gpurun_out/lab
{
timeout -k 10 120 ./tools/lab/bin/fused_lab 8 128 416 50 ${1:-both} &&
timeout -k 10 120 ./tools/lab/bin/fused_lab 128 128 416 10 ${1:-both} &&
timeout -k 10 120 ./tools/lab/bin/fused_lab 32 256 832 10 ${1:-both}
} > gpurun_out/lab/fused_lab.txt 2>&1
cat gpurun_out/lab/fused_lab.txt

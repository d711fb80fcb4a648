#!/bin/bash
# rocprofv3 kernel trace of the bench step: per-step kernel table + one step's kernels in launch order -> gpurun_out/prof/<tag>_*
TAG=${1:-r03}
R=$PWD
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline --no-host-fed --sustained-seconds 0 > $R/gpurun_out/prof/${TAG}_trace.log 2>&1
cd $R
python tools/trace_steps.py gpurun_out/prof/trace/*/*kernel_trace.csv --steps 10 --top 70 --md gpurun_out/prof/${TAG}_bench_step_kernels.md > gpurun_out/prof/${TAG}_trace_steps.log 2>&1
python tools/step_sequence.py gpurun_out/prof/trace/*/*kernel_trace.csv > gpurun_out/prof/${TAG}_step_sequence.txt 2>&1
cp gpurun_out/prof/trace/*/*kernel_stats.csv gpurun_out/prof/${TAG}_bench_kernel_stats_whole_run.csv
rm -rf gpurun_out/prof/trace
tail -2 gpurun_out/prof/${TAG}_trace.log; head -4 gpurun_out/prof/${TAG}_bench_step_kernels.md

R=$PWD
mkdir -p gpurun_out/final
# (the rocprofv3 output directories are hundreds of MB: always removed, whatever fails in between -- gpurun merges back at most 64 MiB)
trap 'rm -rf $R/gpurun_out/final/trace $R/gpurun_out/final/pmc' EXIT
timeout -k 10 500 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench_stderr.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/trace -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline --no-host-fed > $R/gpurun_out/final/trace.log 2>&1
cd $R
python tools/trace_steps.py gpurun_out/final/trace/*/*kernel_trace.csv --steps 10 --top 60 --md gpurun_out/final/step_kernels.md > gpurun_out/final/trace_steps.log 2>&1
cp gpurun_out/final/trace/*/*kernel_stats.csv gpurun_out/final/kernel_stats_whole_run.csv
rm -rf gpurun_out/final/trace
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-host-fed > $R/gpurun_out/final/pmc.log 2>&1
cd $R
python tools/pmc_mfma.py gpurun_out/final/pmc/*/*counter_collection.csv --steps 3 --md gpurun_out/final/pmc_mfma.md > gpurun_out/final/pmc_mfma.log 2>&1
rm -rf gpurun_out/final/pmc

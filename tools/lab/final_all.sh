mkdir -p gpurun_out/final
bash tools/final_measurements.sh > gpurun_out/final/final.log 2>&1
bash tools/config_table.sh > gpurun_out/final/config_table.txt 2>&1
timeout -k 10 500 python tools/hot_replay.py --roofline 30 > gpurun_out/final/hot_replay.txt 2>&1
timeout -k 10 300 python tools/sink_census.py 40 > gpurun_out/final/sink_census.txt 2>&1
bash tools/profile_step.sh r04_z > /dev/null 2>&1
cat gpurun_out/final/bench_line.json; cat gpurun_out/final/config_table.txt; head -5 gpurun_out/final/step_kernels.md

mkdir -p gpurun_out/lab
{
timeout -k 10 900 python -m pytest tests/test_graph_replay.py tests/test_total_loss_gpu.py tests/test_rccl_single_rank_gpu.py tests/test_inloop_metrics_gpu.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do echo "now: $(bash tools/quick_bench.sh --steps 40 --warmup 10)"; done
echo "distributed: $(bash tools/quick_bench.sh --steps 40 --warmup 10 --mode distributed)"
} > gpurun_out/lab/exp_pyr.txt 2>&1
cat gpurun_out/lab/exp_pyr.txt

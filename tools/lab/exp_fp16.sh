mkdir -p gpurun_out/lab
{
timeout -k 10 1500 python -m pytest tests/test_fp16_build_gpu.py -x -q -m gpu 2>&1 | tail -40
echo "bf16: $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "fp16: $(bash tools/quick_bench.sh --steps 40 --warmup 10 --dtype fp16)"
} > gpurun_out/lab/exp_fp16.txt 2>&1
tail -45 gpurun_out/lab/exp_fp16.txt

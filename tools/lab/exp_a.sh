mkdir -p gpurun_out/lab
{
timeout -k 10 900 python -m pytest tests/test_encoder_parity_gpu.py tests/test_ref_nasnet.py tests/test_graph_replay.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
echo "off: $(XPT_DEBUG_STEM_RELU=0 bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "on : $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
done
} > gpurun_out/lab/exp_a.txt 2>&1
cat gpurun_out/lab/exp_a.txt

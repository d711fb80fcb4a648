# same-box A/B of the image-to-XCD numbering of the encoder kernels: previous build / XPT_XCD_AFFINITY=0 / 1, then kernel traces
mkdir -p gpurun_out/lab
BASE=$PWD/xpt_mde_2021_amd/libxpt_hip_base.so
{
timeout -k 10 900 python -m pytest tests/test_cell_tail_gpu.py tests/test_encoder_parity_gpu.py tests/test_grad_sink.py tests/test_small_map_grads.py tests/test_hip_parity.py tests/test_ref_nets.py tests/test_graph_replay.py tests/test_conv_igemm_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
echo "base : $(XPT_HIP_LIB=$BASE bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "xcd=1: $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
done
} > gpurun_out/lab/exp_xcd.txt 2>&1
bash tools/profile_step.sh xcd1 > /dev/null 2>&1
XPT_HIP_LIB=$BASE bash tools/profile_step.sh xcdbase > /dev/null 2>&1
rm -f gpurun_out/prof/xcd*_kernel_stats_whole_run.csv
cat gpurun_out/lab/exp_xcd.txt

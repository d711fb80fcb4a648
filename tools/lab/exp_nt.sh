mkdir -p gpurun_out/lab
BASE=$PWD/xpt_mde_2021_amd/libxpt_hip_base.so
{
for i in 1 2 3; do
echo "base: $(XPT_HIP_LIB=$BASE bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "nt  : $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
done
} > gpurun_out/lab/exp_nt.txt 2>&1
cat gpurun_out/lab/exp_nt.txt

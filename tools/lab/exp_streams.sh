mkdir -p gpurun_out/lab
{
echo "base: $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "base: $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "net-streams: $(bash tools/quick_bench.sh --steps 40 --warmup 10 --net-streams 1)"
echo "wgrad side: $(XPT_WGRAD_SIDE_STREAM=1 bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "both: $(XPT_WGRAD_SIDE_STREAM=1 bash tools/quick_bench.sh --steps 40 --warmup 10 --net-streams 1)"
echo "wgrad 6MiB: $(XPT_WGRAD_TUNE=6,0 bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "wgrad 3MiB: $(XPT_WGRAD_TUNE=3,0 bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "wgrad 24MiB: $(XPT_WGRAD_TUNE=24,0 bash tools/quick_bench.sh --steps 40 --warmup 10)"
} > gpurun_out/lab/exp1.txt 2>&1
cat gpurun_out/lab/exp1.txt

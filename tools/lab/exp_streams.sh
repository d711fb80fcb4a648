mkdir -p gpurun_out/lab
{
for i in 1 2; do
echo "one stream : $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "net-streams: $(bash tools/quick_bench.sh --steps 40 --warmup 10 --net-streams 1)"
done
echo "distributed one stream : $(bash tools/quick_bench.sh --steps 40 --warmup 10 --mode distributed)"
echo "distributed net-streams: $(bash tools/quick_bench.sh --steps 40 --warmup 10 --mode distributed --net-streams 1)"
echo "batch16 one stream : $(bash tools/quick_bench.sh --steps 30 --warmup 10 --batch 16)"
echo "batch16 net-streams: $(bash tools/quick_bench.sh --steps 30 --warmup 10 --batch 16 --net-streams 1)"
echo "stereo one stream : $(bash tools/quick_bench.sh --steps 30 --warmup 10 --stereo)"
echo "stereo net-streams: $(bash tools/quick_bench.sh --steps 30 --warmup 10 --stereo --net-streams 1)"
} > gpurun_out/lab/exp_streams.txt 2>&1
cat gpurun_out/lab/exp_streams.txt

mkdir -p gpurun_out/lab
{
for i in 1 2; do
echo "now        : $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "wgrad defer: $(XPT_WGRAD_DEFER=1 bash tools/quick_bench.sh --steps 40 --warmup 10)"
done
echo "distributed now  : $(bash tools/quick_bench.sh --steps 40 --warmup 10 --mode distributed)"
echo "distributed defer: $(XPT_WGRAD_DEFER=1 bash tools/quick_bench.sh --steps 40 --warmup 10 --mode distributed)"
python tools/determinism_train.py graph noaug 6 2>&1 | grep "LOSSES\|PARAMSUM"
XPT_WGRAD_DEFER=1 python tools/determinism_train.py graph noaug 6 2>&1 | grep "LOSSES\|PARAMSUM\|Error\|error" | head -5
} > gpurun_out/lab/exp_defer.txt 2>&1
cat gpurun_out/lab/exp_defer.txt

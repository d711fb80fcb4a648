mkdir -p gpurun_out/lab
{
python tools/determinism_train.py graph noaug 8 2>&1 | grep "LOSSES\|PARAMSUM"
XPT_EARLY_UPDATE=1 python tools/determinism_train.py graph noaug 8 2>&1 | grep "LOSSES\|PARAMSUM\|Error\|error" | head -5
for i in 1 2; do
echo "plain: $(bash tools/quick_bench.sh --steps 40 --warmup 10)"
echo "early: $(XPT_EARLY_UPDATE=1 bash tools/quick_bench.sh --steps 40 --warmup 10)"
done
echo "batch16 plain: $(bash tools/quick_bench.sh --steps 30 --warmup 10 --batch 16)"
echo "batch16 early: $(XPT_EARLY_UPDATE=1 bash tools/quick_bench.sh --steps 30 --warmup 10 --batch 16)"
echo "c4 plain: $(bash tools/quick_bench.sh --steps 30 --warmup 10 --config c4)"
echo "c4 early: $(XPT_EARLY_UPDATE=1 bash tools/quick_bench.sh --steps 30 --warmup 10 --config c4)"
} > gpurun_out/lab/exp_early.txt 2>&1
cat gpurun_out/lab/exp_early.txt

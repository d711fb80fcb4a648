mkdir -p gpurun_out/lab
for a in "--dtype fp32" "--nets joint --width 384"; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-host-fed $a 2>gpurun_out/lab/fp32chk_err.txt | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['config']['mode'], d['config'].get('graph_nodes'))"
grep -i "fallback\|eager\|memset\|StepGraph" gpurun_out/lab/fp32chk_err.txt | head -5
done

# The "other configurations" table of DESIGN.md section 7: one bench line per configuration (20 timed steps each).
set -u
run() { echo -n "$* :: "; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-host-fed "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['config']['mode'][:40])"; }
run
run --batch 16
run --batch 32
run --config c4
run --stereo
run --config c5
run --mode distributed
XPT_DP_OVERLAP=1 run --mode distributed
run --mode eager
run --dtype fp16
run --dtype fp32
run --nets flow --width 384
run --nets joint --width 384

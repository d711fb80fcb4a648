#!/bin/bash
# usage: tools/quick_bench.sh [bench.py flags...]   -> "<images/s> <ms/step> <median step ms>"
timeout -k 10 250 python bench.py --no-cpu-baseline --no-roofline --no-host-fed --sustained-seconds 0 "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['step_ms']['median'])"

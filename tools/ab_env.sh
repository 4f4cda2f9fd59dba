#!/bin/bash
# A/B of one environment switch on bench.py: tools/ab_env.sh VAR=value [repeats] -> ms per sweep, rrLU ms, dominant launch ms
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['breakdown_ms_per_sweep']['rrlu_kernel'],2), round(d['roofline']['avg_launch_ms'],4))"; }
for i in $(seq 1 ${2:-3}); do
timeout 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | pj default
env "$1" timeout 200 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | pj "$1"
done

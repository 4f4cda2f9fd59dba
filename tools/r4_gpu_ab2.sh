#!/bin/bash
# A/B of an environment switch on the headline: alternating runs of bench.py (no aux, no CPU leg).  usage: r4_gpu_ab2.sh OUTDIR "ENV=1" [reps]
O=gpurun_out/$1; mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
reps=${3:-3}
for r in $(seq 1 $reps); do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-aux 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step'],3))"
  env $2 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-aux 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],3))"
done | tee $O/ab.txt

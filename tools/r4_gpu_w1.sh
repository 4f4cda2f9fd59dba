#!/bin/bash
# round-4 GPU session: the one-wave rrLU kernel — time per step against the one-workgroup kernel and what ran before, the GPU
# test suite, the bench line with and without it.   usage: tools/r4_gpu_w1.sh OUTDIR
O=gpurun_out/$1
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
T4A_WG_MIN=0 timeout 900 python tools/probe_wg.py 8 8 8 16 16 16 32 32 32 64 32 32 32 64 32 64 64 64 > $O/probe_w1.log 2>&1
cat $O/probe_w1.log
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2; do
for cfg in "w1:T4A_X=0" "now1:T4A_NO_W1=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_${name}_$rep.json 2> $O/bench_$name.err
done
done
for f in $O/bench_*.json; do echo -n "$f: "; python -c "import json,sys; d=json.load(open('$f')); a=d.get('aux',{}); print(round(d['ms_per_step'],3), {k.replace('cfg5_patch_from_scratch_ms_','c5_').replace('_time_to_solution',''):(round(v,2) if isinstance(v,float) else v) for k,v in a.items() if 'ms' in k})"; done

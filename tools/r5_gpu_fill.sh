#!/bin/bash
O=gpurun_out/fill; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" updold; do
if [ -n "$v" ]; then export T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_$v.so; else unset T4A_GPU_LIB; fi
echo "== lib=${v:-default}"
timeout 300 python tools/probe_fill.py 30
timeout 300 python bench.py --no-cpu-baseline --no-aux --no-floor --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['breakdown_ms_per_sweep'])"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/stats$v -o x --output-format csv -- python3 bench.py --no-cpu-baseline --no-aux --no-floor --steps 10 --warmup 3 > $O/stats.log 2>&1 </dev/null
grep "lu_panel\|lu_update\|lu_solve\|pi_eval_b" $O/stats$v/x_kernel_stats.csv | cut -c1-150
rm -rf $O/stats$v
done

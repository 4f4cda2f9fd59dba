#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sf in 0.5 0.8 0.95 0.99 1.5; do
for v in nopf_diag novm_diag; do
export T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_$v.so
echo -n "== specfrac=$sf lib=$v  "
T4A_XCD_SPECFRAC=$sf timeout 300 python bench.py --no-cpu-baseline --no-aux --no-floor --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['breakdown_ms_per_sweep']['rrlu_kernel'],3), round(d['breakdown_ms_per_sweep']['fill_site_tensors'],2))"
done
done

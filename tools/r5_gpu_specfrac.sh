#!/bin/bash
# speculation threshold of the single-XCD rrLU kernel (T4A_XCD_SPECFRAC, diagnostic build of engine.hip): ms of rrLU kernels per cfg3 sweep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_diag.so
for rep in 1 2; do
for sf in 0.6 0.7 0.75 0.8 0.85 0.9; do
echo -n "specfrac=$sf  "
T4A_XCD_SPECFRAC=$sf timeout 300 python bench.py --no-cpu-baseline --no-aux --no-floor --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['breakdown_ms_per_sweep']['rrlu_kernel'],3))"
done
done

#!/bin/bash
# full keys requested together with the early keys (T4A_X2_KHEARLY=1, lib/libt4a_gpu_khe.so) against the production kernel
O=gpurun_out/khe; mkdir -p $O
L=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['breakdown_ms_per_sweep']['rrlu_kernel'],2), round(d['roofline']['avg_launch_ms'],4))"; }
for i in 1 2 3; do
for v in "" khe; do
if [ -n "$v" ]; then export T4A_GPU_LIB=$L/libt4a_gpu_$v.so; else unset T4A_GPU_LIB; fi
timeout 300 python bench.py --no-cpu-baseline --no-aux --no-floor --steps 10 2>/dev/null | tail -1 | pj "v=$v"
done; done | tee $O/ab.txt
T4A_GPU_LIB=$L/libt4a_gpu_khe.so timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_fuzz.py tests/test_gpu_chain.py -x -q 2>&1 | tail -3

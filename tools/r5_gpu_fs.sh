#!/bin/bash
# staggered reads of the remote finalists (default build) against a single early read (lib/libt4a_gpu_fs0.so): cfg4-size sweep
O=gpurun_out/fs; mkdir -p $O
L=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib
for i in 1 2; do
for v in "" fs0; do
if [ -n "$v" ]; then export T4A_GPU_LIB=$L/libt4a_gpu_$v.so; else unset T4A_GPU_LIB; fi
echo "== lib=${v:-default}"; timeout 600 python tools/probe_cfg4_variants.py 2>&1 | grep "full sweep\|xcd2m_kernel<24, 2, .*, 3>" | cut -c1-150
done; done | tee $O/ab.txt
unset T4A_GPU_LIB
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_tci2.py -x -q -k "xcd or cfg4 or multi or big or large" 2>&1 | tail -3

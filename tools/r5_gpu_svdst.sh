#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/svdst
echo "== production"; timeout 300 python tools/probe_linalg.py 2>&1 | grep -v "^qr"
if [ -n "$STAMPS" ]; then
export T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_diag.so
timeout 300 python tools/probe_linalg.py > gpurun_out/svdst/log.txt 2>&1
grep "stamps" gpurun_out/svdst/log.txt | grep "m=256" | awk 'NR%40==1' | head -8 | cut -c1-260
grep "stamps" gpurun_out/svdst/log.txt | grep "m=64" | awk 'NR%40==1' | head -4 | cut -c1-260
unset T4A_GPU_LIB
fi
timeout 900 python -m pytest tests/test_gpu_tt.py tests/test_gpu_dense.py -x -q -m gpu -k "svd or compress or qr" 2>&1 | tail -4

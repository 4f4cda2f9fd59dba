#!/bin/bash
# phase stamps of jacobi_block_kernel (library built with -DT4A_JB_STAMPS, see tools/README.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/svdst
T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_jbst.so timeout 300 python tools/probe_svd_once.py 512 256 > gpurun_out/svdst/log.txt 2>&1
grep "stamps" gpurun_out/svdst/log.txt | grep "m=256" | awk 'NR%25==1' | head -14 | cut -c1-260

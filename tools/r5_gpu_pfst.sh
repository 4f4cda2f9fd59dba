#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in alt_w3; do
echo -n "$lib: "
T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_$lib.so T4A_RRLU_STAMPS=1 timeout 120 python3 tools/probe_xcd.py child 685 688 256 1 2>&1 | grep "stamps xcd" | tail -1 | cut -c52-400
done

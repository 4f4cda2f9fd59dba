#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_diag.so
echo "== fused small fill"; timeout 300 python tools/probe_small_fill.py
echo "== general path"; T4A_NO_SMALL_FILL=1 timeout 300 python tools/probe_small_fill.py
unset T4A_GPU_LIB
echo "== production lib"; timeout 300 python tools/probe_small_fill.py
T4A_OPT_PROF=1 T4A_HOST_PROFILE=1 timeout 300 python tools/probe_cfg2_host.py 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_tci2.py -x -q -m gpu 2>&1 | tail -5

#!/bin/bash
# phase stamps of the one-wave kernel (diagnostic twin library).   usage: tools/r4_gpu_w1s.sh OUTDIR
O=gpurun_out/$1
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
export T4A_GPU_LIB=$PWD/tensor4all-rs_amd/lib/libt4a_gpu_alt.so T4A_RRLU_STAMPS=1 T4A_WG_MIN=0
for s in "8 8 8 1" "16 16 16 1" "32 32 32 1" "64 64 64 1" "64 64 64 0"; do
  timeout 120 python tools/probe_xcd.py child $s 2>&1 | tail -2
done > $O/stamps_w1.log 2>&1
cat $O/stamps_w1.log

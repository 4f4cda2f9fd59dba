#!/bin/bash
# poller A/B on single matrices (GPU): default kernel vs round-2 kernel, with phase stamps
L=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib
for shape in "660 660 256" "448 448 200"; do
  for lib in "" p0; do
    for left in 1; do
      if [ -n "$lib" ]; then export T4A_GPU_LIB=$L/libt4a_gpu_$lib.so; else unset T4A_GPU_LIB; fi
      echo "== lib=${lib:-default} shape=$shape"; timeout 120 python3 tools/probe_xcd.py child $shape $left
    done
  done
  for st in 1 2; do T4A_GPU_LIB=$L/libt4a_gpu_alt.so T4A_RRLU_STAMPS=$st timeout 120 python3 tools/probe_xcd.py child $shape 1 2>&1 | grep "stamps xcd" | tail -1; done
  T4A_GPU_LIB=$L/libt4a_gpu_p0s.so T4A_RRLU_STAMPS=1 timeout 120 python3 tools/probe_xcd.py child $shape 1 2>&1 | grep "stamps xcd" | tail -1
done

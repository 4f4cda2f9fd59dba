#!/bin/bash
# polling wave: how long to sleep after barrier (C) before the first key sweep (units of s_sleep 1), single matrices
L=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib
for shape in "660 660 256" "685 688 256" "300 300 128"; do
  for s in 0 2 4 6 8 10 12 16; do
    echo -n "sleep $s: "; T4A_XCD_POLLSLEEP=$s T4A_GPU_LIB=$L/libt4a_gpu_alt.so T4A_RRLU_STAMPS=2 timeout 120 python3 tools/probe_xcd.py child $shape 1 2>&1 | grep "stamps xcd" | tail -1 | sed 's/launch (cycles).*steps=/steps_total=/; s/write-out.*//' | cut -c1-330
  done
done

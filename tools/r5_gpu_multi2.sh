#!/bin/bash
# round-5: kernels for matrices beyond one XCD — fuzz parity, phase stamps at 1464 x 1448 (K = 3) and 1424 x 512 (K = 1), cfg4 tests
O=gpurun_out/$1; mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
L=$PWD/tensor4all-rs_amd/lib
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "beyond_one_xcd or widest" > $O/pytest_fuzz.log 2>&1; echo "rc=$?" >> $O/pytest_fuzz.log; tail -n 6 $O/pytest_fuzz.log
for sh in "1464 1448 256" "1428 1024 256" "1424 512 256" "1024 1024 256"; do
  T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$L/libt4a_gpu_alt.so timeout 120 python tools/probe_xcd.py child $sh 1 2>&1 | grep stamps | tail -n 1
  T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$L/libt4a_gpu_alt_w3.so timeout 120 python tools/probe_xcd.py child $sh 1 2>&1 | grep stamps | tail -n 1
done | tee $O/stamps.txt
timeout 900 python -m pytest tests/test_gpu_tci2.py -m gpu -x -q -k "cfg4" > $O/pytest_cfg4.log 2>&1; echo "rc=$?" >> $O/pytest_cfg4.log; tail -n 5 $O/pytest_cfg4.log

#!/bin/bash
# round-5: kernels for matrices beyond one XCD (kernels_rrlu_xcd2m.hip) — parity first, then per-step times and the cfg4-size sweep
O=gpurun_out/$1; mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "beyond_one_xcd or widest" > $O/pytest_fuzz.log 2>&1; echo "rc=$?" >> $O/pytest_fuzz.log; tail -n 15 $O/pytest_fuzz.log
for sh in "1464 1448 256" "1024 1428 256" "1424 512 256" "1536 1536 128" "1024 1024 256"; do
  for K in "" 2 3; do
    T4A_XCD_K=$K timeout 120 python tools/probe_xcd.py child $sh 1 2>&1 | sed "s/^/K=${K:-auto} /" | tail -n 1
  done
  T4A_NO_XCD_BIG=1 timeout 120 python tools/probe_xcd.py child $sh 1 2>&1 | sed "s/^/nobig /" | tail -n 1
done | tee $O/probe.txt
timeout 900 python -m pytest tests/test_gpu_tci2.py -m gpu -x -q -k "cfg4" > $O/pytest_cfg4.log 2>&1; echo "rc=$?" >> $O/pytest_cfg4.log; tail -n 15 $O/pytest_cfg4.log
timeout 600 python tools/probe_cfg4_variants.py 2>&1 | head -30 | tee $O/cfg4_variants.txt

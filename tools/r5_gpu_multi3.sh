#!/bin/bash
# round-5: two-level arg-max of the kernels for matrices beyond one XCD — fuzz parity, stamps, cfg4 tests, cfg4-size sweep by kernel
O=gpurun_out/$1; mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
L=$PWD/tensor4all-rs_amd/lib
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "beyond_one_xcd or widest" > $O/pytest_fuzz.log 2>&1; echo "rc=$?" >> $O/pytest_fuzz.log; tail -n 6 $O/pytest_fuzz.log
for sh in "1464 1448 256" "1428 1024 256" "1424 512 256"; do
  T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$L/libt4a_gpu_alt.so timeout 120 python tools/probe_xcd.py child $sh 1 2>&1 | grep stamps | tail -n 1
done | tee $O/stamps.txt
timeout 600 python tools/probe_cfg4_variants.py 2>&1 | head -12 | tee $O/cfg4_variants.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|Error" $O/pytest.log | tail -8

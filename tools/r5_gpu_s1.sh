#!/bin/bash
# round-5 GPU session 1: deferred-full-key protocol of the second-generation single-XCD kernel — digests against the chip-wide kernel,
# per-step time, phase stamps (polling wave + wave 3), rrLU parity tests, A/B of the headline against the round-4 kernel
O=gpurun_out/r5_s1
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
L=$PWD/tensor4all-rs_amd/lib
SH="685 688 256 512 512 256 768 768 256 256 256 128 100 700 100 700 100 100 130 130 130 300 260 100 1024 1024 64"
echo "== probe new" > $O/probe.log
timeout 600 python tools/probe_xcd.py $SH >> $O/probe.log 2>&1
echo "== probe base" >> $O/probe.log
T4A_GPU_LIB=$L/libt4a_gpu_base.so timeout 300 python tools/probe_xcd.py child 685 688 256 1 >> $O/probe.log 2>&1
T4A_GPU_LIB=$L/libt4a_gpu_base.so timeout 300 python tools/probe_xcd.py child 512 512 256 1 >> $O/probe.log 2>&1
echo "== stamps" > $O/stamps.log
T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$L/libt4a_gpu_alt.so timeout 300 python tools/probe_xcd.py child 685 688 256 1 >> $O/stamps.log 2>&1
T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$L/libt4a_gpu_alt_w3.so timeout 300 python tools/probe_xcd.py child 685 688 256 1 >> $O/stamps.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_dense.py tests/test_gpu_fuzz.py tests/test_gpu_rrlu_global.py -m gpu -x -q > $O/pytest_rrlu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_rrlu.log
for r in 1 2; do
for v in "" _base; do
  T4A_GPU_LIB=$L/libt4a_gpu$v.so timeout 300 python bench.py --no-cpu-baseline --no-aux --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$v', d['ms_per_step'], d['roofline']['latency_view']['us_per_pivot_step'])"
done; done | tee $O/ab.txt
grep -v "^reg" $O/probe.log | tail -40
cat $O/stamps.log | grep stamps
tail -3 $O/pytest_rrlu.log

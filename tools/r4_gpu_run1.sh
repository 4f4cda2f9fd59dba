#!/bin/bash
# round-4 GPU session 1: second-generation single-XCD kernel — digests against the chip-wide kernel, per-step time both generations,
# phase stamps, the GPU test suite, headline bench both generations
O=gpurun_out/r4_run1
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
SH="685 688 256 512 512 256 768 768 256 256 256 128 100 700 100 700 100 100 130 130 130 300 260 100"
echo "== probe V=2" > $O/probe.log
T4A_XCD_V=2 timeout 600 python tools/probe_xcd.py $SH >> $O/probe.log 2>&1
echo "== probe V=1" >> $O/probe.log
T4A_XCD_V=1 timeout 600 python tools/probe_xcd.py 685 688 256 512 512 256 >> $O/probe.log 2>&1
echo "== stamps" > $O/stamps.log
for V in 2 1; do
T4A_XCD_V=$V T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$PWD/tensor4all-rs_amd/lib/libt4a_gpu_alt.so timeout 300 python tools/probe_xcd.py child 685 688 256 1 >> $O/stamps.log 2>&1
done
T4A_XCD_V=2 T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$PWD/tensor4all-rs_amd/lib/libt4a_gpu_alt_w3.so timeout 300 python tools/probe_xcd.py child 685 688 256 1 >> $O/stamps.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux > $O/bench_v2.json 2> $O/bench_v2.err
T4A_XCD_V=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux > $O/bench_v1.json 2> $O/bench_v1.err
tail -3 $O/pytest.log
cat $O/bench_v2.json | cut -c1-400
cat $O/bench_v1.json | cut -c1-400

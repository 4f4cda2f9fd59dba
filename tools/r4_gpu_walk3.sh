#!/bin/bash
# one-wave kernel stamps + probe, the persistent half-sweep's phases, parity tests of everything that can reach them
O=gpurun_out/$1
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
( export T4A_GPU_LIB=$PWD/tensor4all-rs_amd/lib/libt4a_gpu_alt.so T4A_RRLU_STAMPS=1 T4A_WG_MIN=0 T4A_W1_MAXN=64
for s in "8 8 8 1" "16 16 16 1" "32 32 32 1" "64 64 64 1"; do timeout 120 python tools/probe_xcd.py child $s 2>&1 | tail -2; done ) > $O/stamps_w1.log 2>&1
cut -c1-330 $O/stamps_w1.log
T4A_WG_MIN=0 T4A_W1_MAXN=64 timeout 900 python tools/probe_wg.py 8 8 8 16 16 16 32 32 32 64 64 64 > $O/probe_w1.log 2>&1
cat $O/probe_w1.log
T4A_WALK_DEBUG=1 T4A_OPT_PROF=1 timeout 120 python3 tools/probe_cfg2_trace.py 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|Error" $O/pytest.log | tail -5

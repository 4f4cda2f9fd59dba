#!/bin/bash
# quick: probe of the one-wave kernel + the rrLU / chain / fuzz tests.   usage: tools/r4_gpu_w1b.sh OUTDIR
O=gpurun_out/$1
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
T4A_WG_MIN=0 timeout 900 python tools/probe_wg.py 8 8 8 16 16 16 32 32 32 64 32 32 32 64 32 64 64 64 > $O/probe_w1.log 2>&1
cat $O/probe_w1.log
timeout 1800 python -m pytest tests -m gpu -x -q -k "dense or fuzz or chain or rrlu or luci" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error" $O/pytest.log | tail -3

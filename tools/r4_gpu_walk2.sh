#!/bin/bash
# the persistent half-sweep: phase times inside the kernel (T4A_WALK_DEBUG) for configs[1]
O=gpurun_out/$1
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
T4A_WALK_DEBUG=1 T4A_OPT_PROF=1 timeout 120 python3 tools/probe_cfg2_trace.py > $O/walkdbg.log 2>&1
tail -14 $O/walkdbg.log

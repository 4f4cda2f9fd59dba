#!/bin/bash
O=gpurun_out/svd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/stats -o x --output-format csv -- python3 tools/probe_linalg.py > $O/stats.log 2>&1 </dev/null
head -12 $O/stats/x_kernel_stats.csv | cut -c1-200
python3 tools/trace_idle.py $O/stats/x_kernel_trace.csv 10 2>&1 | head -12
rm -rf $O/stats

#!/bin/bash
O=gpurun_out/fill; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/probe_fill.py 2>&1 | tee $O/probe_fill.txt
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/stats -o x --output-format csv -- python3 tools/probe_fill.py 30 > $O/stats.log 2>&1 </dev/null
grep "lu_panel\|lu_update\|lu_solve\|trsm\|pi_eval\|pack_fill" $O/stats/x_kernel_stats.csv | cut -c1-170
rm -rf $O/stats
timeout 900 python -m pytest tests/test_gpu_dense.py -q -k "solve or trsm or luci" 2>&1 | tail -3

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 python3 tools/probe_tree7.py 2>&1 | tail -3
O=gpurun_out/tree7; mkdir -p $O
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/t -o x --output-format csv -- python3 tools/probe_tree7.py > $O/log.txt 2>&1 </dev/null
python3 - <<'P'
import csv
rows=list(csv.DictReader(open("gpurun_out/tree7/t/x_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-150:]
t0=int(rows[0]["Start_Timestamp"]); prev=t0
for r in rows[:60]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%9.1f us  gap %7.1f  dur %6.1f  %s"%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,r["Kernel_Name"][:70]))
    prev=e
P
rm -rf $O/t

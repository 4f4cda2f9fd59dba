#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/svdtr; mkdir -p $O
for sz in "512 256 qr" "64 64 qr" "512 256" "64 64"; do
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/t -o x --output-format csv -- python3 tools/probe_svd_once.py $sz > $O/log.txt 2>&1 </dev/null
tail -1 $O/log.txt
python - <<'P'
import csv
rows=list(csv.DictReader(open("gpurun_out/svdtr/t/x_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per call (6 calls): %.3f ms"%(tot/6e6))
for r in rows[:4]:
    print("%-90s calls %6s  total %9.1f us/call  avg %8.1f us"%(r["Name"][:90], r["Calls"], float(r["TotalDurationNs"])/6e3, float(r["AverageNs"])/1e3))
P
rm -rf $O/t
done
T4A_SVD_DEBUG=1 timeout 300 python tools/probe_linalg.py 2>&1 | sort | uniq -c | sort -k2 | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_tt.py tests/test_gpu_dense.py tests/test_gpu_tensor.py -x -q -m gpu 2>&1 | tail -3

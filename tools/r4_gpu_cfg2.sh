#!/bin/bash
# round-4 GPU session: where BASELINE configs[1] (rank 2, d = 20) spends its 1.4 ms — host phases and the kernel timeline
O=gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=tensor4all-rs_amd/python
T4A_OPT_PROF=1 timeout 120 python3 tools/probe_cfg2_trace.py > $O/optprof.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/tr -o x --output-format csv -- python3 tools/probe_cfg2_trace.py > $O/trace.log 2>&1 </dev/null
python3 - <<'PY' > $O/timeline.txt
import csv, os, sys
p = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "", "tr")
PY
python3 - "$O" <<'PY' > $O/timeline.txt
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/tr/**/x_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last solve: take the final 300 kernels
rows = rows[-260:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
PY
head -50 $O/tr/*/x_kernel_stats.csv 2>/dev/null | cut -c1-160 > $O/stats_head.txt
rm -rf $O/tr
tail -30 $O/optprof.log

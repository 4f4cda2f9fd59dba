#!/bin/bash
# round-4 GPU session: the persistent half-sweep — parity tests, configs[1] with and without it (host phases + kernel timeline)
O=gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=tensor4all-rs_amd/python
timeout 1500 python -m pytest tests -m gpu -x -q -k "chain or tci2 or optimize or patch" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|Error" $O/pytest.log | tail -5
T4A_OPT_PROF=1 timeout 120 python3 tools/probe_cfg2_trace.py > $O/optprof_walk.log 2>&1
T4A_NO_WALK=1 T4A_OPT_PROF=1 timeout 120 python3 tools/probe_cfg2_trace.py > $O/optprof_nowalk.log 2>&1
tail -12 $O/optprof_walk.log; echo ---; tail -4 $O/optprof_nowalk.log
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/tr -o x --output-format csv -- python3 tools/probe_cfg2_trace.py > $O/trace.log 2>&1 </dev/null
python3 - "$O" <<'PY' > $O/timeline.txt
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/tr/**/x_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-120:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
PY
rm -rf $O/tr
tail -70 $O/timeline.txt

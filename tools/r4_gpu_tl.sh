#!/bin/bash
# kernel timeline of the last configs[1] solve (rocprofv3 --kernel-trace).   usage: tools/r4_gpu_tl.sh OUTDIR
O=gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=tensor4all-rs_amd/python
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/tr -o x --output-format csv -- python3 tools/probe_cfg2_trace.py > $O/trace.log 2>&1 </dev/null
python3 - "$O" <<'PY' > $O/timeline.txt
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/tr/**/x_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-90:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("t4a::(anonymous namespace)::", "").split("(")[0][:48]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
PY
rm -rf $O/tr
tail -4 $O/trace.log
cat $O/timeline.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_diag.so
for w in 8 16; do
echo "== T4A_SVD_BLOCK_W=$w"
T4A_SVD_BLOCK_W=$w T4A_SVD_DEBUG=1 timeout 300 python tools/probe_linalg.py 2>&1 | grep -v "^qr" | sort | uniq -c | sort -k2 | cut -c1-220
done
O=gpurun_out/svdw; mkdir -p $O
T4A_SVD_BLOCK_W=8 timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/s8 -o x --output-format csv -- python3 tools/probe_linalg.py > $O/s8.log 2>&1 </dev/null
grep "jacobi" $O/s8/x_kernel_stats.csv | cut -c1-200
T4A_SVD_BLOCK_W=16 timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/s16 -o x --output-format csv -- python3 tools/probe_linalg.py > $O/s16.log 2>&1 </dev/null
grep "jacobi" $O/s16/x_kernel_stats.csv | cut -c1-200
python - <<'P'
import csv,sys
for w in ("s8","s16"):
    rows=list(csv.DictReader(open(f"gpurun_out/svdw/{w}/x_kernel_trace.csv")))
    rows=[r for r in rows if "jacobi_block" in r["Kernel_Name"]]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    gaps=[];durs=[]
    for a,b in zip(rows,rows[1:]):
        g=int(b["Start_Timestamp"])-int(a["End_Timestamp"])
        if g<50000: gaps.append(g)
    durs=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
    print(w,"launches",len(rows),"mean dur us",sum(durs)/len(durs)/1e3,"mean gap us",sum(gaps)/max(1,len(gaps))/1e3)
P
rm -rf $O/s8 $O/s16

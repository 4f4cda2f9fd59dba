#!/bin/bash
# round 4, second half: evidence for the one-wave kernel, the persistent half-sweep and the chained 1-site sweep -> gpurun_out/$1
O=gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=tensor4all-rs_amd/python
{
echo "# one-wave rrLU kernel (kernels_rrlu_w1_body.hpp): phase stamps of a diagnostic build (tools/build_stamps_lib.sh, T4A_RRLU_STAMPS=1,"
echo "# T4A_W1_MAXN=64 T4A_WG_MIN=0 so that every shape takes it) and time per pivot step against the one-workgroup kernel (wg) and what ran"
echo "# before both (old) — slope between a factorisation capped at r and at r / 2 steps (tools/probe_wg.py)"
( export T4A_GPU_LIB=$PWD/tensor4all-rs_amd/lib/libt4a_gpu_alt.so T4A_RRLU_STAMPS=1 T4A_WG_MIN=0 T4A_W1_MAXN=64
for s in "8 8 8 1" "16 16 16 1" "32 32 32 1" "64 64 64 1"; do timeout 120 python tools/probe_xcd.py child $s 2>&1 | tail -2; done )
T4A_WG_MIN=0 T4A_W1_MAXN=64 timeout 900 python tools/probe_wg.py 8 8 8 16 16 16 32 32 32 64 64 64
} > $O/w1_probe.txt 2>&1
{
echo "# BASELINE configs[1] (d = 20, cos(10x) exp(-x), tol 1e-8, chi <= 64; rank 2): four solves — phases inside the persistent"
echo "# half-sweep kernel (T4A_WALK_DEBUG=1), host phases of optimize (T4A_OPT_PROF=1); then the same without the persistent workgroup"
echo "# (T4A_NO_WALK=1), without the chained 1-site sweep (T4A_NO_CHAIN_1SITE=1), and without both (the state before); T4A_WALK_OLD_PREP=1: the 1 024-thread preparation inside the walk"
T4A_WALK_DEBUG=1 T4A_OPT_PROF=1 timeout 120 python3 tools/probe_cfg2_trace.py 2>&1 | tail -14
echo "# T4A_NO_WALK=1"; T4A_NO_WALK=1 timeout 120 python3 tools/probe_cfg2_trace.py 2>&1 | tail -3
echo "# T4A_NO_CHAIN_1SITE=1"; T4A_NO_CHAIN_1SITE=1 timeout 120 python3 tools/probe_cfg2_trace.py 2>&1 | tail -3
echo "# T4A_NO_WALK=1 T4A_NO_CHAIN_1SITE=1 T4A_NO_W1=1"; T4A_NO_WALK=1 T4A_NO_CHAIN_1SITE=1 T4A_NO_W1=1 timeout 120 python3 tools/probe_cfg2_trace.py 2>&1 | tail -3
} > $O/cfg2_walk_phases.txt 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/tr -o x --output-format csv -- python3 tools/probe_cfg2_trace.py > $O/trace.log 2>&1 </dev/null
python3 - "$O" <<'PY' > $O/cfg2_timeline.txt
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/tr/**/x_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-46:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
print("# kernel timeline of the last configs[1] solve under rocprofv3 --kernel-trace (tools/probe_cfg2_trace.py): start, gap to the previous kernel's end, duration")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("t4a::(anonymous namespace)::", "").split("(")[0][:56]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
PY
rm -rf $O/tr
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench.err
tail -c 1500 $O/bench_n1.json

#!/bin/bash
# dedicated polling wave: phase stamps (poller = wave 8, agent wave 3) and the sleep in front of the first poll
O=gpurun_out/dp; mkdir -p $O
L=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib
for w in alt_w8 alt_w3 alt; do
T4A_GPU_LIB=$L/libt4a_gpu_$w.so T4A_RRLU_STAMPS=1 timeout 120 python3 tools/probe_xcd.py child 685 688 256 1 2>&1 | grep "stamps xcd" | tail -1 | sed "s/^/$w /"
done | tee $O/stamps.txt
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), round(d['breakdown_ms_per_sweep']['rrlu_kernel'],2), round(d['roofline']['avg_launch_ms'],4))"; }
for v in "" ps12 ps28 dp0; do
if [ -n "$v" ]; then export T4A_GPU_LIB=$L/libt4a_gpu_$v.so; else unset T4A_GPU_LIB; fi
timeout 300 python bench.py --no-cpu-baseline --no-aux --no-floor --steps 10 2>/dev/null | tail -1 | pj "v=$v"
done | tee $O/ab.txt

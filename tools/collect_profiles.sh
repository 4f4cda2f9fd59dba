#!/bin/bash
# Round-2 profile artifacts, to be run on the GPU box from the repo root (writes gpurun_out/r02/*; copy into profiles/).
# Every rocprofv3 call profiles a single-process program under `timeout -k` with stdin closed (a profiled process that does
# not exit would otherwise hold the box until gpurun's own limit).
# The phase stamps need the diagnostic build of the single-XCD kernel next to the default one (built here, CPU side):
#   hipcc <flags of build.py> -DT4A_XCD_STAMPS -c csrc/kernels_rrlu_xcd.hip, linked with the other objects into
#   tensor4all-rs_amd/lib/libt4a_gpu_alt.so   (tools/build_stamps_lib.sh)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-aux"
prof() { timeout -k 5 300 rocprofv3 "$@" </dev/null; }
timeout 600 python3 bench.py --steps 10 --warmup 3 2>$O/bench_n1.err | tail -1 > $O/bench_n1.json
prof --kernel-trace --stats -d $O/stats -o x --output-format csv -- $B --steps 10 --warmup 3 > $O/stats.log 2>&1
python3 tools/trace_gaps.py $O/stats/x_kernel_trace.csv 150 > $O/launch_gaps.txt 2>&1
prof --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o x --output-format csv -- $B > $O/pmc_fetch.log 2>&1
prof --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o x --output-format csv -- $B > $O/pmc_write.log 2>&1
prof --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA -d $O/pmc_mfma -o x --output-format csv -- $B > $O/pmc_mfma.log 2>&1
prof --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o x --output-format csv -- $B > $O/pmc_sq.log 2>&1
python3 tools/pmc_sq_summary.py $O/pmc_sq $O/pmc_sq_instruction_mix_per_kernel.csv "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $B" > /dev/null
python3 tools/pmc_sq_summary.py $O/pmc_mfma $O/pmc_mfma_per_kernel.csv "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA -- $B" > /dev/null
timeout 120 tools/xcd_bench > $O/xcd_bench.log 2>&1
timeout 60 tools/mfma_peak > $O/mfma_peak.log 2>&1
prof --kernel-trace -d $O/gemm -o x --output-format csv -- python3 tools/probe_gemm.py > $O/gemm.log 2>&1
python3 tools/gemm_trace_summary.py $O/gemm/x_kernel_trace.csv > $O/gemm_probe.txt 2>&1
if [ -f tensor4all-rs_amd/lib/libt4a_gpu_alt.so ]; then
  T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_alt.so T4A_RRLU_STAMPS=1 timeout 120 python3 tools/probe_xcd.py child 685 688 256 1 2>&1 | grep "stamps xcd" | tail -1 > $O/xcd_phase_stamps.txt
fi
# keep only the summaries (the merge back is limited to 64 MiB)
for d in stats pmc_fetch pmc_write gemm; do find $O/$d -name "*agent_info.csv" -delete 2>/dev/null; done
rm -rf $O/pmc_mfma $O/pmc_sq
ls -R $O | head -60

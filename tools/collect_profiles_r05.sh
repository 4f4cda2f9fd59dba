#!/bin/bash
# Round-5 profile artifacts, to be run on the GPU box from the repo root (writes gpurun_out/r05/*; copy into profiles/).
# Every rocprofv3 call profiles a single-process program under `timeout -k` with stdin closed; counters in their own passes.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export T4A_ROUND=r05
O=gpurun_out/r05
mkdir -p $O profiles
B="python3 bench.py --no-cpu-baseline --no-aux --no-floor"  # (--no-floor: no second GPU process under the profiler)
prof() { timeout -k 5 300 rocprofv3 "$@" </dev/null; }
timeout 900 python3 bench.py --steps 10 --warmup 3 2>$O/bench_n1.err | tail -1 > $O/bench_n1.json
cp $O/bench_n1.json profiles/r05_bench_n1.json
prof --kernel-trace --stats -d $O/stats -o x --output-format csv -- $B --steps 10 --warmup 3 > $O/stats.log 2>&1
cp $O/stats/x_kernel_stats.csv $O/bench_n1_kernel_stats.csv
python3 tools/trace_chain.py $O/stats/x_kernel_trace.csv 300 > $O/launch_gaps.txt 2>&1
python3 tools/trace_idle.py $O/stats/x_kernel_trace.csv 60 >> $O/launch_gaps.txt 2>&1
python3 tools/trace_timeline.py $O/stats/x_kernel_trace.csv 2300 160 > $O/chain_timeline.txt 2>&1
prof --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o x --output-format csv -- $B > $O/pmc_fetch.log 2>&1
prof --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o x --output-format csv -- $B > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.log 2>&1
prof --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA -d $O/pmc_mfma -o x --output-format csv -- $B > $O/pmc_mfma.log 2>&1
python3 tools/pmc_sq_summary.py $O/pmc_mfma $O/pmc_mfma_per_kernel.csv "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA -- $B" > /dev/null
python3 tools/mfma_summary.py $O/pmc_mfma_per_kernel.csv $O/bench_n1_kernel_stats.csv $O/mfma_kernels.json > /dev/null 2>&1
prof --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o x --output-format csv -- $B > $O/pmc_sq.log 2>&1
python3 tools/pmc_sq_summary.py $O/pmc_sq $O/pmc_sq_instruction_mix_per_kernel.csv "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $B" > /dev/null
cp profiles/r05_pmc_* $O/ 2>/dev/null
prof --kernel-trace -d $O/gemm -o x --output-format csv -- python3 tools/probe_gemm.py > $O/gemm.log 2>&1
python3 tools/gemm_trace_summary.py $O/gemm/x_kernel_trace.csv > $O/gemm_probe.txt 2>&1
timeout 300 python3 tools/probe_cfg5_group.py 8 > $O/cfg5_group_probe.txt 2>&1
timeout 300 python3 tools/probe_cfg5_threads.py 8 > $O/cfg5_threads_probe.txt 2>&1
timeout 300 python3 tools/probe_cfg5_scratch.py 3 > $O/cfg5_scratch_probe.txt 2>&1
timeout 300 python3 tools/probe_linalg.py > $O/linalg_probe.txt 2>&1
(timeout 120 ./tools/lat_bench; timeout 120 ./tools/hop_bench | grep pingpong) > $O/latencies.txt 2>&1
timeout 300 python3 tools/probe_fill.py > $O/fill_probe.txt 2>&1
prof --kernel-trace --stats -d $O/fillstats -o x --output-format csv -- python3 tools/probe_fill.py 30 > $O/fillstats.log 2>&1
grep "lu_panel\|lu_update\|lu_solve\|trsm\|pi_eval\|pack_fill" $O/fillstats/x_kernel_stats.csv > $O/fill_kernel_stats.csv
timeout 600 python3 bench.py --mode site-shard --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_site_shard_n1.json
T4A_XCD_V=1 T4A_NO_WG=1 timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux 2>/dev/null | tail -1 > $O/bench_n1_first_generation_kernels.json
if [ -f tensor4all-rs_amd/lib/libt4a_gpu_alt_w3.so ]; then
  T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_alt_w3.so T4A_RRLU_STAMPS=1 timeout 120 python3 tools/probe_xcd.py child 685 688 256 1 2>&1 | grep "stamps xcd" | tail -1 > $O/xcd_phase_stamps_wave3.txt
fi
if [ -f tensor4all-rs_amd/lib/libt4a_gpu_alt.so ]; then
  T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_alt.so T4A_RRLU_STAMPS=1 timeout 120 python3 tools/probe_xcd.py child 685 688 256 1 2>&1 | grep "stamps xcd" | tail -1 > $O/xcd_phase_stamps.txt
fi
timeout 600 python3 tools/probe_cfg4_variants.py > $O/cfg4_variants.txt 2>&1
timeout 300 python3 tools/probe_cfg5_variants.py > $O/cfg5_variants.txt 2>&1
timeout 1500 python3 tools/bench_components.py > $O/components.json 2> $O/components.err
if [ -f tensor4all-rs_amd/lib/libt4a_gpu_alt.so ]; then  # (tools/build_stamps_lib.sh)
for sh in "1464 1448 256" "1428 1024 256" "1424 512 256"; do
  T4A_GPU_LIB=$GRAFT_REPO_ROOT/tensor4all-rs_amd/lib/libt4a_gpu_alt.so T4A_RRLU_STAMPS=1 timeout 120 python3 tools/probe_xcd.py child $sh 1 2>&1 | grep "stamps xcd" | tail -1
done > $O/xcd2m_phase_stamps.txt
fi
# keep only the summaries (the merge back is limited to 64 MiB)
rm -rf $O/fillstats $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_sq $O/gemm
ls -la $O | head -40

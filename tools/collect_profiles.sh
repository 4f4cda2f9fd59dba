#!/bin/bash
# Round-2 profile artifacts, to be run on the GPU box from the repo root (writes gpurun_out/r02_*; copy into profiles/).
# Every rocprofv3 call profiles `python3 bench.py` itself (single process) under a timeout.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
timeout 600 python3 bench.py --steps 10 --warmup 3 2>$O/bench_n1.err | tail -1 > $O/bench_n1.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o x --output-format csv -- python3 bench.py --no-cpu-baseline --no-aux --steps 10 --warmup 3 > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o x --output-format csv -- python3 bench.py --no-cpu-baseline --no-aux > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o x --output-format csv -- python3 bench.py --no-cpu-baseline --no-aux > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA -d $O/pmc_mfma -o x --output-format csv -- python3 bench.py --no-cpu-baseline --no-aux > $O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o x --output-format csv -- python3 bench.py --no-cpu-baseline --no-aux > $O/pmc_sq.log 2>&1
timeout 120 tools/xcd_bench > $O/xcd_bench.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/gemm -o x --output-format csv -- python3 tools/probe_gemm.py > $O/gemm.log 2>&1
T4A_RRLU_STAMPS=1 timeout 120 python3 tools/probe_xcd.py child 685 688 256 1 2>&1 | grep "stamps xcd" | tail -1 > $O/xcd_phase_stamps.txt
ls -R $O | head -60

#!/bin/bash
# round-4 GPU session: split-K GEMM — tests that go through the GEMM, then device times of the probe shapes (kernel trace)
O=gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=tensor4all-rs_amd/python
timeout 900 python3 -m pytest tests/test_gpu_dense.py tests/test_gpu_tt.py tests/test_gpu_tensor.py -x -q 2>&1 | tail -3
timeout -k 5 300 rocprofv3 --kernel-trace -d $O/gemm -o x --output-format csv -- python3 tools/probe_gemm.py > $O/gemm.log 2>&1 </dev/null
python3 tools/gemm_trace_summary.py $O/gemm/x_kernel_trace.csv > $O/gemm_probe.txt 2>&1
T4A_GEMM_KSPLIT=1 timeout -k 5 300 rocprofv3 --kernel-trace -d $O/gemm1 -o x --output-format csv -- python3 tools/probe_gemm.py > $O/gemm1.log 2>&1 </dev/null
python3 tools/gemm_trace_summary.py $O/gemm1/x_kernel_trace.csv > $O/gemm_probe_nosplit.txt 2>&1
rm -rf $O/gemm $O/gemm1
cat $O/gemm.log | grep err; cat $O/gemm_probe.txt; echo ---; cat $O/gemm_probe_nosplit.txt

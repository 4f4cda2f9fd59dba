#!/bin/bash
# usage (on the GPU box): tools/r6_svd.sh LABEL — wall time, residuals and sweep counts of svd_backend at small shapes, the Jacobi kernels'
# durations from a kernel trace, and the SVD / compression tests.  Output under gpurun_out/r6_svd/.
L=${1:-run}
out=gpurun_out/r6_svd
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
T4A_SVD_DEBUG=1 timeout 200 python3 tools/probe_svd_small.py default > $out/${L}.txt 2> $out/${L}_sweeps.err
grep "^\[default\]" $out/${L}.txt | cut -c1-200
grep "one launch" $out/${L}_sweeps.err | sort | uniq -c
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $out/raw -o x --output-format csv -- python3 tools/probe_svd_small.py default > $out/prof.log 2>&1 </dev/null
python3 tools/svd_trace_summary.py $out/raw/x_kernel_trace.csv | tee $out/${L}_kernels.txt
rm -rf $out/raw
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_tt.py tests/test_gpu_tensor.py -x -q -k "svd or compress or qr" 2>&1 | tail -3

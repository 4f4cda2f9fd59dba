#!/bin/bash
O=gpurun_out/svd; mkdir -p $O
timeout 600 python tools/probe_linalg.py 2>&1 | tee $O/linalg_probe.txt
timeout 900 python -m pytest tests/test_gpu_tt.py tests/test_gpu_tensor.py tests/test_gpu_dense.py -q -x 2>&1 | tail -5
timeout 900 python tools/bench_components.py --only tt,dense 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('tt','dense') if k in d})[:2500])"

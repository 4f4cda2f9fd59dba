#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/probe_linalg.py 2>&1 | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_tt.py tests/test_gpu_dense.py tests/test_gpu_tensor.py -x -q -m gpu 2>&1 | tail -4

#!/bin/bash
# round-5: kernels beyond one XCD — fuzz parity, stamps, cfg4-size sweep by kernel (short session), then the component benchmarks
O=gpurun_out/$1; mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
L=$PWD/tensor4all-rs_amd/lib
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_rrlu_global.py -m gpu -x -q -k "beyond_one_xcd or widest or forced" > $O/pytest_fuzz.log 2>&1; echo "rc=$?" >> $O/pytest_fuzz.log; tail -n 4 $O/pytest_fuzz.log
timeout 900 python -m pytest tests/test_gpu_tci2.py -m gpu -x -q -k "cfg4 or raw_cores" > $O/pytest_cfg4.log 2>&1; echo "rc=$?" >> $O/pytest_cfg4.log; tail -n 4 $O/pytest_cfg4.log
for sh in "1464 1448 256" "1428 1024 256"; do
  T4A_RRLU_STAMPS=1 T4A_GPU_LIB=$L/libt4a_gpu_alt.so timeout 120 python tools/probe_xcd.py child $sh 1 2>&1 | grep stamps | tail -n 1
done | tee $O/stamps.txt
timeout 600 python tools/probe_cfg4_variants.py 2>&1 | head -8 | tee $O/cfg4_variants.txt
timeout 1500 python tools/bench_components.py > $O/components.json 2> $O/components.err
python - "$O/components.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, json.dumps(v)[:900])
PY
tail -n 5 $O/components.err

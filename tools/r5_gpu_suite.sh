#!/bin/bash
# full GPU suite + default bench line.   usage: tools/r5_gpu_suite.sh OUTDIR
O=gpurun_out/$1
mkdir -p $O
export PYTHONPATH=tensor4all-rs_amd/python
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|error|Error" $O/pytest.log | tail -8
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - "$O/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); a = d.get("aux", {})
print(round(d["ms_per_step"], 3), d["value"], d["roofline"]["frac"], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in a.items() if "ms" in k})
PY

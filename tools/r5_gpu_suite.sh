#!/bin/bash
# whole GPU suite + the default bench line
O=gpurun_out/suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<'P'
import json
d = json.loads(open("gpurun_out/suite/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "fill", d["breakdown_ms_per_sweep"], "roofline.frac", d["roofline"]["frac"])
a = d.get("aux", {})
print({k: v for k, v in a.items() if k != "components"})
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
P

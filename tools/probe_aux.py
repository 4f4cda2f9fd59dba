"""bench.py's aux block alone (cfg2 / cfg5 / cfg5 concurrent / cfg4 timings on one GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    print(json.dumps(bench.aux_timings()), flush=True)

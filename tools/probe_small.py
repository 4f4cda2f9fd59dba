"""Per-pivot-step time of the rrLU kernel on small square matrices (the TreeTCI / quantics regime), single- vs
multi-workgroup plans: run under `rocprofv3 --kernel-trace --output-format csv` with T4A_RRLU_SINGLE_MAX=<elements>, then
`probe_small.py trace.csv`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (16, 32, 48, 64, 96, 128, 160, 192, 256)
REP = 8
if len(sys.argv) > 1:
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rrlu" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) == len(SIZES) * REP, len(rows)
    for i, n in enumerate(SIZES):
        d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i * REP:(i + 1) * REP]][2:])
        name = rows[i * REP]["Kernel_Name"].split("rrlu_")[1][:28]
        print(f"n={n:4d}: {d.mean():8.1f} us per launch = {d.mean() / n:5.2f} us per pivot step  [{name} grid {rows[i * REP]['Grid_Size_X']}]")
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import t4a_amd
rng = np.random.default_rng(0)
for n in SIZES:
    a = rng.standard_normal((n, n))
    for _ in range(REP):
        t4a_amd.rrlu(a, rel_tol=0.0)

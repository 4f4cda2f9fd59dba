"""Fixed (step-independent) cost of one rrLU launch at the mid-chain shape: run with max_bond_dim = 1, 9, 65, 256 under
`rocprofv3 --kernel-trace --output-format csv` and read the per-dispatch durations.  Usage: probe_fixed.py [trace.csv]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rrlu_reg_kernel" in r["Kernel_Name"]]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    caps = [1, 9, 65, 256]
    per = len(durs) // len(caps)
    for i, c in enumerate(caps):
        d = np.array(durs[i * per:(i + 1) * per][2:])
        print(f"max_bond_dim {c:4d}: {d.mean():8.1f} us per launch (min {d.min():.1f})")
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
import t4a_amd
rng = np.random.default_rng(0)
a = rng.standard_normal((685, 688))
for cap in (1, 9, 65, 256):
    for _ in range(12):
        t4a_amd.rrlu(a, max_bond_dim=cap, rel_tol=0.0)

"""Summarise the Jacobi kernels of a `rocprofv3 --kernel-trace --output-format csv` run of tools/probe_svd_small.py: duration per launch
geometry (workgroup size identifies the shape).  usage: python3 tools/svd_trace_summary.py DIR/x_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "jacobi" not in name and "svd" not in name and "qr_" not in name:
        continue
    import re
    mm = re.search(r"(jacobi_\w+|svd_\w+|qr_\w+)", name)
    short = mm.group(1) if mm else name[:40]
    d[(short, r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print(f"{k[0]:32s} wg {k[1]:>5s} lds {k[2]:>7s}: {len(v):4d} launches, min {min(v):8.1f} us, median {sorted(v)[len(v) // 2]:8.1f} us")

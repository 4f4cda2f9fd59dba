"""Per launch geometry of jacobi_groups_kernel: the averages of the counters of one or more `rocprofv3 --kernel-trace --pmc ... --output-format
csv` passes over tools/probe_svd_pmc.py, next to the launch duration.  usage: svd_pmc_summary.py DIR [DIR ...]"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if "jacobi_groups_kernel" not in r["Kernel_Name"]:
            continue
        key = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "End_Timestamp" in r and r["Counter_Name"]:
            dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for key in sorted(acc, key=lambda k: int(k) if k.isdigit() else 0):
    line = f"workgroup {key:>5s}:"
    for c, v in sorted(acc[key].items()):
        line += f"  {c} {sum(v) / len(v):12.0f} (n={len(v)})"
    if dur[key]:
        line += f"  duration under the profiler {sum(dur[key]) / len(dur[key]) / 1e3:8.1f} us"
    print(line)

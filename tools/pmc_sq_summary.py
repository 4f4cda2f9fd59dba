"""Per-kernel averages of a `rocprofv3 --kernel-trace --pmc <counters> --output-format csv` pass of bench.py.
Usage: pmc_sq_summary.py <pass_dir> <out.csv> "<command line for the header>"   (sums over all waves per dispatch)"""
import csv
import glob
import os
import sys
from collections import defaultdict

f = max(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
names = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
    names.add(r["Counter_Name"])
names = sorted(names)
rows = sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get(names[0], 0.0)))
with open(sys.argv[2], "w") as out:
    out.write("# " + (sys.argv[3] if len(sys.argv) > 3 else "") + "; averages per dispatch (sums over all waves; SQ_*_CYCLES in quad-cycles)\n")
    w = csv.writer(out)
    w.writerow(["Kernel_Name", "Dispatches"] + [n + "_per_dispatch" for n in names])
    for k in rows:
        n = len(cnt[k])
        w.writerow([k, n] + ["%.1f" % (acc[k][c] / n) for c in names])
print(open(sys.argv[2]).read()[:1500])

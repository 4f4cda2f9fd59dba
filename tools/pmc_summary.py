"""Summarise two `rocprofv3 --kernel-trace --pmc X --output-format csv` passes (X = FETCH_SIZE, WRITE_SIZE) of bench.py into
profiles/<round>_pmc_fetch_write_per_kernel.csv and profiles/<round>_pmc_dominant_kernel.json (round from T4A_ROUND, default r02).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes,
MI355X_MICROARCH.md, HBM section).   Usage: pmc_summary.py <fetch_dir> <write_dir>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("T4A_ROUND", "r02")


def per_kernel(d, counter):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)  # newest
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return {k: (v[0] / v[1], v[1], v[2]) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    fk, n, dur = fetch.get(k, (0.0, 0, 0.0))
    wk, _, _ = write.get(k, (0.0, 0, 0.0))
    rows.append((k, n, fk, wk, (2.0 * fk + wk) * 1024.0, dur))
rows.sort(key=lambda r: -r[5])
with open(os.path.join(ROOT, "profiles", ROUND + "_pmc_fetch_write_per_kernel.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Dispatches", "FETCH_SIZE_avg_KB", "WRITE_SIZE_avg_KB", "hbm_bytes_per_launch(2*FETCH+WRITE)"])
    for r in rows:
        w.writerow(r[:5])
# the kernel instantiation bench.py reported as dominant (profiles/r01_bench_n1.json), or argv[3]
want = sys.argv[3] if len(sys.argv) > 3 else json.load(open(os.path.join(ROOT, "profiles", ROUND + "_bench_n1.json")))["roofline"]["kernel"]
want = want.split(" (")[0].replace("t4a::", "").strip()
dom = next(r for r in rows if want in r[0])
json.dump({"kernel": dom[0],
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline "
                      "(default --steps 5 --warmup 2; two separate passes), summarised by tools/pmc_summary.py",
           "FETCH_SIZE_avg_KB": dom[2], "WRITE_SIZE_avg_KB": dom[3], "hbm_bytes_per_launch": dom[4],
           "correction": "gfx950: FETCH_SIZE counts 128-B read requests at 64 B -> doubled (MI355X_MICROARCH.md §HBM); units are KiB",
           "dispatches": dom[1],
           # every rrLU register-kernel instantiation of the run (the two tie orders of a shape take turns as the dominant one)
           "rrlu_variants": {r[0]: r[4] for r in rows if "rrlu_" in r[0]}}, open(os.path.join(ROOT, "profiles", ROUND + "_pmc_dominant_kernel.json"), "w"), indent=1)
print(dom)

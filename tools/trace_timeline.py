#!/usr/bin/env python3
"""Compact timeline of a window of a `rocprofv3 --kernel-trace` CSV: start (us, relative), duration, queue, short kernel name.
Usage: python tools/trace_timeline.py <kernel_trace.csv> <first_kernel_index> <count>"""
import csv
import sys


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "t4a::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n[:44]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        rd = csv.DictReader(f)
        for r in rd:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"],
                         r.get("Workgroup_Size_X", "?"), r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "1")))
    rows.sort()
    a, n = int(sys.argv[2]), int(sys.argv[3])
    t0 = rows[a][0]
    for s, e, q, name, wg, grid in rows[a:a + n]:
        print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  q{q:>3} grid {grid:>10}  {short(name)}")


if __name__ == "__main__":
    main()

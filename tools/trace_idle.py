#!/usr/bin/env python3
"""GPU idle periods in a `rocprofv3 --kernel-trace` CSV: busy time (union over all queues), and the longest gaps with the kernels
around them.  Usage: python tools/trace_idle.py <kernel_trace.csv> [min_gap_us] [t_from_ms] [t_to_ms]"""
import csv
import sys


def short(name):
    n = name
    for p in ("void ", "t4a::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n.split("(")[0][:40]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 50e3
    t0 = rows[0][0]
    lo = float(sys.argv[3]) * 1e6 + t0 if len(sys.argv) > 3 else rows[0][0]
    hi = float(sys.argv[4]) * 1e6 + t0 if len(sys.argv) > 4 else rows[-1][1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    busy, cur_end, gaps = 0, rows[0][0], []
    prev = None
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, cur_end, prev, n))
            busy += e - s
            cur_end = e
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
        prev = n if e >= cur_end else prev
    span = rows[-1][1] - rows[0][0]
    print(f"{len(rows)} kernels over {span/1e6:.2f} ms: busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps")
    hist = {}
    for g, *_ in gaps:
        k = "<5us" if g < 5e3 else "<10us" if g < 10e3 else "<20us" if g < 20e3 else "<50us" if g < 50e3 else "<200us" if g < 200e3 else ">=200us"
        hist.setdefault(k, [0, 0])
        hist[k][0] += 1
        hist[k][1] += g
    for k in ("<5us", "<10us", "<20us", "<50us", "<200us", ">=200us"):
        if k in hist:
            print(f"   gaps {k:8s}: {hist[k][0]:5d}  total {hist[k][1]/1e6:7.2f} ms")
    for g, at, a, b in sorted(gaps, reverse=True)[:25]:
        if g >= min_gap:
            print(f"   {g/1e3:8.1f} us idle at {(at-rows[0][0])/1e6:8.3f} ms   after {short(a or '')}  before {short(b)}")


if __name__ == "__main__":
    main()

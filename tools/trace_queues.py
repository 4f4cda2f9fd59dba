#!/usr/bin/env python3
"""Per-queue view of a `rocprofv3 --kernel-trace` CSV (concurrent handles): for every queue the busy time, the time in rrLU
kernels, their mean duration, and the idle time between its kernels.  Usage: trace_queues.py <kernel_trace.csv> [t_from_ms] [t_to_ms]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
    rows.sort()
    t0 = rows[0][0]
    lo = float(sys.argv[2]) * 1e6 + t0 if len(sys.argv) > 2 else rows[0][0]
    hi = float(sys.argv[3]) * 1e6 + t0 if len(sys.argv) > 3 else rows[-1][1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    q = defaultdict(list)
    for r in rows:
        q[r[2]].append(r)
    print(f"window {(hi-lo)/1e6:.1f} ms, {len(rows)} kernels, {len(q)} queues")
    for k, v in sorted(q.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _, _ in v)
        rr = [(e - s) for s, e, _, n in v if "rrlu_" in n]
        span = v[-1][1] - v[0][0]
        gaps = sum(max(0, v[i + 1][0] - v[i][1]) for i in range(len(v) - 1))
        names = defaultdict(lambda: [0, 0])
        for s, e, _, n in v:
            nn = n.replace("void ", "").replace("t4a::", "").replace("(anonymous namespace)::", "").split("(")[0][:28]
            names[nn][0] += 1
            names[nn][1] += e - s
        top = sorted(names.items(), key=lambda kv: -kv[1][1])[:4]
        print(f"  queue {k:>3}: {len(v):5d} kernels, span {span/1e6:7.2f} ms, busy {busy/1e6:7.2f} ms, gaps {gaps/1e6:7.2f} ms, rrLU {len(rr):4d} x "
              f"{(sum(rr)/max(len(rr),1))/1e3:7.1f} us = {sum(rr)/1e6:6.2f} ms | " + ", ".join(f"{n} {c}x{t/c/1e3:.0f}us" for n, (c, t) in top))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Timeline of the bond chain from a `rocprofv3 --kernel-trace` CSV: for every rrLU launch, what ran between the end of the
previous rrLU kernel and its start (per kernel name: count, mean duration) and how much of that gap was idle.
Usage: python tools/trace_chain.py <kernel_trace.csv> [skip_first_n_rrlu]"""
import csv
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "t4a::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n[:60]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rr = [i for i, r in enumerate(rows) if "rrlu_" in r[2]][skip:]
    gaps, idle = [], []
    per = defaultdict(lambda: [0, 0.0])
    for a, b in zip(rr[:-1], rr[1:]):
        end_a, start_b = rows[a][1], rows[b][0]
        gap = start_b - end_a
        if gap > 2_000_000:
            continue
        busy = 0
        for j in range(a + 1, b):
            s, e = max(rows[j][0], end_a), min(rows[j][1], start_b)
            if e > s:
                busy += e - s
                k = per[short(rows[j][2])]
                k[0] += 1
                k[1] += rows[j][1] - rows[j][0]
        gaps.append(gap)
        idle.append(max(gap - busy, 0))
    n = len(gaps)
    if not n:
        print("no rrLU launches")
        return
    gs = sorted(gaps)
    print(f"{n} consecutive rrLU launches: gap mean {sum(gaps)/n/1e3:.1f} us (median {gs[n//2]/1e3:.1f}, p90 {gs[9*n//10]/1e3:.1f}), "
          f"idle inside the gap mean {sum(idle)/n/1e3:.1f} us; total gap time {sum(gaps)/1e6:.2f} ms")
    for name, (cnt, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:60s} x{cnt:5d}  mean {tot/cnt/1e3:7.1f} us  per rrLU gap {tot/n/1e3:6.1f} us")


if __name__ == "__main__":
    main()

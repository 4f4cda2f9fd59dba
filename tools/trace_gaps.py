#!/usr/bin/env python3
"""Idle gaps of the GPU between consecutive rrLU launches of the bench sweep, from a `rocprofv3 --kernel-trace` CSV
(`x_kernel_trace.csv`): for every rrLU kernel the time since the previous rrLU kernel ended, split into kernels that ran in
between on any stream and true idle time.  Usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n_rrlu]"""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    is_rrlu = lambda n: "rrlu_" in n
    rr = [i for i, r in enumerate(rows) if is_rrlu(r[2])]
    rr = rr[skip:]
    gaps, between, dur = [], [], []
    for a, b in zip(rr[:-1], rr[1:]):
        end_a, start_b = rows[a][1], rows[b][0]
        gap = start_b - end_a
        if gap > 2_000_000:  # a new bench phase
            continue
        busy = 0
        for j in range(a + 1, b):
            s, e = max(rows[j][0], end_a), min(rows[j][1], start_b)
            if e > s:
                busy += e - s
        gaps.append(gap)
        between.append(busy)
        dur.append(rows[b][1] - rows[b][0])
    n = len(gaps)
    if not n:
        print("no rrLU launches found")
        return
    gaps_s = sorted(gaps)
    print(f"{n} consecutive rrLU launches: mean kernel {sum(dur)/n/1e3:.1f} us; gap between them mean {sum(gaps)/n/1e3:.1f} us "
          f"(median {gaps_s[n//2]/1e3:.1f}, p10 {gaps_s[n//10]/1e3:.1f}, p90 {gaps_s[9*n//10]/1e3:.1f}); "
          f"of which other kernels in the gap {sum(between)/n/1e3:.1f} us")


if __name__ == "__main__":
    main()

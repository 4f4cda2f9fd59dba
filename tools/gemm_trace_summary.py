"""Per-call device time of gemm_kernel from a `rocprofv3 --kernel-trace` CSV of tools/probe_gemm.py (three calls per shape, the
shapes in probe order): prints the best time and the fraction of the fp64 MFMA peak (78.6 TF/s).
Usage: gemm_trace_summary.py <kernel_trace.csv>"""
import csv
import sys

SHAPES = [(1024, 1024, 1024), (256, 512, 256), (512, 1400, 256), (2048, 2048, 2048)]
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_kernel" in r["Kernel_Name"]:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for i, (m, k, n) in enumerate(SHAPES):
    calls = rows[3 * i:3 * i + 3]
    if not calls:
        break
    best = min(c[1] for c in calls)
    print(f"m={m} k={k} n={n}: {best / 1e3:.1f} us  {2.0 * m * k * n / best / 1e3:.2f} TF/s  {2.0 * m * k * n / best / 1e3 / 78.6 * 100:.1f} % of fp64 MFMA peak  [{calls[0][2].split('(')[0][-40:]}]")

"""Per-call device time of the GEMM from a `rocprofv3 --kernel-trace` CSV of tools/probe_gemm.py (three calls per shape, the
shapes in probe order): gemm_kernel plus, for split-K products, the reduction pass behind it (start of the first to end of the
second launch, i.e. including the boundary between them).  Prints the best time and the fraction of the fp64 MFMA peak (78.6 TF/s).
Usage: gemm_trace_summary.py <kernel_trace.csv>"""
import csv
import sys

SHAPES = [(1024, 1024, 1024), (256, 512, 256), (512, 1400, 256), (2048, 2048, 2048)]
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "gemm_kernel" in name or "gemm_splitk_reduce_kernel" in name:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "reduce" if "splitk_reduce" in name else "gemm"))
rows.sort()
calls = []  # (start, end, n_launches)
for s, e, kind in rows:
    if kind == "reduce" and calls:
        calls[-1] = (calls[-1][0], e, calls[-1][2] + 1)
    else:
        calls.append((s, e, 1))
for i, (m, k, n) in enumerate(SHAPES):
    mine = calls[3 * i:3 * i + 3]
    if not mine:
        break
    best = min(c[1] - c[0] for c in mine)
    print(f"m={m} k={k} n={n}: {best / 1e3:.1f} us  {2.0 * m * k * n / best / 1e3:.2f} TF/s  {2.0 * m * k * n / best / 1e3 / 78.6 * 100:.1f} % of fp64 MFMA peak"
          f"  [{'gemm_kernel + split-K reduction' if mine[0][2] > 1 else 'gemm_kernel'}]")

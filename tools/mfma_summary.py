#!/usr/bin/env python3
"""MFMA view of the kernels that use the f64 matrix cores (fill_site_tensors' LU trailing update and triangular solve, the GEMM):
combines the per-kernel PMC averages of a `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 ...` pass (tools/pmc_sq_summary.py CSV)
with the kernel durations of a `rocprofv3 --kernel-trace --stats` pass of the SAME command, and writes the JSON bench.py puts
into its line as roofline.mfma.
  flops per launch = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 (one count = 512 flop, MI355X_MICROARCH.md / VERDICT round 2)
Usage: mfma_summary.py <pmc_mfma_per_kernel.csv> <kernel_stats.csv> <out.json>"""
import csv
import json
import sys

PEAK = 78.6      # TF/s, dense f64 MFMA (MI355X_MICROARCH.md)
SUSTAINED = 47.8  # TF/s, measured with tools/mfma_peak.hip (profiles/r02_mfma_peak.log)


def main():
    pmc, stats, out = sys.argv[1:4]
    dur = {}
    with open(stats) as f:
        for r in csv.DictReader(f):
            dur[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
    rows = {}
    with open(pmc) as f:
        lines = [ln for ln in f if not ln.startswith("#")]
    for r in csv.DictReader(lines):
        mops = float(r.get("SQ_INSTS_VALU_MFMA_MOPS_F64_per_dispatch", 0.0) or 0.0)
        if mops <= 0.0:
            continue
        name = r["Kernel_Name"]
        avg_ns, calls = dur.get(name, (0.0, 0))
        flops = mops * 512.0
        tf = flops / avg_ns / 1e3 if avg_ns > 0 else None
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("t4a::", "").split("(")[0]
        rows[short] = {"kernel": name, "launches": calls, "avg_us": avg_ns / 1e3, "mfma_flops_per_launch": flops,
                       "mfma_instructions_per_launch": float(r.get("SQ_INSTS_MFMA_per_dispatch", 0.0) or 0.0),
                       "tflops": tf, "frac_of_peak": tf / PEAK if tf else None, "frac_of_sustained": tf / SUSTAINED if tf else None}
    json.dump({"peak_tflops": PEAK, "sustained_tflops": SUSTAINED, "kernels": rows,
               "source": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 (x 512 flop) per dispatch and rocprofv3 --kernel-trace --stats average "
                         "durations of `python3 bench.py --no-cpu-baseline --no-aux`"}, open(out, "w"), indent=1)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()

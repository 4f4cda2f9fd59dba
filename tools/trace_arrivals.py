#!/usr/bin/env python3
"""Arrival skew of the per-pivot key exchange (diagnostic build only).

    T4A_EXTRA_FLAGS=-DT4A_RRLU_TRACE python tensor4all-rs_amd/build.py      # here, then on the GPU box:
    T4A_RRLU_TRACE_FILE=gpurun_out/trace.bin python bench.py --no-cpu-baseline --steps 2 --warmup 1
    python tools/trace_arrivals.py gpurun_out/trace.bin

Every workgroup stamps the 100 MHz real-time clock when it publishes its key (t_pub) and when its key gather has finished
(t_done).  Reported per traced launch: the spread of t_pub over the workgroups per step, the lag of the last arrival behind
the median, gather time after the last arrival, and which workgroups / XCCs are the late ones."""
import sys

import numpy as np


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    pos, launch = 0, 0
    while pos < len(raw):
        M, N, W, steps = (int(x) for x in raw[pos:pos + 4].astype(np.int64))
        pos += 4
        words = W * (1 + 4 * (steps + 1))
        blk = raw[pos:pos + words]
        pos += words
        xcc = blk[:W].astype(int)
        t = blk[W:].reshape(steps + 1, W, 4).astype(np.float64) * 10.0  # ns: publish, gather done, column in registers, pass done
        valid = [k for k in range(1, steps - 1) if (t[k:k + 2] > 0).all()]
        if not valid:
            continue
        pub, done = t[valid, :, 0], t[valid, :, 1]
        first, med, last = pub.min(axis=1), np.median(pub, axis=1), pub.max(axis=1)
        step_ns = np.diff(last).mean()
        late = np.argmax(pub, axis=1)
        print(f"launch {launch}: {M}x{N} W={W} steps traced={len(valid)}  step {step_ns:.0f} ns")
        print(f"  publish spread last-first {np.mean(last - first):.0f} ns (p90 {np.percentile(last - first, 90):.0f}), "
              f"last-median {np.mean(last - med):.0f} ns")
        print(f"  gather done after the last publish: median over WGs {np.mean(np.median(done, axis=1) - last):.0f} ns, "
              f"slowest WG {np.mean(done.max(axis=1) - last):.0f} ns")
        nxt_first = first[1:] - done.max(axis=1)[:-1]
        print(f"  slowest gather done -> first publish of the next step {np.mean(nxt_first):.0f} ns; "
              f"median gather done -> median publish {np.mean(med[1:] - np.median(done, axis=1)[:-1]):.0f} ns")
        # median chain of one step: gather done (kn) -> column in registers (kn) -> pass done (stamped under kn + 1) -> publish (kn + 1)
        v = np.array(valid)
        md = lambda a: np.median(a, axis=1)
        g2c = md(t[v, :, 2]) - md(t[v, :, 1])
        c2p = md(t[v + 1, :, 3]) - md(t[v, :, 2])
        p2k = md(t[v + 1, :, 0]) - md(t[v + 1, :, 3])
        k2g = md(t[v + 1, :, 1]) - md(t[v + 1, :, 0])
        print(f"  median chain: gather done -> column in registers {g2c.mean():.0f} ns -> update pass done {c2p.mean():.0f} ns -> "
              f"key published {p2k.mean():.0f} ns -> gather done {k2g.mean():.0f} ns")
        print(f"  wave-0 spread inside a step: column arrival last-median {np.mean(t[v, :, 2].max(axis=1) - md(t[v, :, 2])):.0f} ns")
        cnt = np.bincount(late, minlength=W)
        top = np.argsort(-cnt)[:8]
        print("  most often last to publish: " + ", ".join(f"wg{w}(xcc{xcc[w]}):{cnt[w]}" for w in top))
        by_xcc = np.bincount(xcc[late], minlength=8)
        print(f"  last publisher by XCC: {by_xcc.tolist()}  (workgroups per XCC: {np.bincount(xcc, minlength=8).tolist()})")
        lag = (pub - med[:, None]).mean(axis=0)
        print(f"  mean lag behind the median publish per XCC: {[round(float(lag[xcc == x].mean())) if (xcc == x).any() else None for x in range(8)]} ns")
        launch += 1


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — TCI2 full-sweep benchmark (BASELINE.json metric) on N MI355X of one node.

One "step" = one FULL TCI2 sweep (forward half-sweep + backward half-sweep, each = update_pivots over all
bonds + fill_site_tensors, i.e. two iterations of optimize_with_finder, tensorci2.rs:1659-1776) at saturated
rank, for the configuration the metric is quoted on: d = 30 binary sites (interleaved quantics of a 2-variable
oscillatory integrand), chi_max = 256, fp64, tolerance 1e-12, nsearch = 0 (BASELINE.json configs[2]; configs[1]
— cos(10x)exp(-x) at d=20 — has exact TT rank 2 and is a parity case, not a bench line).

N = 1 : the cfg3 sweep itself.
N > 1 : weak scaling over PartitionedTT-style patches (BASELINE.json configs[4]): rank p
        interpolates the patch of the same integrand whose leading log2(N) bits are fixed to p (30 active sites,
        chi_max 256 per patch — per-GPU work is fixed), and after every sweep the patch cores are all-gathered
        over RCCL/xGMI (the only exchange step this path has).  The update_pivots chain of one TCI does not
        shard (bond b+1 needs the pivots of bond b), so there is no data-path collective inside a sweep.

Prints ONE JSON line on rank 0 (see the contract in the task description).  Inputs are generated on the device
(built-in functor), so the timed region starts with everything resident in HBM.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tensor4all-rs_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
N_SITES = 30
CHI = 256
# chosen so that the converged link dimensions are exactly min(2^b, 2^(d-b), 256) (tools/probe_rank.py)
K1, K2, K3, EPS, K4, DELTA = 37, 53, 2111, 0.5, 16411, 0.5


def patch_spec(rank, world):
    """Built-in function of patch `rank` out of `world` (= 2^b) patches: the leading b bits (alternating x, y)
    of a (30 + b)-bit interleaved quantics grid are fixed; the 30 remaining bits are the active sites."""
    import numpy as np
    from t4a_amd.functions import FnSpec, FN_QUANTICS_OSC2D
    b = int(round(math.log2(world))) if world > 1 else 0
    bx, by = (b + 1) // 2, b // 2  # extra MSBs of x and y
    nbx, nby = N_SITES // 2 + bx, N_SITES // 2 + by
    # prefix bits of this patch: bit j of `rank` (MSB first) goes to x if j even else y
    px = py = 0
    for j in range(b):
        bit = (rank >> (b - 1 - j)) & 1
        if j % 2 == 0:
            px = (px << 1) | bit
        else:
            py = (py << 1) | bit
    w = np.zeros((2, 2 * N_SITES), dtype=np.uint64)
    # active site s: after the b prefix bits the interleaving continues; site s is global bit position b + s
    cx, cy = bx, by  # how many bits of x / y are already consumed
    for s in range(N_SITES):
        g = b + s
        if g % 2 == 0:
            w[0, 2 * s + 1] = np.uint64(1) << np.uint64(nbx - 1 - cx)
            cx += 1
        else:
            w[1, 2 * s + 1] = np.uint64(1) << np.uint64(nby - 1 - cy)
            cy += 1
    assert cx == nbx and cy == nby
    # fold the constant prefix contribution into BOTH entries of site 0 (the accumulators are plain sums)
    offx = np.uint64(px) << np.uint64(nbx - bx) if bx else np.uint64(0)
    offy = np.uint64(py) << np.uint64(nby - by) if by else np.uint64(0)
    w[0, 0] += offx
    w[0, 1] += offx
    w[1, 0] += offy
    w[1, 1] += offy
    return FnSpec(FN_QUANTICS_OSC2D, [K1, K2, K3, EPS, K4, DELTA, nbx, nby], w, [2] * N_SITES)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run with N ranks
    on this node (one process per GPU, rendezvous on 127.0.0.1).  Called before anything has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the cfg2 / cfg4 / cfg5 single-GPU timings of the aux block")
    ap.add_argument("--no-components", action="store_true",
                    help="skip aux.components (tools/bench_components.py: callback path, rook, tree, quantics, ACI, tensor-train utilities, "
                         "dense kernels, each with the CPU oracle's time beside it; about a minute)")
    ap.add_argument("--no-floor", action="store_true",
                    help="do not start tools/xcd_bench (a second GPU process) for the exchange floor: use the committed constant — for "
                         "runs under rocprofv3, whose preload and counters a child process would inherit")
    ap.add_argument("--mode", choices=["headline", "site-shard", "patch-farm", "pi-shard"], default="headline",
                    help="headline: BASELINE.json configs[2] (N = 1) / patch farm (N > 1).  site-shard: configs[3] — d = 40, chi = 512, "
                         "bond chain replicated on every rank, fill_site_tensors sharded by site, one device-resident core "
                         "all-gather per half-sweep overlapped with the next half-sweep.  patch-farm: configs[4] as stated — 64 statically "
                         "projected patches, chi = 128, farmed over the ranks (eight at a time per GPU), ONE all_gather_into_tensor of the "
                         "padded patch cores; strong scaling.  pi-shard: configs[2] through a native HOST callback, every candidate matrix "
                         "evaluated by column blocks over the ranks and all-gathered, rrLU replicated (SURVEY.md 8e row 2)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: start the N ranks ourselves (fresh child processes, nothing has touched the GPU yet)
        # and leave with the launcher's exit code
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")

    import numpy as np
    import torch
    import torch.distributed as dist
    import t4a_amd

    if not torch.cuda.is_available() or t4a_amd.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the TCI2 backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    t4a_amd.set_device(local_rank)
    # torch's work of this process (collectives, the copies around them) runs on a non-blocking side stream, never on stream 0: the
    # library exchanges cores with torch through events, and the legacy default stream's implicit ordering broke the replay of
    # the captured fill graph (profiles/r05_fill_graph_fault_bisect.txt)
    from t4a_amd import parallel as _par
    _par.leave_legacy_stream(torch)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the RCCL process group has {dist.get_world_size()} ranks")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode in ("site-shard", "patch-farm", "pi-shard"):
        {"site-shard": site_shard_mode, "patch-farm": patch_farm_mode, "pi-shard": pi_shard_mode}[args.mode](args, world, rank, dist, torch, t4a_amd, barrier)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    spec = patch_spec(rank, world)
    tci = t4a_amd.TensorCI2([2] * N_SITES)
    tci.set_function(spec)
    tci.add_global_pivots([[0] * N_SITES])
    tci.set_max_sample_value(1.0)
    tci.set_keep_site_tensors(True)  # keep the cores of the last fill_site_tensors for the patch-core gather

    def opts(iters):
        return t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=CHI, max_iter=iters, ncheck_history=10 ** 6,
                                   nsearch=0, max_nglobal_pivot=0, seed=42)

    # all-gather buffers for the patch cores (N > 1): every core padded to the cap shape (chi, 2, chi); two buffer pairs
    # so that the gather of sweep k overlaps with the bond updates of sweep k+1
    core_cap = CHI * 2 * CHI
    if world > 1:
        send = [torch.zeros(N_SITES * core_cap, dtype=torch.float64, device="cuda") for _ in range(2)]
        recv = [torch.zeros(world * N_SITES * core_cap, dtype=torch.float64, device="cuda") for _ in range(2)]
        done = [None, None]   # torch events: the all-gather that last used buffer pair k has finished
        n_gathers = [0]

    def gather_cores():
        """RCCL all-gather of this rank's patch cores, issued without stalling the host: the device-to-device export
        is ordered after the fill_site_tensors still in flight and the collective waits for it on the device."""
        if world == 1:
            return
        k = n_gathers[0] % 2
        n_gathers[0] += 1
        if done[k] is not None:
            done[k].synchronize()          # finished a full sweep ago: no stall, just the buffer-reuse guarantee
        stream = torch.cuda.current_stream()
        tci.export_site_tensors_async(send[k].data_ptr(), core_cap, stream.cuda_stream)
        work = dist.all_gather_into_tensor(recv[k], send[k], async_op=True)
        work.wait()                        # stream-level dependency only
        ev = torch.cuda.Event()
        ev.record(stream)
        done[k] = ev

    def full_sweep():
        tci.optimize(opts(2), final_sweep1site=False)  # forward + backward half-sweep, each with fill_site_tensors
        gather_cores()

    # ---- untimed: grow the rank to saturation, then W warm-up sweeps ----
    tci.optimize(opts(10), final_sweep1site=False)
    if max(tci.link_dims()) != CHI:
        raise SystemExit(f"rank did not saturate: link dims {tci.link_dims()}")
    for _ in range(args.warmup):
        full_sweep()

    # ---- timed region: exactly K full sweeps ----
    tci.profile_enable(True)
    tci.profile_reset()
    barrier()
    t0 = time.perf_counter()
    if world == 1 and not os.environ.get("T4A_BENCH_PER_SWEEP"):
        # K full sweeps = 2K iterations of optimize_with_finder in one call: fill_site_tensors of iteration t runs on
        # its own stream and overlaps with the bond updates of iteration t+1 (they only need the index sets)
        tci.optimize(opts(2 * args.steps), final_sweep1site=False)
    else:
        for _ in range(args.steps):
            full_sweep()  # the patch-core all-gather needs the cores after every sweep
    barrier()
    dt = time.perf_counter() - t0
    prof = tci.profile()
    variants_timed = tci.profile_variants()
    # calibration outside the timed region: the same sweeps with HIP events around every rrLU launch (the timed region uses the
    # kernels' own start / end stamps: event records between the launches of a chain would be two more packets per bond)
    ev_ms = {}
    if world == 1:
        tci.profile_reset()
        tci.set_chain(True, event_timing=True)
        tci.optimize(opts(4), final_sweep1site=False)
        ev_ms = {v["code"]: v["ms"] / max(v["launches"], 1) for v in tci.profile_variants()}
        tci.set_chain(True)
    tci.profile_enable(False)

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    fl_t = torch.tensor([prof["flops"]], dtype=torch.float64, device="cuda")
    per_rank_ms = [dt / args.steps * 1e3]
    gather_view = None
    if world > 1:
        # what the 8-GPU run needs to be read against (round-4 review, item 10): every rank's own time per step, and what the
        # patch-core all-gather costs where it is NOT hidden — the same K sweeps once more without it, outside the timed region
        all_dt = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
        dist.all_gather(all_dt, dt_t)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in all_dt]
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(fl_t, op=dist.ReduceOp.SUM)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tci.optimize(opts(2), final_sweep1site=False)
        barrier()
        nog = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(nog, op=dist.ReduceOp.MAX)
        gather_view = {
            "collective": "all_gather_into_tensor (RCCL), one per full sweep, device resident, issued behind the sweep's last fill_site_tensors",
            "bytes_per_rank_per_sweep": int(N_SITES * core_cap * 8),
            "bytes_gathered_per_sweep": int(world * N_SITES * core_cap * 8),
            "ms_per_step_without_gather": float(nog.item()) / args.steps * 1e3,
            "exposed_ms_per_step": float(dt_t.item()) / args.steps * 1e3 - float(nog.item()) / args.steps * 1e3,
            "note": "exposed = timed region (with the gather) minus the same sweeps without it, max over ranks each; negative values are run-to-run noise",
        }
    dt_max = float(dt_t.item())
    flops_all = float(fl_t.item())

    if rank == 0:
        steps = args.steps
        shapes = tci.last_sweep_shapes()
        # dominant kernel = the rrLU instantiation with the largest total time (the mid-chain bonds)
        rrlu_ms_avg = prof["dom_ms"] / max(prof["dom_launches"], 1)
        bytes_per_launch = prof["dom_bytes"] / max(prof["dom_launches"], 1)
        kname = rrlu_kernel_name(prof["dom_code"])
        variants = [v for v in variants_timed if v["code"] < 10000000]
        saturated = {int(v["code"]) - 10000000: v for v in variants_timed if v["code"] >= 10000000}  # sub-aggregates, see t4a_gpu.h
        dom = max(variants, key=lambda v: v["ms"]) if variants else None
        dom_steps = dom["steps"] / max(dom["launches"], 1) if dom else float(shapes[N_SITES // 2][2])
        achieved = bytes_per_launch / (rrlu_ms_avg * 1e-3) / 1e9 if rrlu_ms_avg > 0 else 0.0
        floor_us, floor_src = exchange_floor(skip=args.no_floor)
        out = {
            "metric": "TCI2 full-sweep GF/s (d=30, chi=256 fp64)",
            "value": flops_all / dt_max / 1e9,
            "unit": "GF/s",
            "n_gpus": world,
            "rccl_world_size": dist.get_world_size() if world > 1 else 1,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / steps * 1e3,
            "per_rank_ms_per_step": per_rank_ms,
            "patch_core_gather": gather_view,
            "full_sweep_sec": dt_max / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "scaling_note": "N = 1: a single-GPU number, no scaling claim" if world == 1 else "one patch of fixed size per GPU",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "TensorCI2 d=30 interleaved-quantics 2-variable oscillatory integrand, chi_max=256, "
                            "tol=1e-12, nsearch=0 (BASELINE.json configs[2]); one step = forward+backward half-sweep "
                            "incl. fill_site_tensors" + ("" if world == 1 else f"; {world} patches (leading bits fixed), "
                                                         "one per GPU, RCCL all-gather of patch cores per sweep"),
                "n_sites": N_SITES, "local_dim": 2, "chi_max": CHI,
                "function": {"k1": K1, "k2": K2, "k3": K3, "eps": EPS, "k4": K4, "delta": DELTA},
                "parallelism": "single GPU" if world == 1 else f"patch-farm x{world}",
                "mid_bond_MNr": [int(x) for x in shapes[N_SITES // 2]],
                "pivot_steps_per_sweep": prof["pivot_steps"] / steps,
                "evals_per_sweep": prof["evals"] / steps,
                "flops_per_sweep": prof["flops"] / steps,
            },
            # device time of the two kernel families of a sweep (they overlap: fill_site_tensors of one half-sweep runs on its own
            # stream beside the bond chain of the next, so the two do not add up to ms_per_step).  rrlu_kernel: device-side time
            # stamps of every rrLU launch (wall_clock64 at the start and the end of rank 0); fill_site_tensors: HIP events on the
            # fill stream.  The candidate matrices are evaluated inside the rrLU launches (pass-through workgroups) and by
            # chain_pi_kernel for the first bonds; the LUCI factors are not built (every site tensor comes from the fill).
            "breakdown_ms_per_sweep": {
                "rrlu_kernel": prof["rrlu_ms"] / steps, "fill_site_tensors": prof["fill_ms"] / steps,
                "outside_rrlu_kernels": dt_max / steps * 1e3 - prof["rrlu_ms"] / steps,
            },
            "chain": tci.chain_stats(),
            "roofline": {
                "kernel": kname + " (full-pivot rank-revealing LU, one launch per bond)",
                "launches": prof["dom_launches"],
                "share_of_rrlu_time": prof["dom_ms"] / max(prof["rrlu_ms"], 1e-30),
                "bound": "hbm",  # the nominal roofline of the streaming model (AI 0.125 flop/B); see binding_constraint
                "binding_constraint": "per-pivot exchange + instruction latency inside one XCD (the slab is register resident: measured "
                                      "HBM traffic is a few percent of the algorithmic bytes) - compare latency_view / latency_frac",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(kname),
                "traffic_source": os.path.basename(latest_profile("pmc_dominant_kernel.json")) + " (profiles/): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two "
                                  "separate passes of this command, (2*FETCH_SIZE + WRITE_SIZE) KiB per launch "
                                  "(gfx950 FETCH_SIZE correction); null when the kernel name does not match",
                "avg_launch_ms": rrlu_ms_avg,
                "avg_launch_ms_source": "device-side time stamps of rank 0 (wall_clock64) in the timed region",
                "avg_launch_ms_hip_events": ev_ms.get(int(prof["dom_code"])),
                "avg_launch_ms_hip_events_source": "hipEventRecord around every launch of this instantiation on the handle's stream, two "
                                                   "extra sweeps after the timed region",
                # the kernel is a chain of dependent pivot steps; each step needs one key all-gather and one pivot-column
                # hand-off between the workgroups of one XCD: the honest bound is latency, not bandwidth.  The floor is the
                # measured cost of exactly that exchange with no arithmetic around it (tools/xcd_bench.hip on MI355X:
                # 1 715 cycles per round at 2.38 GHz for 32 workgroups and a 700-row column,
                # profiles/r02_xcd_bench.log)
                "latency_view": {
                    "pivot_steps_per_launch": dom_steps,
                    "us_per_pivot_step": 1e3 * rrlu_ms_avg / max(dom_steps, 1.0),
                    "exchange_floor_us": floor_us,
                    "exchange_floor_source": floor_src,
                },
                "latency_frac": floor_us / max(1e3 * rrlu_ms_avg / max(dom_steps, 1.0), 1e-30),
                # the launches of this instantiation that ran all chi_max pivot steps (the saturated mid-chain bonds): chain
                # launches are planned for upper bounds, so the instantiation's average above also contains smaller bonds
                "saturated_launches": saturated_view(saturated.get(int(prof["dom_code"]))),
                # every rrLU instantiation of the timed region, and their time-weighted aggregate
                "all_rrlu_kernels": {
                    "achieved": sum(v["bytes"] for v in variants) / max(sum(v["ms"] for v in variants), 1e-30) / 1e6,
                    "frac": sum(v["bytes"] for v in variants) / max(sum(v["ms"] for v in variants), 1e-30) / 1e6 / HBM_PEAK_GBS,
                    "us_per_pivot_step": 1e3 * sum(v["ms"] for v in variants) / max(sum(v["steps"] for v in variants), 1.0),
                    "variants": [{"kernel": rrlu_kernel_name(v["code"]), "launches": v["launches"], "ms": v["ms"],
                                  "share": v["ms"] / max(prof["rrlu_ms"], 1e-30),
                                  "achieved": v["bytes"] / max(v["ms"], 1e-30) / 1e6,
                                  "us_per_pivot_step": 1e3 * v["ms"] / max(v["steps"], 1.0)}
                                 for v in sorted(variants, key=lambda v: -v["ms"])],
                },
                # the kernels of the sweep that run on the f64 matrix cores (BASELINE.json metric: "... + MFMA %"): flops per
                # launch from the PMC counters, time from the kernel trace of the same command (committed summaries, like `traffic`)
                "mfma": mfma_view(),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "note": "streaming model 8MN + sum_k 16(M-k-1)(N-k-1) bytes (BASELINE.md §2); the slab is register "
                        "resident (one XCD), the kernel is bound by the per-pivot exchange and instruction latency",
            },
        }
        if world == 1 and not args.no_cpu_baseline:  # reported on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(tci, spec)
        if world == 1 and not args.no_aux:
            out["aux"] = aux_timings()
            if not args.no_components:
                try:
                    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
                    import bench_components
                    out["aux"]["components"] = bench_components.components()
                except Exception as e:  # noqa: BLE001 - auxiliary numbers must never break the bench line
                    out["aux"]["components_error"] = repr(e)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


XCD_EXCHANGE_FLOOR_US_R4 = 1.12  # what `tools/xcd_bench floor` measured on the round-4 boxes (key gather + column + two barriers)
XCD_EXCHANGE_FLOOR_US = 1715 / 2380.0  # tools/xcd_bench.hip on MI355X, round 2 (profiles/r02_xcd_bench.log): the fallback of exchange_floor()


def exchange_floor(skip=False):
    """(us, source) of the single-XCD exchange floor: one pivot step's key all-gather + 700-row column hand-off + two barriers
    between 32 workgroups of one XCD with no arithmetic around it.  Re-measured on THIS box by `tools/xcd_bench floor` (built
    by __graft_entry__.build()); the round-2 constant when the binary is missing or fails."""
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "xcd_bench")
    # a fresh child process (never an exec); a profiler's preload and its variables stay with this process
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTX", "HSA_TOOLS"))}
    under_profiler = len(env) != len(os.environ)
    if skip or under_profiler:
        return XCD_EXCHANGE_FLOOR_US_R4, ("constant: tools/xcd_bench floor as measured by the round-4 bench runs (1.12 us; profiles/r04_bench_n1.json)"
                                          + (" - not re-measured: this run is profiled" if under_profiler else " - not re-measured: --no-floor"))
    try:
        out = subprocess.run([exe, "floor"], capture_output=True, text=True, timeout=60, env=env).stdout
        for line in out.splitlines():
            if line.startswith("floor_ns_per_round="):
                ns = float(line.split("=", 1)[1])
                if 100.0 < ns < 1e5:
                    return ns * 1e-3, "measured by this run: tools/xcd_bench floor (best of 4 launches of 2000 rounds, wall time)"
    except (OSError, ValueError, subprocess.SubprocessError):
        pass
    return XCD_EXCHANGE_FLOOR_US, "constant: tools/xcd_bench.hip measured on MI355X in round 2 (profiles/r02_xcd_bench.log); tools/xcd_bench was not available to this run"


def site_shard_mode(args, world, rank, dist, torch, t4a_amd, barrier):
    """BASELINE.json configs[3]: TCI2 at d = 40, chi_max = 512.  The update_pivots chain does not shard (bond b + 1 needs
    the pivots of bond b), so every rank runs it (deterministic: identical pivots everywhere); fill_site_tensors is
    sharded by site (tensorci2.rs:1065-1186: sites are independent given the I/J sets) and after every half-sweep ONE
    all-gather moves the padded cores between the GPUs, device to device, overlapped with the next half-sweep's bond
    updates.  Strong scaling: the total work is fixed, only the fill shrinks with N."""
    from t4a_amd import parallel
    from t4a_amd.functions import quantics_osc2d
    d4, chi4 = 40, 512
    spec = quantics_osc2d(d4, k1=37, k2=53, k3=20011, eps=0.5, k4=1048583, delta=0.5)  # tools/probe_cfg4.py: saturates 512
    tci = t4a_amd.TensorCI2([2] * d4)
    tci.set_function(spec)
    tci.add_global_pivots([[0] * d4])
    tci.set_max_sample_value(1.0)
    tci.set_keep_site_tensors(True)
    tci.set_site_shard(rank, world)

    def opts(iters):
        return t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi4, max_iter=iters, ncheck_history=10 ** 6, nsearch=0,
                                   max_nglobal_pivot=0, seed=42)

    cap = chi4 * 2 * chi4
    xchg = parallel.ShardedCoreExchange(dist if world > 1 else None, torch, d4, cap, parallel.DeviceShardAdapter(tci, torch, cap), "cuda")

    no_exchange = bool(os.environ.get("T4A_SS_NO_EXCHANGE"))  # (diagnosis of ADVICE round 4: the same calls without the core exchange)

    def half_sweep():
        tci.optimize(opts(1), final_sweep1site=False)  # one half-sweep: all bond updates + the local part of the fill
        if not no_exchange:
            xchg.exchange()

    tci.optimize(opts(11), final_sweep1site=False)      # untimed: grow to saturation
    if max(tci.link_dims()) != chi4:
        raise SystemExit(f"rank did not saturate: link dims {tci.link_dims()}")
    for _ in range(2 * args.warmup):
        half_sweep()
    tci.profile_enable(True)
    tci.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(2 * args.steps):
        half_sweep()
    xchg.finish()
    barrier()
    dt = time.perf_counter() - t0
    prof = tci.profile()
    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    fl_t = torch.tensor([prof["flops"]], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(fl_t, op=dist.ReduceOp.SUM)
    if rank == 0:
        # executed flops: the replicated bond chain counts once (it is the same work on every rank), the fill is summed
        chain_flops = prof["flops"]  # this rank: chain + its share of the fill
        out = {
            "metric": "TCI2 full-sweep GF/s (d=40, chi=512 fp64, site-sharded fill)",
            "value": chain_flops / float(dt_t.item()) / 1e9,
            "unit": "GF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": float(dt_t.item()) / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "TensorCI2 d=40 interleaved-quantics 2-variable oscillatory integrand, chi_max=512, tol=1e-12, nsearch=0 "
                                   "(BASELINE.json configs[3]); one step = forward + backward half-sweep; bond chain replicated, "
                                   "fill_site_tensors sharded by site, one RCCL all-gather of padded cores per half-sweep",
                       "n_sites": d4, "local_dim": 2, "chi_max": chi4, "parallelism": f"site-shard x{world}",
                       "gather_bytes_per_half_sweep": int(world * xchg.per_rank * cap * 8)},
            "breakdown_ms_per_sweep": {"rrlu_kernel": prof["rrlu_ms"] / args.steps, "fill_site_tensors_local": prof["fill_ms"] / args.steps},
            # what shards and what does not, so that a measured N-GPU line can be read against its bound: the bond chain
            # (replicated_ms: wall time per sweep minus nothing — the fill runs on its own stream behind it) is the same on
            # every rank; only the fill's device time (sharded_ms at this N, i.e. 1 / N of the whole fill) shrinks.  The fill
            # overlaps the next half-sweep's chain, so the best case for N -> infinity is the chain alone.
            "amdahl": {
                "replicated_ms": prof["rrlu_ms"] / args.steps,
                "sharded_ms": prof["fill_ms"] / args.steps,
                "sharded_ms_at_n1": prof["fill_ms"] / args.steps * world,
                "wall_ms": float(dt_t.item()) / args.steps * 1e3,
                "speedup_bound_vs_n1": (max(prof["rrlu_ms"], prof["fill_ms"] * world) / max(prof["rrlu_ms"], 1e-30)),
                "note": "speedup_bound_vs_n1 = max(chain, whole fill) / chain: the fill (device time) hides behind the replicated "
                        "chain already at N = 1 unless it is longer than the chain, so configs[3] cannot gain more than this from "
                        "more GPUs in the bit-exact mode; the patch farm (--mode headline, N > 1) is the mode that scales",
            },
            "note": "value = flops executed by rank 0 (replicated chain + local share of the fill) / wall time; the chain does not "
                    "shard in the bit-exact mode (SURVEY.md §8e), so N > 1 only shortens the fill",
        }
        print(json.dumps(out), flush=True)


def patch_farm_mode(args, world, rank, dist, torch, t4a_amd, barrier):
    """BASELINE.json configs[4] as stated: PartitionedTT adaptive patching — 64 statically projected patches (the 6 leading bits of the
    configs[2] integrand fixed, 30 active sites each), per-patch TCI2 from scratch with chi_max = 128, farmed over the ranks (patch p on
    rank p % N, FIFO inside a rank: adaptive_interpolation.rs:171-330), eight patches at a time per GPU in lock-step (one XCD each,
    t4a_gpu_tci2_optimize_group), site tensors exported device to device into ONE padded tensor and moved by ONE all_gather_into_tensor
    over RCCL.  One step = the whole job (64 patches + the gather).  Strong scaling: the total work is fixed."""
    from t4a_amd import parallel
    n_patches, chi, n = 64, 128, N_SITES
    opt = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=9, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)

    def make_patch(p):
        t = t4a_amd.TensorCI2([2] * n)
        t.set_function(patch_spec(p, n_patches))
        t.add_global_pivots([[0] * n])
        t.set_max_sample_value(1.0)
        return t

    cap = chi * 2 * chi
    farm = parallel.PaddedPatchFarm(dist if world > 1 else None, torch, n_patches, n, cap, "cuda")
    exporter = parallel.DevicePatchExporter(t4a_amd, torch, make_patch, opt, group=8)
    for _ in range(args.warmup):
        farm.run(exporter, timed=True)
    barrier()
    t0 = time.perf_counter()
    exp_s = gat_s = 0.0
    for _ in range(args.steps):
        farm.run(exporter, timed=True)
        exp_s += farm.last_export_s
        gat_s += farm.last_gather_s
    barrier()
    dt = time.perf_counter() - t0
    stats = torch.tensor([dt, exp_s, gat_s], dtype=torch.float64, device="cuda")
    allr = [torch.zeros_like(stats) for _ in range(world)]
    if world > 1:
        dist.all_gather(allr, stats)
    else:
        allr = [stats]
    if rank == 0:
        per = [[float(v) for v in a.cpu()] for a in allr]
        wall = max(p[0] for p in per)
        dims_ok = all(max(farm.core_dims(p, s)[2] for s in range(n - 1)) == chi for p in range(n_patches))
        print(json.dumps({
            "metric": "PartitionedTT patch farm: patches/s (64 patches, per-patch TCI2 chi=128 fp64, from scratch, cores gathered)",
            "value": n_patches * args.steps / wall, "unit": "patches/s", "n_gpus": world, "rccl_world_size": world if world > 1 else 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[4]: 64 statically projected patches of the d=30 interleaved-quantics 2-variable integrand "
                                   "(6 leading bits fixed), per-patch crossinterpolate2-style optimize (9 iterations from one pivot, chi_max=128, "
                                   "tol=1e-12, nsearch=0), fill_site_tensors, ONE all_gather_into_tensor of the padded cores",
                       "n_patches": n_patches, "chi_max": chi, "n_sites": n, "group": 8, "patches_per_rank": farm.per_rank,
                       "parallelism": f"patch farm x{world}"},
            "per_rank_ms_per_step": [p[0] / args.steps * 1e3 for p in per],
            "per_rank_compute_ms_per_step": [p[1] / args.steps * 1e3 for p in per],
            "exposed_gather_ms_per_step": [p[2] / args.steps * 1e3 for p in per],
            "gather_bytes_per_step": farm.gather_bytes,
            "all_patches_reach_chi": bool(dims_ok),
            "note": "the gather is timed fully exposed (the device is synchronised in front of it); at N = 1 it is a device-to-device copy",
        }), flush=True)


def pi_shard_mode(args, world, rank, dist, torch, t4a_amd, barrier):
    """SURVEY.md 8e row 2 (tensorci2.rs:1859-1893): the candidate matrix of a HOST-callback function evaluated by column blocks over the
    ranks, one all-gather per matrix, rrLU replicated (identical pivots on every rank).  Workload: BASELINE configs[2] (d = 30, chi = 256)
    saturated, the integrand behind tools/native_callback.c (a native t4a_gpu_batch_eval_fn on ONE host thread per rank; its measured cost
    per point is part of the line).  One step = forward + backward half-sweep incl. fill_site_tensors."""
    import ctypes
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_components import NativeCallback
    from t4a_amd import parallel
    spec = patch_spec(0, 1)
    tci = t4a_amd.TensorCI2([2] * N_SITES)
    tci.set_function(spec)
    tci.add_global_pivots([[0] * N_SITES])
    tci.set_max_sample_value(1.0)

    def opts(iters):
        return t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=CHI, max_iter=iters, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)

    tci.optimize(opts(10), final_sweep1site=False)  # untimed, built-in functor: grow to saturation (identical on every rank)
    if max(tci.link_dims()) != CHI:
        raise SystemExit(f"rank did not saturate: link dims {tci.link_dims()}")
    cb = NativeCallback(spec)
    cb.attach(tci)
    gather = None
    if world > 1:
        gather = parallel.PiShardGather(dist, torch, device="cuda")
        tci.set_pi_shard(rank, world, gather)
    # the callback alone: nanoseconds per point on this host
    rng = np.random.default_rng(0)
    n_probe = 1 << 18
    idx = np.ascontiguousarray(rng.integers(0, 2, size=(n_probe, N_SITES)), dtype=np.uint32)
    buf = np.zeros(n_probe)
    fn = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p)(cb.fn_addr)
    ns_per_point = float("inf")
    for _ in range(3):  # (best of three: the estimate callback_ms = points x this must not exceed the wall time it is compared with)
        t0 = time.perf_counter()
        fn(cb.ctx_addr, idx.ctypes.data, N_SITES, n_probe, buf.ctypes.data)
        ns_per_point = min(ns_per_point, (time.perf_counter() - t0) / n_probe * 1e9)
    for _ in range(max(args.warmup, 1)):
        tci.optimize(opts(2), final_sweep1site=False)
    tci.profile_enable(True)
    tci.profile_reset()
    p0 = cb.ctx.points
    g0 = getattr(gather, "seconds", 0.0) if gather else 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tci.optimize(opts(2), final_sweep1site=False)
    barrier()
    dt = time.perf_counter() - t0
    prof = tci.profile()
    points = int(cb.ctx.points - p0)
    gather_s = (getattr(gather, "seconds", 0.0) - g0) if gather else 0.0
    stats = torch.tensor([dt, points * ns_per_point * 1e-9, gather_s, prof["rrlu_ms"] * 1e-3, float(points)], dtype=torch.float64, device="cuda")
    allr = [torch.zeros_like(stats) for _ in range(world)]
    if world > 1:
        dist.all_gather(allr, stats)
    else:
        allr = [stats]
    if rank == 0:
        per = [[float(v) for v in a.cpu()] for a in allr]
        wall = max(p[0] for p in per)
        total_points = sum(p[4] for p in per)
        print(json.dumps({
            "metric": "TCI2 full sweep through a host callback, candidate matrices sharded by column blocks (d=30, chi=256 fp64): sweeps/s",
            "value": args.steps / wall, "unit": "sweeps/s", "n_gpus": world, "rccl_world_size": world if world > 1 else 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2] integrand behind a native host batch callback (tools/native_callback.c, one host "
                                   "thread per rank), saturated chi=256; every candidate matrix evaluated by column blocks over the ranks, one "
                                   "all-gather per matrix, rrLU replicated; one step = forward + backward half-sweep incl. fill_site_tensors",
                       "n_sites": N_SITES, "chi_max": CHI, "parallelism": f"pi column-block shard x{world}",
                       "callback_ns_per_point_one_thread": ns_per_point},
            "per_rank_ms_per_step": [p[0] / args.steps * 1e3 for p in per],
            "callback_ms_per_step": [p[1] / args.steps * 1e3 for p in per],
            "gather_ms_per_step": [p[2] / args.steps * 1e3 for p in per],
            "rrlu_ms_per_step": [p[3] / args.steps * 1e3 for p in per],
            "callback_points_per_step_all_ranks": total_points / args.steps,
            "gather_note": "PiShardGather stages every matrix through a host buffer, a device tensor, RCCL and back (the callback's values are "
                           "host values): fine for an expensive f, the dominant overhead for a cheap one",
        }), flush=True)


def saturated_view(v):
    if not v or v["launches"] <= 0 or v["ms"] <= 0:
        return None
    ms = v["ms"] / v["launches"]
    by = v["bytes"] / v["launches"]
    return {"launches": v["launches"], "avg_launch_ms": ms, "algorithmic_bytes_per_launch": by, "achieved": by / (ms * 1e-3) / 1e9,
            "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "us_per_pivot_step": 1e3 * v["ms"] / max(v["steps"], 1.0)}


def rrlu_kernel_name(code):
    """Kernel instantiation behind a profile code (include/t4a_gpu.h, t4a_gpu_tci2_profile_variants)."""
    code = int(code)
    if code >= 400000:  # kernels for matrices beyond one XCD (kernels_rrlu_xcd2m.hip): 400000 + K * 10000 + RPT * 100 + CPT * 10
        c = code - 400000
        return "t4a::(anonymous namespace)::rrlu_xcd2m_kernel<%d, %d, %s, %d>" % ((c % 10000) // 100, (c % 100) // 10, "true" if (c % 10) & 4 else "false", c // 10000)
    if code >= 300000:  # one-wave kernel (kernels_rrlu_w1.hip): register columns; no factored matrix in a 2-site chain
        c = code - 300000
        return "t4a::(anonymous namespace)::rrlu_w1_kernel<%d, %s, false>" % ((c % 1000) // 10, "true" if (c % 10) & 4 else "false")
    if code >= 200000:  # one-workgroup kernel (kernels_rrlu_wg.hip): rows per lane, columns per wave
        c = code - 200000
        return "t4a::(anonymous namespace)::rrlu_wg_kernel<%d, %d, %s>" % (c // 1000, (c % 1000) // 10, "true" if (c % 10) & 4 else "false")
    if code >= 100000:  # single-XCD kernel (second generation; the first was retired in round 6)
        c = code - 100000
        gen = "rrlu_xcd2_kernel"
        return "t4a::%s<%d, %d, %s>" % (gen, c // 100, (c % 100) // 10, "true" if (c % 10) & 4 else "false")
    if code >= 0:
        return "t4a::rrlu_reg_kernel<%d, %d, %s, %s, %s>" % (code // 1000, (code % 1000) // 10, "true" if (code % 10) & 2 else "false",
                                                            "true" if (code % 10) & 1 else "false", "true" if (code % 10) & 4 else "false")
    return {-1: "t4a::rrlu_kernel<true>", -2: "t4a::rrlu_kernel<false>"}.get(code, "t4a::rg_* (HBM-resident rrLU)")


def mfma_view():
    """MFMA kernels of the sweep (fill_site_tensors: LU trailing update, triangular solve; GEMM when issued): flops per launch,
    average time, TF/s and the fraction of the 78.6 TF/s f64 MFMA peak — from the latest profiles/rNN_mfma_kernels.json
    (tools/mfma_summary.py over the rocprofv3 passes of this command)."""
    path = latest_profile("mfma_kernels.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    out = {"peak_tflops": d.get("peak_tflops"), "sustained_tflops": d.get("sustained_tflops"), "source": "profiles/" + os.path.basename(path) + "",
           "kernels": {}}
    for k, v in d.get("kernels", {}).items():
        out["kernels"][k] = {kk: v.get(kk) for kk in ("launches", "avg_us", "mfma_flops_per_launch", "tflops", "frac_of_peak", "frac_of_sustained")}
    return out


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that committed one."""
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    try:
        names = sorted(n for n in os.listdir(pdir) if n.endswith("_" + suffix) and n[0] == "r" and n[1:3].isdigit())
    except OSError:
        names = []
    return os.path.join(pdir, names[-1]) if names else os.path.join(pdir, "r00_" + suffix)


def pmc_traffic(kname):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (bench.py itself cannot run under
    rocprofv3 --pmc); only reported when the summary is for the kernel instantiation that dominated this run."""
    path = latest_profile("pmc_dominant_kernel.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    short = kname.replace("t4a::", "")
    if not short:
        return None
    for name, nbytes in d.get("rrlu_variants", d.get("rrlu_reg_variants", {})).items():
        if short in name:
            return nbytes
    return d["hbm_bytes_per_launch"] if short in d.get("kernel", "") else None


def host_cpu():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cpu_baseline(tci, spec):
    """The CPU oracle (C++ restatement of the reference algorithm) timed on the GPU box's host cores on a bounded sample of the
    same workload, started from the device's saturated I/J sets.  Two rows: the native build with OpenMP over
    the candidate-matrix evaluations and the fill sites on all usable cores (the reference's TCI2 path itself is single
    threaded apart from its BLAS pool), and the same library pinned to one thread.  Reported baseline, not the target."""
    import subprocess
    import t4a_amd
    model, nproc, usable = host_cpu()
    native = os.path.join(ROOT, "oracle", "_native", "liboracle.so")
    built_native = False
    try:  # build on THIS machine: -march=native of another host must not be reused
        subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=300)
        built_native = os.path.exists(native)
    except (OSError, subprocess.SubprocessError):
        built_native = False
    if built_native:
        os.environ["T4A_ORACLE_LIB"] = native
    import oracle_binding as ob

    def prepare():
        o = ob.OracleTCI2([2] * N_SITES)
        o.set_function(spec)
        for p in range(N_SITES):
            o.set_index_set(0, p, tci.i_set(p))
            o.set_index_set(1, p, tci.j_set(p))
        o.set_max_sample_value(tci.max_sample_value())
        o.clear_history()
        return o

    def opts(n_sweeps):
        return t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=CHI, max_iter=2 * n_sweeps, ncheck_history=10 ** 6, nsearch=0,
                                   max_nglobal_pivot=0, seed=42)

    # flop count of exactly this work: the device's executed flops for the same sweeps, plus the LUCI factors the reference
    # (and the oracle) builds and discards when history extras were merged (tensorci2.rs:1942-1949) and the device skips
    n_sweeps = 3
    o = prepare()
    tci.clear_history()
    tci.profile_enable(True)
    tci.profile_reset()
    tci.optimize(opts(n_sweeps), final_sweep1site=False)
    flops = tci.profile()["flops"]
    tci.profile_enable(False)
    shapes = tci.last_sweep_shapes()
    factor_flops = sum(float((m - r) * r * r + 2.0 * r * r * n) for (m, n, r) in shapes) * 2 * n_sweeps
    threads_native = ob._lib.oracle_openmp_threads() if built_native else 0
    t0 = time.perf_counter()
    o.optimize(opts(n_sweeps), final_sweep1site=False)
    sec_all = time.perf_counter() - t0
    same = all((tci.i_set(p).shape == o.i_set(p).shape) and (tci.i_set(p) == o.i_set(p)).all() for p in range(N_SITES))
    rows = [{"threads": max(threads_native, 1), "full_sweep_sec": sec_all / n_sweeps, "value": flops / sec_all / 1e9,
             "value_incl_discarded_factors": (flops + factor_flops) / sec_all / 1e9,
             "build": "-O3 -march=native -fopenmp -ffp-contract=off (OpenMP over candidate-matrix evaluations and fill sites)" if built_native
             else "-O3 -march=x86-64-v2 -ffp-contract=off (prebuilt, scalar)"}]
    if built_native and threads_native > 1:  # the same library on one thread: the reference's own threading model
        ob._lib.oracle_set_threads(1)
        o1 = prepare()
        t0 = time.perf_counter()
        o1.optimize(opts(1), final_sweep1site=False)
        sec1 = time.perf_counter() - t0
        ob._lib.oracle_set_threads(threads_native)
        rows.append({"threads": 1, "full_sweep_sec": sec1, "value": flops / n_sweeps / sec1 / 1e9,
                     "value_incl_discarded_factors": (flops + factor_flops) / n_sweeps / sec1 / 1e9,
                     "build": "same library, OMP threads = 1"})
    best = max(rows, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "GF/s", "cores": best["threads"], "kind": "port", "full_sweep_sec": best["full_sweep_sec"],
            "cpu_model": model, "nproc": nproc, "usable_cores": usable, "rows": rows,
            "sample": "%d full sweeps (forward + backward half-sweep incl. fill_site_tensors) of the same d=30 chi=256 workload, started "
                      "from the device's saturated index sets; oracle = oracle/ C++ restatement of the reference algorithm, no FMA; "
                      "`value` counts the flops the device executes for these sweeps, `value_incl_discarded_factors` adds the LUCI "
                      "factors the reference builds and discards when history extras are merged" % n_sweeps,
            "pivots_identical_to_device": bool(same)}


def aux_timings():
    """Single-GPU times of the other BASELINE.json configurations (driver-visible regression guard, not the headline):
    cfg2 (d=20, cos(10x)exp(-x), chi<=64, tol 1e-8) time to solution, cfg4-size (d=40, chi=512) full sweep, cfg5-size (one of the
    64 patches, 30 active sites, chi=128) full sweep."""
    import t4a_amd
    from t4a_amd.functions import quantics_trig_exp
    out = {}
    try:
        spec2 = quantics_trig_exp(20)
        o2 = t4a_amd.TCI2Options(tolerance=1e-8, max_bond_dim=64, max_iter=20, nsearch=0, max_nglobal_pivot=0, seed=42)
        best = None
        for _ in range(3):
            t = t4a_amd.TensorCI2([2] * 20)
            t.set_function(spec2)
            t0 = time.perf_counter()
            t.crossinterpolate2([[0] * 20], o2)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["cfg2_time_to_solution_ms"] = best * 1e3
        out["cfg2_rank"] = int(max(t.link_dims()))
        st = t.chain_stats()  # (three 2-site half-sweeps + the final 1-site sweep, each ONE persistent workgroup: DESIGN.md section 5.6)
        out["cfg2_sweeps_chained_walked_one_site"] = [st["half_sweeps"] + st["one_site_sweeps"], st["walked_sweeps"], st["one_site_sweeps"]]
    except Exception as e:  # noqa: BLE001 - auxiliary numbers must never break the bench line
        out["cfg2_error"] = str(e)
    try:
        n_patches, chi5 = 64, 128
        t = t4a_amd.TensorCI2([2] * N_SITES)
        t.set_function(patch_spec(17, n_patches))
        t.add_global_pivots([[0] * N_SITES])
        t.set_max_sample_value(1.0)
        o5 = lambda it: t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi5, max_iter=it, ncheck_history=10 ** 6, nsearch=0,
                                            max_nglobal_pivot=0, seed=42)
        t.optimize(o5(9), final_sweep1site=False)
        t0 = time.perf_counter()
        t.optimize(o5(4), final_sweep1site=False)
        out["cfg5_patch_full_sweep_ms"] = (time.perf_counter() - t0) / 2 * 1e3
        out["cfg5_patch_max_link_dim"] = int(max(t.link_dims()))
    except Exception as e:  # noqa: BLE001
        out["cfg5_error"] = str(e)
    try:  # eight cfg5 patches side by side: every handle runs its rank-revealing LUs on its own XCD (one GPU, 8 host threads)
        import threading
        n_patches, chi5 = 64, 128

        def grow(p, out):
            tp = t4a_amd.TensorCI2([2] * N_SITES)
            tp.set_function(patch_spec(p, n_patches))
            tp.add_global_pivots([[0] * N_SITES])
            tp.set_max_sample_value(1.0)
            tp.optimize(t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi5, max_iter=11, ncheck_history=10 ** 6, nsearch=0,
                                            max_nglobal_pivot=0, seed=42), final_sweep1site=False)
            tp.fill_site_tensors()
            out[p] = float(tp.sum())

        seq, par = {}, {}
        t0 = time.perf_counter()
        for p in range(2):
            grow(p, seq)
        t_seq = (time.perf_counter() - t0) / 2
        thread_errors = []

        def grow_guarded(p, out):
            try:
                grow(p, out)
            except Exception as e:  # noqa: BLE001 - an exception inside a thread would otherwise vanish
                thread_errors.append(f"patch {p}: {e}")

        ths = [threading.Thread(target=grow_guarded, args=(p, par)) for p in range(8)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        t_par = (time.perf_counter() - t0) / 8
        if thread_errors:
            raise RuntimeError("; ".join(thread_errors))
        out["cfg5_patch_from_scratch_ms_one_at_a_time"] = t_seq * 1e3
        out["cfg5_patch_from_scratch_ms_eight_side_by_side"] = t_par * 1e3
        out["cfg5_side_by_side_results_identical"] = bool(all(par[p] == seq[p] for p in seq))
        # the same eight patches through t4a_gpu_tci2_optimize_group: ONE host thread, one chain of launches for all eight
        # (every kernel serves all handles; handle i's rank-revealing LU runs on XCD i of the same launch)
        o_grp = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi5, max_iter=11, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0, seed=42)
        best, grp = None, {}
        for _ in range(2):
            t0 = time.perf_counter()
            tps = []
            for p in range(8):
                tp = t4a_amd.TensorCI2([2] * N_SITES)
                tp.set_function(patch_spec(p, n_patches))
                tp.add_global_pivots([[0] * N_SITES])
                tp.set_max_sample_value(1.0)
                tps.append(tp)
            t4a_amd.optimize_group(tps, o_grp, final_sweep1site=False)
            t4a_amd.fill_site_tensors_group(tps)  # all eight fills issued, then completed (each on its handle's own stream)
            for p, tp in enumerate(tps):
                grp[p] = float(tp.sum())
            dt = (time.perf_counter() - t0) / 8
            best = dt if best is None else min(best, dt)
            group_half_sweeps = int(tps[0].chain_stats()["group_half_sweeps"])
            del tps
        out["cfg5_patch_from_scratch_ms_group_of_eight"] = best * 1e3
        out["cfg5_group_results_identical"] = bool(all(grp[p] == seq[p] for p in seq))
        out["cfg5_group_half_sweeps_per_handle"] = group_half_sweeps
    except Exception as e:  # noqa: BLE001
        out["cfg5_concurrent_error"] = str(e)
    try:
        from t4a_amd.functions import quantics_osc2d
        d4, chi4 = 40, 512
        spec4 = quantics_osc2d(d4, k1=37, k2=53, k3=20011, eps=0.5, k4=1048583, delta=0.5)  # tools/probe_cfg4.py
        t = t4a_amd.TensorCI2([2] * d4)
        t.set_function(spec4)
        t.add_global_pivots([[0] * d4])
        t.set_max_sample_value(1.0)
        o4 = lambda it: t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi4, max_iter=it, ncheck_history=10 ** 6, nsearch=0,
                                            max_nglobal_pivot=0, seed=42)
        t.optimize(o4(11), final_sweep1site=False)
        t0 = time.perf_counter()
        t.optimize(o4(2), final_sweep1site=False)
        out["cfg4_size_full_sweep_ms"] = (time.perf_counter() - t0) * 1e3
        out["cfg4_size_max_link_dim"] = int(max(t.link_dims()))
        # the same size through the plain sweep2site API (tensorci2.rs:746-798: no history extras): the mid-chain matrices are
        # exactly 1024 x 1024, which one XCD holds (the iterations of optimize above merge the extras: ~1450 x 1450, chip-wide kernel)
        t.sweep2site(True, o4(1))
        t.sweep2site(False, o4(1))
        t0 = time.perf_counter()
        t.sweep2site(True, o4(1))
        t.sweep2site(False, o4(1))
        out["cfg4_size_sweep2site_pair_ms"] = (time.perf_counter() - t0) * 1e3
    except Exception as e:  # noqa: BLE001
        out["cfg4_error"] = str(e)
    return out


if __name__ == "__main__":
    main()

"""GPU test of the multi-GPU plumbing bench.py uses at --gpus N > 1, on ONE GPU: pipelined optimize (the last
fill_site_tensors stays in flight), asynchronous core export ordered after that fill, and an RCCL all-gather
(world_size 1 process group) waiting for the export on the device."""
import os
import socket

import numpy as np
import pytest
# PyTorch-ROCm ships its own HIP runtime; whichever runtime a process loads first serves it.  With torch imported here (at
# collection time) it is loaded before libt4a_gpu.so in every selection of test files — loaded after it, torch finds the
# system runtime in place of its own and reports no device.
import torch  # noqa: F401

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def rccl():
    import torch
    import torch.distributed as dist
    import t4a_amd
    if t4a_amd.device_count() < 1 or not torch.cuda.is_available():
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_pipelined_export_and_rccl_all_gather(rccl):
    import torch
    import t4a_amd
    dist = rccl
    if True:
        n, chi = 14, 16
        spec = t4a_amd.quantics_trig_exp(n)  # cos(10x) exp(-x): low rank, so the capped interpolant is accurate
        tci = t4a_amd.TensorCI2([2] * n)
        tci.set_function(spec)
        tci.add_global_pivots([[0] * n])
        tci.set_keep_site_tensors(True)
        opt = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=2, ncheck_history=10 ** 6, nsearch=0,
                                  max_nglobal_pivot=0)
        cap = chi * 2 * chi
        send = [torch.zeros(n * cap, dtype=torch.float64, device="cuda") for _ in range(2)]
        recv = [torch.zeros(n * cap, dtype=torch.float64, device="cuda") for _ in range(2)]
        stream = torch.cuda.current_stream()
        for sweep in range(4):
            k = sweep % 2
            tci.optimize(opt, final_sweep1site=False)          # returns with the last fill still in flight
            tci.export_site_tensors_async(send[k].data_ptr(), cap, stream.cuda_stream)
            work = dist.all_gather_into_tensor(recv[k], send[k], async_op=True)
            work.wait()
        torch.cuda.synchronize()
        got = recv[1].cpu().numpy()                             # sweep 3 used buffer pair 1
        for s in range(n):
            core = tci.site_tensor(s)
            assert core.size > 0
            flat = core.reshape(-1, order="F")
            assert np.array_equal(got[s * cap:s * cap + flat.size], flat), f"core {s} differs after the gather"
        # the gathered cores form the interpolant of the last sweep
        pts = np.random.default_rng(0).integers(0, 2, size=(50, n))
        from oracle_binding import fn_eval
        assert np.abs(tci.evaluate(pts) - fn_eval(spec, pts)).max() < 1e-6


def test_patch_farm_on_device_matches_oracle(rccl):
    """BASELINE config 5 in miniature: independent crossinterpolate2 runs on patches of the bench integrand
    (leading bits projected), farmed through parallel.run_patch_farm over RCCL; every patch equals the oracle's."""
    import torch
    import t4a_amd
    import bench
    import oracle_binding as ob
    from t4a_amd import parallel
    n_patches, chi = 4, 24
    opt = t4a_amd.TCI2Options(tolerance=1e-9, max_bond_dim=chi, max_iter=4, nsearch=0, max_nglobal_pivot=0)
    n = bench.N_SITES

    def run_patch(p):
        t = t4a_amd.TensorCI2([2] * n)
        t.set_function(bench.patch_spec(p, n_patches))
        t.crossinterpolate2([[0] * n], opt)
        return [t.site_tensor(s) for s in range(n)]

    farmed = parallel.run_patch_farm(rccl, torch, n_patches, run_patch, device="cuda")
    assert len(farmed) == n_patches
    for p in range(n_patches):
        o = ob.OracleTCI2([2] * n)
        o.set_function(bench.patch_spec(p, n_patches))
        o.crossinterpolate2([[0] * n], opt)
        for s in range(n):
            a, b = farmed[p][s], o.site_tensor(s)
            assert a.shape == b.shape, f"patch {p} site {s}"
            assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()), f"patch {p} site {s}"
    # patches differ from each other (the projected bits matter)
    assert not np.array_equal(farmed[0][n - 1], farmed[1][n - 1]) or not np.array_equal(farmed[0][0], farmed[1][0])


def test_device_resident_shard_exchange_matches_unsharded_fill(rccl):
    """BASELINE.json configs[3] plumbing on one GPU: two handles stand for two ranks (identical index sets, site shards 0 / 1 of
    2); their local cores travel through parallel.ShardedCoreExchange with the device adapter — device-to-device export into the
    padded send buffer, all_gather_into_tensor over RCCL (world size 1 per process group, so the two shards are spliced by hand),
    device-to-device import — and the assembled train is bitwise the unsharded fill_site_tensors."""
    import torch
    import t4a_amd
    from t4a_amd import parallel
    n, chi, world = 16, 24, 2
    spec = t4a_amd.quantics_osc2d(n, k1=3, k2=5, k3=11, eps=0.5, k4=101, delta=0.5)
    opt = t4a_amd.TCI2Options(tolerance=1e-10, max_bond_dim=chi, max_iter=4, nsearch=0, max_nglobal_pivot=0)
    ref = t4a_amd.TensorCI2([2] * n)
    ref.set_function(spec)
    ref.add_global_pivots([[0] * n])
    ref.optimize(opt, final_sweep1site=False)
    ref.fill_site_tensors()
    cap = chi * 2 * chi
    per_rank = (n + world - 1) // world
    shards, sends = [], []
    for r in range(world):
        t = t4a_amd.TensorCI2([2] * n)
        t.set_function(spec)
        t.add_global_pivots([[0] * n])
        t.optimize(opt, final_sweep1site=False)   # same chain on every "rank": identical index sets
        t.set_site_shard(r, world)
        t.fill_site_tensors()                      # local sites only
        send = torch.zeros(per_rank * cap, dtype=torch.float64, device="cuda")
        parallel.DeviceShardAdapter(t, torch, cap).export_shard(send)
        shards.append(t)
        sends.append(send)
    gathered = torch.cat(sends)                    # what all_gather_into_tensor delivers: [world][per_rank][cap]
    # the collective itself (world size 1 group): must reproduce its input
    out = torch.zeros_like(sends[0])
    rccl.all_gather_into_tensor(out, sends[0])
    torch.cuda.synchronize()
    assert torch.equal(out, sends[0])
    for r, t in enumerate(shards):
        parallel.DeviceShardAdapter(t, torch, cap).import_shard(gathered, per_rank)
        torch.cuda.synchronize()
        for s in range(n):
            a, b = t.site_tensor(s), ref.site_tensor(s)
            assert a.shape == b.shape and np.array_equal(a, b), f"rank {r} site {s}"
    # the single-process form of the exchange object (world == 1): export, local copy, import round trip
    one = shards[0]
    one.set_site_shard(0, 1)
    one.fill_site_tensors()
    x = parallel.ShardedCoreExchange(None, torch, n, cap, parallel.DeviceShardAdapter(one, torch, cap), "cuda")
    x.exchange()
    x.exchange()
    x.finish()
    torch.cuda.synchronize()
    for s in range(n):
        assert np.array_equal(one.site_tensor(s), ref.site_tensor(s))


def test_cfg5_full_size_64_patches_chi128(rccl):
    """BASELINE.json configs[4] at size on one GPU: 64 static patches (6 leading bits of the d = 30 bench integrand
    projected, 30 active sites each), per-patch TCI2 with max_bond_dim = 128, farmed through the device-resident
    parallel.PaddedPatchFarm (world size 1 over RCCL), eight patches at a time with t4a_gpu_tci2_optimize_group.  All 64 patches come back in patch (FIFO) order and tile the domain;
    a sample of 8 patches is compared with the CPU oracle: bit-exact I/J sets and bond errors, interpolant values to 1e-10."""
    import torch
    import t4a_amd
    import bench
    import oracle_binding as ob
    from t4a_amd import parallel
    n_patches, chi, n = 64, 128, bench.N_SITES
    opt = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=9, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0,
                              seed=42)
    sample = [0, 9, 18, 27, 36, 45, 54, 63]
    pts = np.random.default_rng(5).integers(0, 2, size=(200, n))
    kept = {}
    order = []

    def make_patch(p):
        t = t4a_amd.TensorCI2([2] * n)
        t.set_function(bench.patch_spec(p, n_patches))
        t.add_global_pivots([[0] * n])
        t.set_max_sample_value(1.0)
        return t

    def keep(p, t):
        order.append(p)
        if p in sample:
            kept[p] = t

    # device-resident farm: eight patches at a time in lock-step (one XCD each, t4a_gpu_tci2_optimize_group), cores exported
    # device-to-device into one padded tensor, ONE all_gather_into_tensor over RCCL for the payload and one for the shapes
    cap = chi * 2 * chi
    farm = parallel.PaddedPatchFarm(rccl, torch, n_patches, n, cap, "cuda")
    farm.run(parallel.DevicePatchExporter(t4a_amd, torch, make_patch, opt, group=8, keep=keep))
    torch.cuda.synchronize()
    assert order == list(range(n_patches))          # FIFO inside the rank
    # every patch reaches the cap (so its interpolant is a truncation, not exact: values are checked against the oracle below),
    # arrives at ITS place after export / all-gather, and differs from its neighbour
    sums = {}
    for p in range(n_patches):
        dims = [farm.core_dims(p, s)[2] for s in range(n - 1)]
        assert max(dims) == chi, f"patch {p}: link dims {dims}"
        sums[p] = [float(farm.core_device(p, s).abs().sum().item()) for s in (0, n // 2, n - 1)]
    for p in range(n_patches):
        assert sums[p] != sums[(p + 1) % n_patches]
    farmed = {p: farm.cores(p) for p in sample}
    for p in sample:  # the gathered cores are the handle's cores, bit for bit
        for s in range(n):
            assert np.array_equal(farmed[p][s], kept[p].site_tensor(s)), f"patch {p} site {s} changed on its way through the farm"
    # the sample against the oracle
    for p in sample:
        o = ob.OracleTCI2([2] * n)
        o.set_function(bench.patch_spec(p, n_patches))
        o.add_global_pivots([[0] * n])
        o.set_max_sample_value(1.0)
        o.optimize(opt, final_sweep1site=False)
        o.fill_site_tensors()
        g = kept[p]
        for s in range(n):
            assert np.array_equal(g.i_set(s), o.i_set(s)), f"patch {p}: I set of site {s}"
            assert np.array_equal(g.j_set(s), o.j_set(s)), f"patch {p}: J set of site {s}"
        assert np.array_equal(g.bond_errors(), o.bond_errors())
        # ... and the patch really went through the GROUP chain (one chain of launches for its eight handles, its rrLU on one XCD
        # of a shared launch), never through the per-bond host loop
        st = g.chain_stats()
        assert st["group_half_sweeps"] == st["half_sweeps"] > 0 and st["fell_back"] == 0 and st["not_eligible"] == 0, st
        # values: the interpolants agree to 1e-10 of the largest value on random points.  (The raw cores are NOT compared: at a
        # saturated, truncated rank some pivot matrices P are ill conditioned, so T = Pi1 P^-1 moves by ~1e-3 between two
        # correct LU orderings — device blocked/MFMA vs the oracle's unblocked loop — while the train it belongs to does not.)
        for s in range(n):
            assert farmed[p][s].shape == o.site_tensor(s).shape
        gv, ov = g.evaluate(pts), o.evaluate(pts)
        assert np.abs(gv - ov).max() <= 1e-10 * max(1.0, np.abs(ov).max()), f"patch {p}"


def test_site_sharded_fill_on_device_is_bitwise_the_unsharded_fill(rccl):
    """BASELINE config 4's sharding: with identical index sets, rank r fills the sites s % world == r and the cores
    are exchanged through device pointers; the assembled train is bitwise the unsharded fill_site_tensors."""
    import torch
    import t4a_amd
    n, chi, world = 16, 24, 2
    spec = t4a_amd.quantics_osc2d(n, k1=3, k2=5, k3=11, eps=0.5, k4=101, delta=0.5)
    opt = t4a_amd.TCI2Options(tolerance=1e-10, max_bond_dim=chi, max_iter=4, nsearch=0, max_nglobal_pivot=0)
    ref = t4a_amd.TensorCI2([2] * n)
    ref.set_function(spec)
    ref.add_global_pivots([[0] * n])
    ref.optimize(opt, final_sweep1site=False)
    ref.fill_site_tensors()
    shards = []
    for r in range(world):
        t = t4a_amd.TensorCI2([2] * n)
        t.set_function(spec)
        for p in range(n):
            t.set_index_set(0, p, ref.i_set(p))
            t.set_index_set(1, p, ref.j_set(p))
        t.set_site_shard(r, world)
        t.fill_site_tensors()
        shards.append(t)
    buf = torch.zeros(chi * 2 * chi, dtype=torch.float64, device="cuda")
    for s in range(n):
        owner, other = shards[s % world], shards[(s + 1) % world]
        dims = owner.site_tensor_dims(s)
        owner.site_tensor_to_device(s, buf.data_ptr())
        torch.cuda.synchronize()
        other.set_site_tensor_from_device(s, dims, buf.data_ptr())
    for t in shards:
        for s in range(n):
            assert np.array_equal(t.site_tensor(s), ref.site_tensor(s)), f"site {s}"
    pts = np.random.default_rng(1).integers(0, 2, size=(64, n))
    assert np.array_equal(shards[0].evaluate(pts), ref.evaluate(pts))


def test_site_shard_exchange_loop_replays_the_fill_graph_on_a_side_stream(rccl):
    """replaying the captured fill_site_tensors graph on a handle whose cores are exported / imported
    asynchronously faulted in 25 - 75 % of `bench.py --mode site-shard` runs.  Round 5 bisected it to the LEGACY DEFAULT STREAM as
    event consumer / producer (profiles/r05_fill_graph_fault_bisect.txt): on a side stream the replay is safe, on stream 0 the
    library issues its fills directly.  
    Round 6: the root cause is not identified, so the DEFAULT is the round-4 guard again — no replay on any handle with
    asynchronously shared site tensors — and the relaxed guard (replay unless the legacy / a blocking stream took part) is an opt-in
    (t4a_gpu_tci2_set_chain bit 4).  Three arms, looped: many exchanges, every gathered core bitwise the unsharded fill."""
    import torch
    import t4a_amd
    from t4a_amd import parallel
    n, chi = 16, 32
    spec = t4a_amd.quantics_osc2d(n, k1=37, k2=53, k3=211, eps=0.3)
    cap = chi * 2 * chi

    def run(legacy, relaxed=False):
        if legacy:
            torch.cuda.set_stream(torch.cuda.default_stream())
        else:
            torch.cuda.set_stream(torch.cuda.Stream())
        tci = t4a_amd.TensorCI2([2] * n)
        tci.set_function(spec)
        tci.add_global_pivots([[0] * n])
        tci.set_max_sample_value(1.0)
        tci.set_keep_site_tensors(True)
        tci.set_site_shard(0, 1)
        tci.set_chain(True, fill_graph_relaxed=relaxed)
        opt = t4a_amd.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=1, ncheck_history=10 ** 6, nsearch=0, max_nglobal_pivot=0)
        import os
        os.environ["T4A_SS_LEGACY_STREAM" if legacy else "T4A_SS_UNUSED"] = "1"
        try:
            xchg = parallel.ShardedCoreExchange(None, torch, n, cap, parallel.DeviceShardAdapter(tci, torch, cap), "cuda")
        finally:
            os.environ.pop("T4A_SS_LEGACY_STREAM", None)
            os.environ.pop("T4A_SS_UNUSED", None)
        assert (torch.cuda.current_stream().cuda_stream == 0) == legacy
        for _ in range(24):           # saturates after ~8 half-sweeps: the remaining fills have identical signatures
            tci.optimize(opt, final_sweep1site=False)
            xchg.exchange()
        xchg.finish()
        torch.cuda.synchronize()
        st = tci.fill_stats()
        cores = [tci.site_tensor(s) for s in range(n)]
        ref = t4a_amd.TensorCI2([2] * n)
        ref.set_function(spec)
        for s in range(n):
            ref.set_index_set(0, s, tci.i_set(s))
            ref.set_index_set(1, s, tci.j_set(s))
        ref.fill_site_tensors()
        for s in range(n):
            assert np.array_equal(cores[s], ref.site_tensor(s)), s
        return st

    try:
        side = run(False)
        assert side["graph_replays"] == 0 and side["fills"] >= 20, side                       # default: direct issue on a shared handle
        opted = run(False, relaxed=True)
        assert opted["graph_replays"] >= 1 and opted["graph_captures"] >= 1, opted           # opt-in, side stream: replay
        legacy = run(True, relaxed=True)
        assert legacy["graph_replays"] == 0 and legacy["fills"] >= 20, legacy                  # opt-in, legacy stream: still direct issue
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())

"""world_size-2 gloo tests of the N > 1 orchestration (tensor4all-rs_amd/python/t4a_amd/parallel.py).
The local compute is the CPU oracle here (tests may use it); production injects the device handle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _patch_fn(p):
    # patch p of f(i,j,k,l) = 1/(1 + i + 2j + 3k + 4l + 0.5p): the patch index plays the projected leading site
    return lambda idx: 1.0 / (1.0 + idx[0] + 2 * idx[1] + 3 * idx[2] + 4 * idx[3] + 0.5 * p)


def _run_patch(p):
    import oracle_binding as ob
    from t4a_amd import TCI2Options
    t = ob.OracleTCI2([3, 3, 3, 3])
    t.set_function(_patch_fn(p))
    t.crossinterpolate2([[1, 1, 1, 1]], TCI2Options(tolerance=1e-10, nsearch=0, max_nglobal_pivot=0))
    return [t.site_tensor(s) for s in range(4)]


def _worker(rank, world, port, n_patches, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from t4a_amd import parallel
    cores = parallel.run_patch_farm(dist, torch, n_patches, _run_patch)
    np.save(os.path.join(out_dir, f"farm_{rank}.npy"),
            np.concatenate([c.ravel(order="F") for pc in cores for c in pc]))

    # the same farm through the padded, device-resident form (parallel.PaddedPatchFarm): one all_gather_into_tensor for the
    # payload, one for the shapes; the local compute here is the CPU oracle writing into the (CPU) send tensor
    n_sites = len(_run_patch(0))
    cap = max(int(np.prod(c.shape)) for q in range(n_patches) for c in _run_patch(q))

    def export_patches(patches, send):
        shapes = []
        for k, q in enumerate(patches):
            cs = _run_patch(q)
            for s_, c in enumerate(cs):
                flat = np.asarray(c, dtype=np.float64).ravel(order="F")
                send[k, s_, :flat.size] = torch.from_numpy(flat.copy())
            shapes.append([c.shape for c in cs])
        return shapes

    farm = parallel.PaddedPatchFarm(dist, torch, n_patches, n_sites, cap, "cpu").run(export_patches)
    for q in range(n_patches):
        for s_, (a, b) in enumerate(zip(farm.cores(q), cores[q])):
            assert a.shape == b.shape and np.array_equal(a, b), (q, s_)

    # site-sharded fill: every rank holds the same index sets; rank r fills sites s % world == r
    import oracle_binding as ob
    from t4a_amd import TCI2Options
    t = ob.OracleTCI2([3, 3, 3, 3])
    t.set_function(_patch_fn(0))
    t.crossinterpolate2([[1, 1, 1, 1]], TCI2Options(tolerance=1e-10, nsearch=0, max_nglobal_pivot=0))
    full = [t.site_tensor(s) for s in range(4)]
    store = {}

    def fill_my_sites(r, w):
        for s in range(4):
            if s % w == r:
                store[s] = full[s].copy()  # stands for the local fill of site s

    parallel.sharded_fill(dist, torch, 4, fill_my_sites, lambda s: store[s], lambda s, c: store.__setitem__(s, c))
    assert sorted(store) == [0, 1, 2, 3]
    for s in range(4):
        assert store[s].shape == full[s].shape and np.array_equal(store[s], full[s])
    # the same exchange through the padded, double-buffered all_gather_into_tensor path that the device uses
    # (parallel.ShardedCoreExchange): three half-sweeps, the cores of the last one must be complete on every rank
    dims = {s: full[s].shape for s in range(4)}
    cap = max(int(np.prod(d)) for d in dims.values())
    store2 = {}
    adapter = parallel.NumpyShardAdapter(torch, store2, 4, rank, world, cap, lambda s: dims[s])
    xchg = parallel.ShardedCoreExchange(dist, torch, 4, cap, adapter, "cpu")
    for half_sweep in range(3):
        for s in range(rank, 4, world):
            store2[s] = full[s] * (half_sweep + 1.0)   # stands for the local fill of this half-sweep
        xchg.exchange()
    xchg.finish()
    assert sorted(store2) == [0, 1, 2, 3]
    for s in range(4):
        assert store2[s].shape == full[s].shape and np.array_equal(store2[s], full[s] * 3.0), s
    # ranks that are NOT replicas of each other: the importer assumes another bond dimension than the exporter packed.  The
    # gathered (l, s, r) header exposes it (the old exchange reinterpreted the remote cores silently).
    wrong = dict(dims)
    wrong[1 - rank] = (dims[1 - rank][0], dims[1 - rank][1], dims[1 - rank][2] + 1)   # a site of the other rank
    cap3 = cap + 64
    bad = parallel.ShardedCoreExchange(dist, torch, 4, cap3, parallel.NumpyShardAdapter(torch, dict(store2), 4, rank, world, cap3,
                                                                                      lambda s: wrong[s]), "cpu")
    bad.exchange()
    try:
        bad.finish()
        raise AssertionError("shape mismatch between exporter and importer went unnoticed")
    except RuntimeError as e:
        assert "do not hold the same index sets" in str(e)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_patches", [3, 4])
def test_patch_farm_and_sharded_fill_two_ranks(tmp_path, n_patches):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_patches, str(tmp_path)), nprocs=world, join=True)
    ref = np.concatenate([c.ravel(order="F") for p in range(n_patches) for c in _run_patch(p)])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"farm_{r}.npy"))
        assert got.shape == ref.shape and np.array_equal(got, ref)


def test_patch_assignment_is_round_robin_and_order_preserving():
    from t4a_amd.parallel import patches_of_rank, pack_cores, unpack_cores
    assert patches_of_rank(7, 0, 2) == [0, 2, 4, 6]
    assert patches_of_rank(7, 1, 2) == [1, 3, 5]
    assert sorted(sum((patches_of_rank(64, r, 8) for r in range(8)), [])) == list(range(64))
    rng = np.random.default_rng(0)
    cores = [np.asfortranarray(rng.normal(size=s)) for s in [(1, 2, 3), (3, 2, 4), (4, 2, 1)]]
    h, f = pack_cores(cores)
    back = unpack_cores(h, f)
    assert all(np.array_equal(a, b) for a, b in zip(cores, back))


# ---- column-block shard of the candidate matrix for callback functions -------------------------------
def _pi_fn(idx):
    # an "expensive" user function of six sites; the value depends on every digit, so a misplaced column shows
    return float(np.cos(0.37 * idx[0] + 1.3 * idx[1] - 0.7 * idx[2]) * np.exp(-0.11 * (idx[3] + 2 * idx[4])) + 0.01 * idx[5] * idx[0])


def _pi_sets():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 3, size=(7, 2))     # row halves: sites 0-1
    b = rng.integers(0, 3, size=(11, 4))    # column halves: sites 2-5 (11 columns: uneven blocks, a short last block)
    return a, b


def _pi_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import t4a_amd
    from t4a_amd import parallel
    seen = []

    def f(idx):
        seen.append(tuple(int(v) for v in idx))
        return _pi_fn(idx)
    a, b = _pi_sets()
    gather = parallel.PiShardGather(dist, torch)
    m = t4a_amd.pi_shard_eval(rank, world, f, a, 0, b, 2, 6, gather)
    np.save(os.path.join(out_dir, f"pi_{rank}.npy"), m)
    np.save(os.path.join(out_dir, f"seen_{rank}.npy"), np.asarray(seen, dtype=np.int64))
    # a single column (fewer columns than ranks: one block is empty) and a single row
    m1 = t4a_amd.pi_shard_eval(rank, world, _pi_fn, a, 0, b[:1], 2, 6, gather)
    m2 = t4a_amd.pi_shard_eval(rank, world, _pi_fn, a[:1], 0, b, 2, 6, gather)
    np.save(os.path.join(out_dir, f"pi1_{rank}.npy"), m1)
    np.save(os.path.join(out_dir, f"pi2_{rank}.npy"), m2)
    assert gather.calls == 3
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pi_column_block_shard_assembles_the_unsharded_matrix(tmp_path, world):
    import t4a_amd
    a, b = _pi_sets()
    full = t4a_amd.pi_shard_eval(0, 1, _pi_fn, a, 0, b, 2, 6)
    want = np.array([[_pi_fn(list(ra) + list(cb)) for cb in b] for ra in a])
    assert np.array_equal(full, want)
    port = _free_port()
    mp.spawn(_pi_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cbk = -(-len(b) // world)
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"pi_{r}.npy")), want), r     # identical on every rank, bitwise
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"pi1_{r}.npy")), want[:, :1])
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"pi2_{r}.npy")), want[:1, :])
        # rank r's callback saw exactly its column block, row index outer / column index inner (tensorci2.rs:1862-1869 restricted)
        seen = np.load(os.path.join(str(tmp_path), f"seen_{r}.npy"))
        cols = b[r * cbk:(r + 1) * cbk]
        expect = np.array([list(ra) + list(cb) for ra in a for cb in cols], dtype=np.int64).reshape(-1, 6)
        assert np.array_equal(seen.reshape(-1, 6), expect), r


def test_pi_shard_argument_errors():
    import t4a_amd
    a, b = _pi_sets()
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.pi_shard_eval(2, 2, _pi_fn, a, 0, b, 2, 6, lambda s: s)
    assert e.value.code == t4a_amd.INVALID_ARGUMENT
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.pi_shard_eval(0, 2, _pi_fn, a, 0, b, 2, 6, lambda s: np.zeros(1))    # a gather that returns the wrong size
    assert e.value.code == t4a_amd.CALLBACK_ERROR
    with pytest.raises(t4a_amd.T4aError) as e:
        t4a_amd.pi_shard_eval(0, 1, _pi_fn, a, 0, b, 3, 6)                             # halves do not cover the sites
    assert e.value.code == t4a_amd.INVALID_ARGUMENT

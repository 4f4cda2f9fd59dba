"""The device-side bond chain (tci2_chain.hip / kernels_chain.hip): a whole 2-site half-sweep enqueued at once, index sets as
device tables, dimensions read on the device.  It must change nothing: every test runs the same problem through the chain,
through the per-bond host path and through the CPU oracle and compares index sets, errors and cores.

Reference: update_pivots / sweep2site / optimize_with_finder, crates/tensor4all-tensorci/src/tensorci2.rs:746-798, 1626-1802,
1821-2007; kronecker_i / kronecker_j :1224-1246; the history extras :1675-1689, :1833-1846."""
import os

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

PARITY = dict(nsearch=0, max_nglobal_pivot=0)


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


def trio(t4a, spec, dims):
    c = t4a.TensorCI2(dims)       # chain, with the read-back check of the device tables after every half-sweep
    c.set_function(spec)
    c.set_chain(True, verify=True)
    h = t4a.TensorCI2(dims)       # per-bond host path
    h.set_function(spec)
    h.set_chain(False)
    o = ob.OracleTCI2(dims)
    o.set_function(spec)
    return c, h, o


def assert_same_state(c, h, o, n, cores=True):
    for p in range(n):
        assert np.array_equal(c.i_set(p), o.i_set(p)), f"I set of the chain differs from the oracle at site {p}"
        assert np.array_equal(c.j_set(p), o.j_set(p)), f"J set of the chain differs from the oracle at site {p}"
        assert np.array_equal(c.i_set(p), h.i_set(p)) and np.array_equal(c.j_set(p), h.j_set(p))
    assert np.array_equal(c.bond_errors(), h.bond_errors()) and np.array_equal(c.bond_errors(), o.bond_errors())
    assert c.max_sample_value() == h.max_sample_value() == o.max_sample_value()
    if cores:
        for p in range(n):
            a, b = c.site_tensor(p), h.site_tensor(p)
            assert a.shape == b.shape and np.array_equal(a, b), f"core {p}: chain and per-bond path differ"


@pytest.mark.parametrize("n,chi,iters", [(10, 8, 5), (14, 24, 7), (16, 48, 8)])
def test_chain_growth_with_history_extras(t4a, n, chi, iters):
    """Ranks grow over the iterations; from the second one on the previous iteration's sets are merged as extras."""
    spec = t4a.quantics_osc2d(n, k1=3, k2=5, k3=7, eps=0.1, k4=11, delta=0.3)
    opts = t4a.TCI2Options(tolerance=1e-12, max_bond_dim=chi, max_iter=iters, ncheck_history=10 ** 6, seed=1, **PARITY)
    c, h, o = trio(t4a, spec, [2] * n)
    for t in (c, h, o):
        t.add_global_pivots([[0] * n])
        t.set_max_sample_value(1.0) if hasattr(t, "set_max_sample_value") else None
        t.optimize(opts, final_sweep1site=False)
    assert_same_state(c, h, o, n)
    assert c.history()[0] == o.history()[0] and np.array_equal(c.history()[1], o.history()[1])
    assert np.array_equal(c.last_sweep_shapes(), o.last_sweep_shapes())
    st = c.chain_stats()
    assert st["half_sweeps"] == iters and st["bonds"] == iters * (n - 1) and st["fell_back"] == 0 and st["not_eligible"] == 0
    assert h.chain_stats()["half_sweeps"] == 0


def test_chain_mixed_local_dimensions(t4a):
    """Mixed-radix codes: local dimensions 3, 2, 4, 5, 2, 3 (Lorentzian on an uneven grid)."""
    from t4a_amd.functions import lorentz
    dims = [3, 2, 4, 5, 2, 3, 4]
    spec = lorentz(dims)
    opts = t4a.TCI2Options(tolerance=1e-10, max_bond_dim=12, max_iter=6, ncheck_history=10 ** 6, **PARITY)
    c, h, o = trio(t4a, spec, dims)
    piv = [[1, 1, 2, 3, 0, 2, 1]]
    for t in (c, h, o):
        t.crossinterpolate2(piv, opts)
    assert_same_state(c, h, o, len(dims))
    assert c.chain_stats()["half_sweeps"] >= 2 and c.chain_stats()["fell_back"] == 0
    pts = np.random.default_rng(3).integers(0, 2, size=(100, len(dims)))
    assert np.abs(c.evaluate(pts) - o.evaluate(pts)).max() <= 1e-10


def test_chain_sweep2site_and_strictly_nested(t4a):
    n = 12
    spec = t4a.quantics_trig_exp(n, a=25.0, b=0.5, cc=0.5, cs=1.0)
    c, h, o = trio(t4a, spec, [2] * n)
    piv = [[0, 1] * (n // 2)]
    opts = t4a.TCI2Options(tolerance=1e-10, max_bond_dim=6, **PARITY)
    for t in (c, h, o):
        t.add_global_pivots(piv)
        t.sweep2site(True, opts)
        t.sweep2site(False, opts)
        t.sweep2site(True, opts)
    assert_same_state(c, h, o, n)
    assert c.chain_stats()["half_sweeps"] == 3
    nested = t4a.TCI2Options(tolerance=1e-10, max_bond_dim=8, max_iter=4, strictly_nested=True, ncheck_history=10 ** 6, **PARITY)
    for t in (c, h, o):
        t.optimize(nested, final_sweep1site=True)
    assert_same_state(c, h, o, n)


def test_chain_survives_host_side_changes_of_the_sets(t4a):
    """Global pivots, sweep1site and set_index_set change the host's sets behind the chain's back: the tables are re-uploaded."""
    n = 12
    spec = t4a.quantics_osc2d(n, k1=3, k2=5, k3=7, eps=0.1, k4=11, delta=0.3)
    opts = t4a.TCI2Options(tolerance=1e-12, max_bond_dim=16, max_iter=3, ncheck_history=10 ** 6, **PARITY)
    c, h, o = trio(t4a, spec, [2] * n)
    for t in (c, h, o):
        t.add_global_pivots([[0] * n])
        t.optimize(opts, final_sweep1site=False)
        t.add_global_pivots([[1, 0] * (n // 2), [1] * n])
        t.optimize(opts, final_sweep1site=True)   # ends with a 1-site sweep (host path)
        t.optimize(opts, final_sweep1site=False)
    assert_same_state(c, h, o, n)
    assert c.chain_stats()["fell_back"] == 0 and c.chain_stats()["half_sweeps"] == 9


def test_chain_is_skipped_for_callbacks_and_rook(t4a):
    n = 8
    f = lambda idx: 1.0 / (1.0 + sum((i + 1) * v for i, v in enumerate(idx)))
    g = t4a.TensorCI2([3] * n)
    g.set_function(f)
    o = ob.OracleTCI2([3] * n)
    o.set_function(f)
    opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=10, max_iter=4, **PARITY)
    g.crossinterpolate2([[0] * n], opts)
    o.crossinterpolate2([[0] * n], opts)
    for p in range(n):
        assert np.array_equal(g.i_set(p), o.i_set(p)) and np.array_equal(g.j_set(p), o.j_set(p))
    st = g.chain_stats()
    assert st["half_sweeps"] == 0 and st["not_eligible"] > 0


def test_chain_at_cfg3_size_equals_per_bond_path(t4a):
    """BASELINE.json configs[2] shapes (d = 30, chi = 256): growth to saturation and one more full sweep, chain vs per-bond path,
    every index set and bond error identical; the mid-chain bonds run on the single-XCD kernel with device-side dimensions."""
    import bench
    n = bench.N_SITES
    spec = bench.patch_spec(0, 1)
    opts = t4a.TCI2Options(tolerance=1e-12, max_bond_dim=bench.CHI, max_iter=12, ncheck_history=10 ** 6, seed=42, **PARITY)
    res = []
    for chain in (True, False):
        t = t4a.TensorCI2([2] * n)
        t.set_function(spec)
        t.set_chain(chain, verify=chain)
        t.add_global_pivots([[0] * n])
        t.set_max_sample_value(1.0)
        t.optimize(opts, final_sweep1site=False)
        res.append(t)
    c, h = res
    assert max(c.link_dims()) == bench.CHI
    for p in range(n):
        assert np.array_equal(c.i_set(p), h.i_set(p)) and np.array_equal(c.j_set(p), h.j_set(p)), p
    assert np.array_equal(c.bond_errors(), h.bond_errors())
    assert np.array_equal(c.last_sweep_shapes(), h.last_sweep_shapes())
    st = c.chain_stats()
    assert st["half_sweeps"] == 12 and st["fell_back"] == 0 and st["not_eligible"] == 0
    pts = np.random.default_rng(11).integers(0, 2, size=(300, n))
    c.fill_site_tensors()   # (optimize invalidates the site tensors at the end of every iteration, tensorci2.rs:707-708)
    h.fill_site_tensors()
    assert np.array_equal(c.evaluate(pts), h.evaluate(pts))


def test_optimize_group_equals_optimize_on_every_handle(t4a):
    """t4a_gpu_tci2_optimize_group: several handles in lock-step from one thread (one XCD each) — the per-GPU form of the patch farm
    (BASELINE.json configs[4]).  Every handle must end exactly where its own optimize() call ends, also when the handles need
    different numbers of iterations (different tolerances do not apply: one option set; different functions do)."""
    import bench
    n = 16
    specs = [t4a.quantics_osc2d(n, k1=3 + p, k2=5, k3=7 + 2 * p, eps=0.1 * (p + 1), k4=11, delta=0.3) for p in range(5)]
    specs.append(t4a.quantics_trig_exp(n))          # converges after a few iterations: drops out of the group early
    opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=24, max_iter=9, seed=3, **PARITY)
    solo, grouped = [], []
    for spec in specs:
        for dst in (solo, grouped):
            t = t4a.TensorCI2([2] * n)
            t.set_function(spec)
            t.add_global_pivots([[0] * n])
            dst.append(t)
    for t in solo:
        t.optimize(opts, final_sweep1site=True)
    t4a.optimize_group(grouped, opts, final_sweep1site=True)
    for k, (a, b) in enumerate(zip(solo, grouped)):
        for p in range(n):
            assert np.array_equal(a.i_set(p), b.i_set(p)) and np.array_equal(a.j_set(p), b.j_set(p)), (k, p)
        assert a.history()[0] == b.history()[0] and np.array_equal(a.history()[1], b.history()[1]), k
        assert a.termination() == b.termination() and a.max_sample_value() == b.max_sample_value()
        for p in range(n):
            assert np.array_equal(a.site_tensor(p), b.site_tensor(p)), (k, p)
        assert b.chain_stats()["fell_back"] == 0 and b.chain_stats()["half_sweeps"] > 0
        assert a.chain_stats()["group_half_sweeps"] == 0
    # while at least two handles are active their half-sweeps run as ONE chain of launches (the last survivor runs alone)
    assert all(b.chain_stats()["group_half_sweeps"] > 0 for b in grouped)
    assert len(set(len(t.history()[0]) for t in grouped)) > 1, "the test wants handles that stop at different iterations"
    with pytest.raises(t4a.T4aError):
        t4a.optimize_group([grouped[0], grouped[0]], opts)


def test_cfg5_patch_in_a_group_with_an_early_finisher_matches_oracle(t4a):
    """One BASELINE configs[4] patch (patch 17 of 64, chi = 128, 30 active sites) from its single initial pivot through
    t4a_gpu_tci2_optimize_group, next to a second patch and to a member that converges after a few iterations and drops out of the
    group: the patch must end where the ORACLE ends — checked after 3, 6 and all 11 iterations (a run of k iterations ends in the
    state a longer run passes through after its k-th)."""
    import bench
    n = bench.N_SITES
    easy = t4a.quantics_trig_exp(n)
    for k in (3, 6, 11):
        opts = t4a.TCI2Options(tolerance=1e-12, max_bond_dim=128, max_iter=k, ncheck_history=3, seed=42, **PARITY)
        members = []
        for spec in (bench.patch_spec(17, 64), bench.patch_spec(40, 64), easy):
            t = t4a.TensorCI2([2] * n)
            t.set_function(spec)
            t.add_global_pivots([[0] * n])
            t.set_max_sample_value(1.0)
            members.append(t)
        t4a.optimize_group(members, opts, final_sweep1site=False)
        o = ob.OracleTCI2([2] * n)
        o.set_function(bench.patch_spec(17, 64))
        o.add_global_pivots([[0] * n])
        o.set_max_sample_value(1.0)
        o.optimize(opts, final_sweep1site=False)
        g = members[0]
        for p in range(n):
            assert np.array_equal(g.i_set(p), o.i_set(p)) and np.array_equal(g.j_set(p), o.j_set(p)), (k, p)
        assert g.link_dims() == o.link_dims() and np.array_equal(g.pivot_errors(), o.pivot_errors()), k
        assert g.history()[0] == o.history()[0] and np.array_equal(g.history()[1], o.history()[1]), k
        assert np.array_equal(g.last_sweep_shapes(), o.last_sweep_shapes()), k
        assert g.chain_stats()["fell_back"] == 0 and g.chain_stats()["group_half_sweeps"] > 0
    assert len(members[2].history()[0]) < len(members[0].history()[0]), "the test wants a member that stops early"
    assert max(g.link_dims()) == 128
    g.fill_site_tensors()
    o.fill_site_tensors()
    pts = np.random.default_rng(6).integers(0, 2, size=(300, n))
    gv, ov = g.evaluate(pts), o.evaluate(pts)
    assert np.abs(gv - ov).max() <= 1e-10 * max(1.0, np.abs(ov).max())


@pytest.mark.parametrize("chi", [40, 96])
def test_chained_one_site_sweep_through_the_launched_chain_matches_oracle(t4a, chi):
    """The final 1-site sweep of crossinterpolate2 (tensorci2.rs:1781-1794, :865-1050) as a LAUNCHED chain — matrices of 2 chi x chi
    are too wide for the persistent workgroup: chi = 40 runs the one-workgroup kernel, chi = 96 the single-XCD kernel, both keeping
    the factored matrix of every bond; ranks above 16 take Engine::build_factors_from bond by bond, the rest the batched launch —
    and a stand-alone backward + forward sweep1site pair: index sets, errors bitwise, site tensors to 1e-9."""
    import bench
    n = bench.N_SITES
    opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=chi, max_iter=5, nsearch=0, max_nglobal_pivot=0)
    g = t4a.TensorCI2([2] * n)
    o = ob.OracleTCI2([2] * n)
    for t in (g, o):
        t.set_function(bench.patch_spec(1, 4))
        t.crossinterpolate2([[0] * n], opts)
    st = g.chain_stats()
    assert st["one_site_sweeps"] == 1 and st["one_site_fell_back"] == 0 and st["one_site_not_eligible"] == 0
    assert st["walked_sweeps"] < st["half_sweeps"] + st["one_site_sweeps"], "the test wants the launched chain"
    assert max(g.link_dims()) > 16

    def same():
        for p in range(n):
            assert np.array_equal(g.i_set(p), o.i_set(p)) and np.array_equal(g.j_set(p), o.j_set(p)), p
        assert g.link_dims() == o.link_dims()
        assert np.array_equal(g.bond_errors(), o.bond_errors()) and np.array_equal(g.pivot_errors(), o.pivot_errors())
        for s_ in range(n):
            a, b = g.site_tensor(s_), o.site_tensor(s_)
            assert a.shape == b.shape, s_
            assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()), s_

    same()
    for t in (g, o):
        t.sweep1site(False, 1e-10, 1e-13, chi // 2, False)
        t.sweep1site(True, 1e-10, 1e-13, chi // 2, True)
    assert g.chain_stats()["one_site_sweeps"] == 3 and g.chain_stats()["one_site_fell_back"] == 0
    same()


def test_fill_site_tensors_group_equals_fill_on_every_handle(t4a):
    """t4a_gpu_tci2_fill_site_tensors_group: the fills of several handles issued first, completed afterwards — site tensors bitwise
    those of handle.fill_site_tensors(), also for handles of different length and for one with a host callback (which fills
    synchronously inside the group call)."""
    specs = [(14, t4a.quantics_osc2d(14, k1=3, k2=5, k3=7, eps=0.1, k4=11, delta=0.3)),
             (14, t4a.quantics_osc2d(14, k1=2, k2=7, k3=5, eps=0.2, k4=13, delta=0.4)),
             (10, t4a.quantics_trig_exp(10))]
    opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=20, max_iter=5, seed=3, **PARITY)
    solo, grouped = [], []
    for n, spec in specs:
        for dst in (solo, grouped):
            t = t4a.TensorCI2([2] * n)
            t.set_function(spec)
            t.add_global_pivots([[0] * n])
            t.optimize(opts, final_sweep1site=False)
            dst.append(t)
    cb = t4a.TensorCI2([3, 2, 3, 2])
    cb.set_function(lambda idx: 1.0 / (1.0 + sum(int(v) for v in idx)))
    cb.add_global_pivots([[0, 0, 0, 0]])
    cb.optimize(opts, final_sweep1site=False)
    cb2 = t4a.TensorCI2([3, 2, 3, 2])
    cb2.set_function(lambda idx: 1.0 / (1.0 + sum(int(v) for v in idx)))
    cb2.add_global_pivots([[0, 0, 0, 0]])
    cb2.optimize(opts, final_sweep1site=False)
    for t in solo + [cb]:
        t.fill_site_tensors()
    t4a.fill_site_tensors_group(grouped + [cb2])
    for a, b in zip(solo + [cb], grouped + [cb2]):
        for p in range(len(a)):
            assert np.array_equal(a.site_tensor(p), b.site_tensor(p)), p
    t4a.fill_site_tensors_group([])


def test_group_chain_with_a_member_on_the_per_bond_path(t4a):
    """A group in which one handle is not eligible for the device-side chain (chain switched off: it runs bond by bond, on an XCD of
    its own, after the group's chain has completed) and the others differ in length of run: results equal the handles' own runs."""
    n = 14
    specs = [t4a.quantics_osc2d(n, k1=2 + p, k2=3, k3=5 + p, eps=0.2, k4=7, delta=0.4) for p in range(4)]
    opts = t4a.TCI2Options(tolerance=1e-8, max_bond_dim=20, max_iter=6, seed=5, **PARITY)
    solo, grouped = [], []
    for spec in specs:
        for dst in (solo, grouped):
            t = t4a.TensorCI2([2] * n)
            t.set_function(spec)
            t.add_global_pivots([[0] * n])
            dst.append(t)
    grouped[1].set_chain(False)
    for t in solo:
        t.optimize(opts, final_sweep1site=False)
    t4a.optimize_group(grouped, opts, final_sweep1site=False)
    for k, (a, b) in enumerate(zip(solo, grouped)):
        for p in range(n):
            assert np.array_equal(a.i_set(p), b.i_set(p)) and np.array_equal(a.j_set(p), b.j_set(p)), (k, p)
        assert np.array_equal(a.history()[1], b.history()[1]), k
    assert grouped[1].chain_stats()["half_sweeps"] == 0
    assert grouped[0].chain_stats()["group_half_sweeps"] > 0 and grouped[0].chain_stats()["fell_back"] == 0


def test_group_chain_mixed_lengths_fall_back_to_own_chains(t4a):
    """Handles with different numbers of sites do not line up: optimize_group runs one chain per handle (group_half_sweeps stays 0)."""
    opts = t4a.TCI2Options(tolerance=1e-8, max_bond_dim=16, max_iter=4, seed=5, **PARITY)
    ts = []
    for n in (10, 12):
        t = t4a.TensorCI2([2] * n)
        t.set_function(t4a.quantics_osc2d(n, k1=3, k2=5, k3=7, eps=0.2, k4=11, delta=0.3))
        t.add_global_pivots([[0] * n])
        ts.append(t)
    ref = []
    for n in (10, 12):
        t = t4a.TensorCI2([2] * n)
        t.set_function(t4a.quantics_osc2d(n, k1=3, k2=5, k3=7, eps=0.2, k4=11, delta=0.3))
        t.add_global_pivots([[0] * n])
        t.optimize(opts, final_sweep1site=False)
        ref.append(t)
    t4a.optimize_group(ts, opts, final_sweep1site=False)
    for a, b in zip(ref, ts):
        for p in range(len(a.link_dims()) + 1):
            assert np.array_equal(a.i_set(p), b.i_set(p)) and np.array_equal(a.j_set(p), b.j_set(p))
        assert b.chain_stats()["group_half_sweeps"] == 0 and b.chain_stats()["half_sweeps"] > 0


def test_exception_while_a_chain_is_in_flight_leaves_the_handles_usable(tmp_path):
    """An error raised between the launch of a bond chain and its completion (here: injected into the issue of the previous
    iteration's fill_site_tensors: T4A_TEST_THROW_IN_FILL, honoured only by the test-hook twin libt4a_gpu_testhooks.so) must not leave the handle with a chain 'in flight' and its XCD — or,
    for a group, the whole chip — reserved: the failed call reports the error, the same handles then optimise normally and reach
    the result of handles that never failed.  Runs in a child process (the injection is read once per process)."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tensor4all-rs_amd/python")
import t4a_amd as t4a
n = 14
def make(k):
    t = t4a.TensorCI2([2] * n)
    t.set_function(t4a.quantics_osc2d(n, k1=3 + k, k2=5, k3=7, eps=0.2, k4=11, delta=0.3))
    t.add_global_pivots([[0] * n])
    return t
opts = t4a.TCI2Options(tolerance=1e-9, max_bond_dim=20, max_iter=6, seed=3, ncheck_history=10**6, nsearch=0, max_nglobal_pivot=0)
# 1. one handle: the 2nd pending fill of the process fails while the 3rd iteration's chain runs
a = make(0)
try:
    a.optimize(opts, final_sweep1site=False)
    print("NO ERROR 1")
except t4a.T4aError as e:
    print("error 1:", "injected" in str(e))
b = make(0)                      # same XCD family, fresh handle: must not wait for a reservation that was never returned
b.optimize(opts, final_sweep1site=False)
a.clear_history()
a2 = make(0)
a2.optimize(opts, final_sweep1site=False)
print("fresh equal:", all(np.array_equal(b.i_set(p), a2.i_set(p)) for p in range(n)))
a.optimize(opts, final_sweep1site=False)   # the handle that failed is usable again
print("failed handle usable:", a.chain_stats()["half_sweeps"] > 0, max(a.link_dims()) == max(b.link_dims()))
'''
    hooks = os.path.join(ROOT, "tensor4all-rs_amd", "lib", "libt4a_gpu_testhooks.so")  # (the production library has no fault injector)
    assert os.path.exists(hooks), "build.py builds the test-hook twin of the library"
    env = dict(os.environ, T4A_TEST_THROW_IN_FILL="2", T4A_GPU_LIB=hooks)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert "error 1: True" in r.stdout, r.stdout + r.stderr
    assert "fresh equal: True" in r.stdout, r.stdout + r.stderr
    assert "failed handle usable: True True" in r.stdout, r.stdout + r.stderr
    # 2. a group: the failure hits a member while the leader holds the chip
    code2 = code.split("# 1.")[0] + r'''
hs = [make(k) for k in range(4)]
try:
    t4a.optimize_group(hs, opts, final_sweep1site=False)
    print("NO ERROR 2")
except t4a.T4aError as e:
    print("error 2:", "injected" in str(e))
ref = [make(k) for k in range(4)]
for t in ref:
    t.optimize(opts, final_sweep1site=False)       # own chains on own XCDs: would hang if the chip were still reserved
hs2 = [make(k) for k in range(4)]
t4a.optimize_group(hs2, opts, final_sweep1site=False)
print("group after failure equal:", all(np.array_equal(x.i_set(p), y.i_set(p)) for x, y in zip(ref, hs2) for p in range(n)))
'''
    env = dict(os.environ, T4A_TEST_THROW_IN_FILL="6", T4A_GPU_LIB=hooks)
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert "error 2: True" in r.stdout, r.stdout + r.stderr
    assert "group after failure equal: True" in r.stdout, r.stdout + r.stderr


def test_group_chain_mixed_local_dimensions_and_strict_nesting(t4a):
    """The group chain with mixed-radix codes (local dimensions 3, 2, 4, 5, 2, 3, 4: the Kronecker step, the parent lookup of the
    extras and the candidate matrix all depend on the site's dimension) and, in a second group, strictly nested sweeps (no
    extras): every member against the CPU oracle, not only against its own solo run."""
    from t4a_amd.functions import lorentz
    dims = [3, 2, 4, 5, 2, 3, 4]
    n = len(dims)
    pivots = [[1, 1, 2, 3, 0, 2, 1], [0, 1, 0, 2, 1, 0, 3], [2, 0, 3, 4, 1, 1, 0]]
    for strictly in (False, True):
        opts = t4a.TCI2Options(tolerance=1e-10, max_bond_dim=12, max_iter=6, ncheck_history=10 ** 6, strictly_nested=strictly, **PARITY)
        spec = lorentz(dims)
        grouped, oracles = [], []
        for piv in pivots:
            g = t4a.TensorCI2(dims)
            g.set_function(spec)
            g.add_global_pivots([piv])
            grouped.append(g)
            o = ob.OracleTCI2(dims)
            o.set_function(spec)
            o.add_global_pivots([piv])
            o.optimize(opts, final_sweep1site=False)
            oracles.append(o)
        t4a.optimize_group(grouped, opts, final_sweep1site=False)
        for k, (g, o) in enumerate(zip(grouped, oracles)):
            for p in range(n):
                assert np.array_equal(g.i_set(p), o.i_set(p)) and np.array_equal(g.j_set(p), o.j_set(p)), (strictly, k, p)
            assert np.array_equal(g.bond_errors(), o.bond_errors()), (strictly, k)
            assert g.history()[0] == o.history()[0] and np.array_equal(g.history()[1], o.history()[1]), (strictly, k)
            st = g.chain_stats()
            assert st["group_half_sweeps"] > 0 and st["fell_back"] == 0, (strictly, k, st)

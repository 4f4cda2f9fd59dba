"""GPU: estimate_true_error / floating_zone / opt_first_pivot through the C ABI (t4a_gpu_tt_floating_zone,
t4a_gpu_tt_estimate_true_error, t4a_gpu_opt_first_pivot): the reference's fixtures and parity with the CPU oracle on the
interpolant of a real TCI2 run (same trajectory: the device evaluate_many is bit-identical to the oracle's)."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def t4a():
    import t4a_amd
    if t4a_amd.device_count() < 1:
        pytest.fail("no MI355X visible: the product path has no CPU fallback")
    return t4a_amd


def test_reference_fixtures(t4a):
    f = lambda i: float(i[0] * i[1])
    one = t4a.SimpleTensorTrain.constant([4, 4], 1.0)
    pivot, err = one.floating_zone(f, [4, 4], init_p=[2, 2])
    assert pivot == [3, 3] and abs(err - 8.0) < 1e-10                       # globalsearch.rs:245-279
    zero = t4a.SimpleTensorTrain.constant([4, 4], 0.0)
    pivot, err = zero.floating_zone(f, [4, 4], init_p=[2, 2])
    assert pivot == [3, 3] and abs(err - 9.0) < 1e-10                       # :141-155
    res = one.estimate_true_error(f, nsearch=10, seed=7)
    assert res[0][0] == [3, 3] and abs(res[0][1] - 8.0) < 1e-10             # :30-46
    errs = [e for _, e in res]
    assert errs == sorted(errs, reverse=True)
    assert t4a.opt_first_pivot(lambda i: (i[0] + i[1] + 1.0) ** 2, [4, 4], [0, 0]) == [3, 3]   # optfirstpivot.rs:80-89
    assert t4a.opt_first_pivot(f, [4, 4], [3, 3]) == [3, 3]


def test_validation(t4a):
    tt = t4a.SimpleTensorTrain.constant([4, 4], 0.0)
    f = lambda i: 1.0
    for kw in (dict(local_dims=[4], init_p=[0]), dict(local_dims=[4, 0], init_p=[0, 0]), dict(local_dims=[4, 4], init_p=[0, 4])):
        with pytest.raises(t4a.T4aError) as e:
            tt.floating_zone(f, **kw)
        assert e.value.code == t4a.INVALID_ARGUMENT
    with pytest.raises(t4a.T4aError) as e:   # a callback that returns too few values is an error, not a silent truncation
        class Short:
            def __call__(self, i):
                return 0.0
            def batched(self, idx):
                return np.zeros(max(len(idx) - 1, 0))
        tt.floating_zone(Short(), [4, 4], init_p=[1, 1])
    assert e.value.code == t4a.CALLBACK_ERROR


def test_parity_with_oracle_on_a_tci2_interpolant(t4a):
    from t4a_amd.functions import quantics_osc2d
    n = 12
    spec = quantics_osc2d(n, k1=3, k2=5, k3=7, eps=0.3)
    opts = t4a.TCI2Options(tolerance=1e-3, max_bond_dim=6, max_iter=4, nsearch=0, max_nglobal_pivot=0)
    g = t4a.TensorCI2([2] * n)
    g.set_function(spec)
    g.crossinterpolate2([[0] * n], opts)
    gtt = g.to_tensor_train()
    cores = [gtt.site_tensor(s) for s in range(n)]
    ott = ob.OracleTT(cores)
    f = lambda i: float(ob.fn_eval(spec, np.asarray([i], dtype=np.uint64))[0])
    rng = np.random.default_rng(11)
    starts = rng.integers(0, 2, size=(6, n)).tolist()
    for s in starts:
        gp, ge = gtt.floating_zone(f, [2] * n, init_p=s)
        op, oe = ott.floating_zone(f, [2] * n, init_p=s)
        assert gp == op and ge == oe
    gr = gtt.estimate_true_error(f, initial_points=starts)
    orr = ott.estimate_true_error(f, initial_points=starts)
    assert gr == orr
    # random starting points: the same splitmix64 stream on both sides ("parity unpinned" against rand 0.9)
    assert gtt.estimate_true_error(f, nsearch=8, seed=5) == ott.estimate_true_error(f, nsearch=8, seed=5)
    assert gtt.floating_zone(f, [2] * n, seed=9) == ott.floating_zone(f, [2] * n, seed=9)
    # opt_first_pivot on the same function
    assert t4a.opt_first_pivot(f, [2] * n, [0] * n) == ob.opt_first_pivot(f, [2] * n, [0] * n)

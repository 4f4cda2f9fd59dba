"""CPU: the oracle's restatement of estimate_true_error / floating_zone / opt_first_pivot
(oracle/t4a_oracle_search.hpp) against the fixtures the reference holds
(crates/tensor4all-tensorci/src/globalsearch.rs:30-46,141-155,245-310 and optfirstpivot.rs:31-37,80-101)."""
import numpy as np
import pytest

import oracle_binding as ob


def test_floating_zone_reference_test_case():
    # globalsearch.rs:245-279: tt == 1 on 4 x 4, f = i*j, start (2, 2) -> pivot (3, 3), error |9 - 1| = 8
    tt = ob.OracleTT(ob.constant_tt([4, 4], 1.0))
    f = lambda i: float(i[0] * i[1])
    pivot, err = tt.floating_zone(f, [4, 4], init_p=[2, 2])
    assert pivot == [3, 3] and abs(err - 8.0) < 1e-10


def test_floating_zone_doc_example():
    # globalsearch.rs:141-155: tt == 0, f = i*j, start (2, 2) -> (3, 3), error 9
    tt = ob.OracleTT(ob.constant_tt([4, 4], 0.0))
    pivot, err = tt.floating_zone(lambda i: float(i[0] * i[1]), [4, 4], init_p=[2, 2])
    assert pivot == [3, 3] and abs(err - 9.0) < 1e-10


def test_floating_zone_validation():
    tt = ob.OracleTT(ob.constant_tt([4, 4], 0.0))
    f = lambda i: 1.0
    with pytest.raises(ob.OracleError):
        tt.floating_zone(f, [4], init_p=[0])           # local_dims length mismatch
    with pytest.raises(ob.OracleError):
        tt.floating_zone(f, [4, 0], init_p=[0, 0])     # zero dimension
    with pytest.raises(ob.OracleError):
        tt.floating_zone(f, [4, 4], init_p=[0, 4])     # pivot out of range


def test_estimate_true_error_doc_example_and_sorting():
    # globalsearch.rs:30-46: tt == 1, f = i*j: worst case at (3, 3) with |9 - 1| = 8; descending order; no duplicates
    tt = ob.OracleTT(ob.constant_tt([4, 4], 1.0))
    f = lambda i: float(i[0] * i[1])
    res = tt.estimate_true_error(f, nsearch=10, seed=7)
    assert res and res[0][0] == [3, 3] and abs(res[0][1] - 8.0) < 1e-10
    errs = [e for _, e in res]
    assert errs == sorted(errs, reverse=True)
    # globalsearch.rs:281-310: tt == 0, f = i + j
    tt0 = ob.OracleTT(ob.constant_tt([4, 4], 0.0))
    res = tt0.estimate_true_error(lambda i: float(i[0] + i[1]), nsearch=10, seed=3)
    errs = [e for _, e in res]
    assert errs == sorted(errs, reverse=True) and res[0] == ([3, 3], 6.0)
    # explicit starting points; consecutive duplicates are removed (dedup_by after the stable sort)
    res = tt.estimate_true_error(f, initial_points=[[2, 2], [3, 3], [0, 1], [1, 0]])
    piv = [p for p, _ in res]
    assert all(piv[k] != piv[k + 1] for k in range(len(piv) - 1)) and piv[0] == [3, 3]


def test_opt_first_pivot_reference_cases():
    f = lambda i: (i[0] + i[1] + 1.0) ** 2
    assert ob.opt_first_pivot(f, [4, 4], [0, 0]) == [3, 3]            # optfirstpivot.rs:80-89
    g = lambda i: float(i[0] * i[1])
    assert ob.opt_first_pivot(g, [4, 4], [3, 3]) == [3, 3]            # :91-99 already optimal
    # no improvement possible from a zero plateau: the pivot stays
    assert ob.opt_first_pivot(lambda i: 0.0, [3, 3, 3], [1, 2, 0]) == [1, 2, 0]
    # max_sweep = 0: nothing is evaluated beyond the start
    assert ob.opt_first_pivot(f, [4, 4], [0, 0], max_sweep=0) == [0, 0]

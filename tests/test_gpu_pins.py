"""The PINS, inside the GPU run.  The driver's `-m gpu` run deselects everything unmarked, so its record would otherwise show
the product against the oracle but not the oracle against the reference.  This module re-runs, under the gpu marker:

* KAT-1 (MultiProductLeadtime.java:41-43, to 17 digits) and the bridge from the reference-pinned two-product family to the
  single-product lead-time family F5 -- on the oracle (tests/test_oracle_kat.py) AND on the product: the GPU's F5 sweep of the
  bridge instance must give the value the reference-pinned recursion gives;
* sdpgpu_getpmf against the 50-digit table (tests/test_pmf_reference.py) and the scipy restatement (tests/test_pmf_abi.py).

Seconds in total."""
import numpy as np
import pytest

from test_oracle_kat import (bridge_f3_f4_workloads, bridge_f5_workload, bridge_instance, _initial_index,  # noqa: F401
                             test_bridge_f3_equals_f4_when_no_interest_accrues,
                             test_bridge_kat_family_equals_f5_with_a_null_second_product, test_kat1_bit_exact)
from test_pmf_abi import test_clsp_variant_matches, test_getpmf_matches_the_scipy_restatement  # noqa: F401
from test_pmf_reference import (test_native_clsp_variant_against_the_50_digit_table,  # noqa: F401
                                test_native_getpmf_against_the_50_digit_table)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel", [0, 1], ids=["auto", "gather"])
def test_gpu_f5_sweep_reproduces_the_reference_pinned_family(sia, oracle, kernel):
    """Product side of the bridge: the HIP path's F5 tables on the bridge instance against (i) the oracle's F5 tables, every
    state and period, bit for bit, and (ii) the two-product recursion the reference's recorded outputs pin (null second
    product, slack 0): V_1(initial state) and the first order."""
    import ctypes
    ml = bridge_instance()
    L = oracle.lib()
    L.sdpref_kat_set_tolerance.argtypes = [ctypes.c_double]
    L.sdpref_kat_set_tolerance(0.0)
    try:
        final, q1, q2, _, _ = oracle.kat_multilead(**ml)
    finally:
        L.sdpref_kat_set_tolerance(0.1)
    w = bridge_f5_workload(ml)
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    V, pol, _ = P.solve()
    d = w.desc()
    d.kernel = kernel
    with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
        eng.solve(sync=True)
        for period in range(1, w.T + 1):
            assert np.array_equal(eng.values(period), V[period - 1]) and np.array_equal(eng.policy(period), pol[period - 1])
        i0 = _initial_index(P, ml)
        assert ml["ini_cash"] + eng.values(1)[i0] == final and (int(eng.policy(1)[i0]), 0) == (q1, q2)


@pytest.mark.parametrize("kernel", [0, 1], ids=["auto", "gather"])
def test_gpu_f3_and_f4_agree_when_no_interest_accrues(sia, oracle, kernel):
    """Product side of the second bridge: the HIP path's F3 (CashConstraint's lambdas) and F4 (CashOverdraft's, whose
    statements the KATs execute) tables on the exact, interest-free instance -- against each other and against the oracle's."""
    w3, w4 = bridge_f3_f4_workloads()
    V, pol, _ = oracle.Problem(w4.desc(), w4.pmf, w4.overhead()).solve()
    tables = []
    for w in (w3, w4):
        d = w.desc()
        d.kernel = kernel
        with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
            eng.solve(sync=True)
            tables.append([(eng.values(p), eng.policy(p)) for p in range(1, w.T + 1)])
    for t in range(w3.T):
        for fam in (0, 1):
            assert np.array_equal(tables[fam][t][0], V[t]) and np.array_equal(tables[fam][t][1], pol[t]), (fam, t)

"""sdp.cash.CashRecursionXR on the GPU (family CASH, cash_formula 2: state (x, R), order-up-to actions): the
driver's own instance (cash.singleItem.CashConstraintXR.main, CashConstraintXR.java:37-77) at full size, every table of
every period bit-identical to the oracle, on both kernels; and the host mirror class (CashRecursionXR / CashStateXR)
against the oracle's literal memoised recursion.  The two small cases of tests/cases.py (f3_xr, f3_xr_fractional --
the latter with a unit cost that makes R - variCost * x inexact) run through test_gpu_parity / test_golden / the
sharded tests with every other family."""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


_XR_MAIN = {}


def _xr_main_oracle(oracle):
    """The oracle's sweep of the main instance (about 20 s on the box's threads): once for both kernels."""
    if not _XR_MAIN:
        w = cases.xr_main_instance()
        threads = min(os.cpu_count() or 1, 16)
        _XR_MAIN["w"] = w
        _XR_MAIN["tables"] = oracle.Problem(w.desc(), w.pmf).solve(nthreads=threads)
    return _XR_MAIN["w"], _XR_MAIN["tables"]


@pytest.mark.parametrize("kernel", [0, 1], ids=["auto_cash_row", "generic"])
def test_cash_constraint_xr_main_full_tables(sia, oracle, kernel):
    w, (V, pol, cells) = _xr_main_oracle(oracle)
    d = w.desc()
    d.kernel = kernel
    with sia.SdpEngine(d, w.pmf) as eng:
        eng.solve()
        assert eng.stats().cells_evaluated == cells
        assert eng.num_states(1) == 501 * 2101
        for period in range(1, w.T + 1):
            assert np.array_equal(eng.values(period), V[period - 1]), f"V_{period}"
            assert np.array_equal(eng.policy(period), pol[period - 1]), f"policy of period {period}"
        # the action list is bounded by R / variCost (up to 1001 levels at cash 2000), not by maxOrderQuantity = 200
        assert cells > 501 * 2101 * 300 * len(w.pmf[0]) * w.T


def test_mirror_class_against_the_literal_recursion(sia, oracle):
    """CashRecursionXR.getExpectedValue / getAction / getOptTable as CashConstraintXR.main uses them (:131-137)."""
    for make in (cases.f3_xr, cases.f3_xr_fractional):
        w = make()
        f = w.functor
        m = oracle.Problem(w.desc(), w.pmf, w.overhead()).memo()
        rec = sia.CashRecursionXR(w.direction, w.pmf, f.feasibleActions and (lambda s, f=f, T=w.T: f.feasibleActions(s, T)),
                                  lambda s, a, r, f=f, T=w.T: f.stateTransition(s, a, r, T),
                                  lambda s, a, r, f=f, T=w.T: f.immediateValue(s, a, r, T), f.discountFactor, functor=f)
        rec.setTreeMapCacheAction()
        ini = sia.CashStateXR(1, f.iniInventory, f.iniCash, f.variCost)
        assert rec.getExpectedValue(ini) == m["value"]
        assert rec.getAction(ini) == m["action"]          # the order-up-to LEVEL
        table = rec.getOptTable()
        assert table.shape == (m["n"], 5)
        got = {(int(r[0]), r[1], r[3]): (r[2], r[4]) for r in table}
        for i in range(m["n"]):
            key = (int(m["period"][i]), m["x"][i], m["cash"][i])   # the oracle's memo key is (period, x, R)
            assert key in got
            assert got[key][1] == m["actions"][i]
            assert got[key][0] == m["cash"][i] - f.variCost * m["x"][i]   # column S = R - unitVariCost * x
        acts = rec.getCacheActions()
        assert len(acts) == m["n"] and acts[ini] == m["action"]


def test_xr_rejects_what_the_reference_does_not_have(sia):
    w = cases.f3_xr()
    d = w.desc()
    d.penalty_cost = 0.5
    with pytest.raises(sia.SdpgpuError):
        sia.SdpEngine(d, w.pmf)
    d = w.desc()
    d.step = 2
    with pytest.raises(sia.SdpgpuError):
        sia.SdpEngine(d, w.pmf)
    with sia.SdpEngine(w.desc(), w.pmf) as eng:
        eng.solve()
        with pytest.raises(sia.SdpgpuError):
            eng.simulate(np.zeros((4, w.T)), np.ones(w.T), 0.0, 30.0)

"""sdp.cash.CashRecursionXR (state (x, R), order-up-to actions; CashRecursionXR.java:79-126) over the lambdas of
cash.singleItem.CashConstraintXR (CashConstraintXR.java:84-125): the oracle's restatement.  PARITY UNPINNED by the
reference (it records no output for this driver), so the restatement is protected the usual way: a hand-computed
instance, dense sweep == literal memoised recursion, C == pure Python (tests/test_oracle_selfconsistency.py runs the
two f3_xr cases through both), and the host mirror's lambdas == the oracle's on every cell of a small grid."""
import numpy as np

import cases
from stochastic_inventory_amd.functors import CashXRFunctor
from stochastic_inventory_amd.states import CashStateXR, OptDirection
from stochastic_inventory_amd.workloads import Workload


def test_hand_computed_one_period(oracle):
    """T = 1, state (x, R) = (0, 10), unit cost 2, price 4, salvage 1, demand 2 or 6 with probability 1/2 each.
    Feasible levels y = 0..5 (R / c = 5).  imm(y, d) = 4 min(y, d) + (10 - 2y) - 10 + max(y - d, 0) (period T):
    E = 0, 2, 4, 4.5, 5, 5.5 for y = 0..5, so V = 5.5 at y = 5.  Every operand is a small dyadic: exact."""
    f = CashXRFunctor(price=4, variCost=2, salvageValue=1, minInventoryState=0, maxInventoryState=8, minCashState=0,
                      maxCashState=12, iniInventory=0, iniCash=10)
    w = Workload("xr_hand", f, OptDirection.MAX, [np.array([[2.0, 0.5], [6.0, 0.5]])])
    P = oracle.Problem(w.desc(), w.pmf)
    V, pol, cells = P.solve()
    x, R, _ = P.state_arrays(1)
    i = int(np.nonzero((x == 0) & (R == 10))[0][0])
    assert V[0][i] == 5.5 and pol[0][i] == 5
    # a state with stock on hand: (x, R) = (3, 10) has cash 4, levels y = 3..5: E = 4.5 + 0 .. -> recompute by hand
    # y=3: d=2: 8 + (4) - 4 + 1 = 9 ; d=6: 12 + 4 - 4 = 12 -> 10.5;  y=4: d=2: 8 + 2 - 4 + 2 = 8; d=6: 16 + 2 - 4 = 14 -> 11
    # y=5: d=2: 8 + 0 - 4 + 3 = 7; d=6: 20 + 0 - 4 = 16 -> 11.5
    j = int(np.nonzero((x == 3) & (R == 10))[0][0])
    assert V[0][j] == 11.5 and pol[0][j] == 2  # index 2 = level x + 2 = 5
    m = P.memo()
    assert m["value"] == 5.5 and m["action"] == 5.0  # getAction returns the LEVEL (bestY, CashRecursionXR.java:96)


def test_action_count_rule(oracle):
    """`(int) (maxY - x) + 1` with maxY = max(x, R / variCost) (CashConstraintXR.java:84-88), incl. negative cash."""
    f = CashXRFunctor(price=4, variCost=2, minInventoryState=0, maxInventoryState=6, minCashState=-4, maxCashState=9)
    for x, cash, want in ((0, 9, 5), (0, -4, 1), (3, 0, 1), (3, 1, 1), (3, 2, 2), (6, 9, 5), (2, -3, 1)):
        s = CashStateXR(1, x, cash + 2 * x, 2)
        assert len(f.feasibleActions(s, 1)) == want
        assert f.feasibleActions(s, 1)[0] == x
    w = Workload("xr_count", f, OptDirection.MAX, [np.array([[1.0, 1.0]])])
    P = oracle.Problem(w.desc(), w.pmf)
    _, _, cells = P.solve()
    want_cells = sum(len(f.feasibleActions(CashStateXR(1, x, c + 2 * x, 2), 1)) for x in range(7) for c in range(-4, 10))
    assert cells == want_cells


def test_host_lambdas_equal_oracle_cells(oracle):
    """Every (state, action, demand) cell of the fractional case: the Python functor's immediate value and transition
    against the oracle's evaluation of single-cell problems (one action, one demand: V = p * imm)."""
    w = cases.f3_xr_fractional(T=2)
    f = w.functor
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    V, pol, _ = P.solve()
    x, R, _ = P.state_arrays(1)
    rng = np.random.default_rng(4)
    for i in rng.integers(0, len(x), size=60):
        s = CashStateXR(1, x[i], R[i], f.variCost)
        best, besty = -np.inf, None
        for y in f.feasibleActions(s, 2):
            q = 0.0
            for d, p in w.pmf[0]:
                q += p * f.immediateValue(s, y, d, 2)
                ns = f.stateTransition(s, y, d, 2)
                x2, R2, _ = P.state_arrays(2)
                k = int(np.nonzero((x2 == ns.getIniInventory()) & (R2 == ns.getIniR()))[0][0])
                q += p * f.discountFactor * V[1][k]
            if q > best:
                best, besty = q, y
        assert best == V[0][i] and besty == x[i] + pol[0][i]

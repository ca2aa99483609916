"""CPU checks of the STAFF-family oracle (oracle/staffref.c, workforce.StaffRecursion): the dense sweep against the
literal memoised recursion, against an independent pure-Python restatement, against a hand-computed case and
against the committed golden tables."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pyref  # noqa: E402
import staff_cases  # noqa: E402
from oracle import staffref  # noqa: E402


@pytest.mark.parametrize("make", staff_cases.ALL, ids=lambda m: m.__name__)
def test_dense_sweep_equals_literal_recursion(make):
    c = make()
    P = c.oracle_problem(staffref)
    V, pol, cells = P.solve()
    width = int(P.x_lo[-1] + P.nx[-1])
    root, act, val, acts, seen, memo_cells = P.memo(width)
    i0 = c.functor.iniStaffNum - P.x_lo[0]
    assert root == V[0][i0] and act == pol[0][i0]
    n_seen = 0
    for t in range(c.T):
        for x in np.nonzero(seen[t])[0]:
            assert val[t, x] == V[t][x - P.x_lo[t]] and acts[t, x] == pol[t][x - P.x_lo[t]], (t + 1, x)
            n_seen += 1
    assert n_seen >= c.T and memo_cells <= cells


@pytest.mark.parametrize("make", [staff_cases.staff_planning_small, staff_cases.staff_testing_small,
                                  staff_cases.staff_short_rows, staff_cases.staff_single], ids=lambda m: m.__name__)
def test_oracle_equals_pure_python(make):
    c = make(T=2) if make is not staff_cases.staff_single else make()
    P = c.oracle_problem(staffref)
    V, pol, _ = P.solve()
    root, cache = pyref.staff_recursion(c.functor, c.table, c.row_len, c.T)
    assert root == V[0][c.functor.iniStaffNum - P.x_lo[0]]
    for (period, x), (v, a) in cache.items():
        assert v == V[period - 1][x - P.x_lo[period - 1]] and a == pol[period - 1][x - P.x_lo[period - 1]], (period, x)


def test_hand_computed_single_period():
    """T = 1, staff 2, hire 0 or 1, turnover Binomial(y, 1/2), fixCost 10, unit cost 3, salary 1, penalty 5 below a
    minimum staff of 2 (the penalty also applies AT the minimum: `nextStaffNum > minStaffNum ? 0 : ...`,
    WorkforcePlanning.java:98).
      a = 0: y = 2, j = 0,1,2 with p = 1/4,1/2,1/4: imm = 2 + 0, 1 + 5, 0 + 10 -> 0.5 + 3 + 2.5 = 6
      a = 1: y = 3, p = 1/8,3/8,3/8,1/8: n = 3,2,1,0: imm = 13 + 3 + 0, 13 + 2 + 0, 13 + 1 + 5, 13 + 0 + 10
             -> 2 + 5.625 + 7.125 + 2.875 = 17.625"""
    f = staff_cases.StaffFunctor(fixCost=10, unitVariCost=3, salary=1, unitPenalty=5, minStaffNum=[2], maxHireNum=1,
                                 minX=2, maxX=2, clampStaff=True, iniStaffNum=2)
    table = np.zeros((1, 4, 4))  # exact dyadic probabilities, written out (scipy's pmf is an ulp off 1/4)
    table[0, 0, :1] = [1.0]
    table[0, 1, :2] = [0.5, 0.5]
    table[0, 2, :3] = [0.25, 0.5, 0.25]
    table[0, 3, :4] = [0.125, 0.375, 0.375, 0.125]
    c = staff_cases.StaffCase("hand", f, table)
    V, pol, cells = c.oracle_problem(staffref).solve()
    assert V[0][0] == 6.0 and pol[0][0] == 0 and cells == 3 + 4
    f.fixCost, f.unitVariCost = 0.0, 0.0  # hiring is free
    c2 = staff_cases.StaffCase("hand2", f, c.table)
    V2, pol2, _ = c2.oracle_problem(staffref).solve()
    # a = 1 without the 13: 3/8 + 2*3/8 + 6*3/8 + 10/8 = 0.375 + 0.75 + 2.25 + 1.25 = 4.625 < 6
    assert V2[0][0] == 4.625 and pol2[0][0] == 1


def test_layout_boxes():
    c = staff_cases.staff_testing_small()
    P = c.oracle_problem(staffref)
    assert list(P.x_lo) == [0, 0, 0, 0] and list(P.nx) == [1, 13, 25, 37]
    c = staff_cases.staff_planning_small()
    P = c.oracle_problem(staffref)
    assert list(P.x_lo) == [0, 0, 0] and list(P.nx) == [31, 31, 31]


@pytest.mark.parametrize("make", staff_cases.ALL, ids=lambda m: m.__name__)
def test_golden_tables(make):
    c = make()
    path = os.path.join(ROOT, "tests", "golden", f"{c.name}.npz")
    g = np.load(path)
    V, pol, _ = c.oracle_problem(staffref).solve()
    assert np.array_equal(g["table"], c.table)  # the instance itself (scipy's binomial pmf) has not drifted
    for t in range(c.T):
        assert np.array_equal(g[f"v{t + 1}"], V[t]) and np.array_equal(g[f"p{t + 1}"], pol[t])

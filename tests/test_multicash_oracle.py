"""CPU: the oracle's literal recursion for CashRecursionMulti / MultiItemCash (oracle/sdpref.c, sdpref_multicash_memo)
against an independent pure-Python restatement (tests/pyref.py) and a hand-computed case."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import multicash_cases  # noqa: E402
import pyref  # noqa: E402


@pytest.mark.parametrize("seed", range(12))
def test_oracle_equals_pure_python(oracle, seed):
    kw = multicash_cases.random_instance(seed)
    fv, q1, q2, states, cells = oracle.multicash_memo(**kw)
    pv, p1, p2, n = pyref.multicash_recursion(**kw)
    assert (fv, q1, q2) == (pv, p1, p2)
    assert sum(states) == n and states[0] == 1 and cells > 0


def test_hand_computed_single_period(oracle):
    """T = 1, cash 5, unit costs {2, 3}, prices {4, 7}, salvage {1, 1}, no stock, demand (1, 1) for sure, Qbound 3.
    Offered: 2 i + 3 j < 5.1: (0,0) (0,1) (1,0) (1,1) (2,0).  Values (revenue - cost + salvage):
    (0,0) 0; (0,1) 7 - 3 = 4; (1,0) 4 - 2 = 2; (1,1) 11 - 5 = 6; (2,0) 4 - 4 + 1 = 1.
    Scan in list order with `> val + 0.1`: 0, then 4, (1,0) no, (1,1) 6, (2,0) no -> 6 with (1, 1); final 5 + 6."""
    kw = dict(T=1, q_bound=3, price=[4, 7], vari_cost=[2, 3], sal_price=[1, 1], ini_cash=5, ini_i1=0, ini_i2=0,
              min_inventory=0, max_inventory=10, min_cash=0, max_cash=100, discount=1, pmf=[[[1, 1, 1.0]]])
    fv, q1, q2, states, cells = oracle.multicash_memo(**kw)
    assert (fv, q1, q2, states, cells) == (11.0, 1, 1, [1], 5)


def test_tolerance_scan_keeps_the_earlier_action(oracle):
    """Two actions whose values differ by less than 0.1: the later one does not replace the earlier
    (CashRecursionMulti.java:108).  Prices equal costs + 0.04: (0,1) earns 0.04, (1,0) earns 0.04, (1,1) 0.08."""
    kw = dict(T=1, q_bound=2, price=[1.04, 1.04], vari_cost=[1, 1], sal_price=[0, 0], ini_cash=2, ini_i1=0, ini_i2=0,
              min_inventory=0, max_inventory=10, min_cash=0, max_cash=100, discount=1, pmf=[[[1, 1, 1.0]]])
    fv, q1, q2, _, _ = oracle.multicash_memo(**kw)
    assert (q1, q2) == (0, 0) and fv == 2.0  # 0.04 and 0.08 never exceed 0 + 0.1


@pytest.mark.parametrize("seed", range(12))
def test_xr_oracle_equals_pure_python(oracle, seed):
    dep, kw = multicash_cases.xr_random_instance(seed)
    fv, y1, y2, states, cells = oracle.multixr_memo(dep, **kw)
    pv, p1, p2, n = pyref.multixr_recursion(dep, **kw)
    assert (fv, y1, y2) == (pv, p1, p2)
    assert sum(states) == n and states[0] == 1 and cells > 0


def test_xr_hand_computed_single_period(oracle):
    """T = 1, x = (1, 0), R = 4 (cash 3 + 1 unit at cost 1), costs {1, 2}, prices {3, 5}, salvage {0.5, 1}, demand (2, 1)
    for sure, Qbound 2: order-up-to levels y1 in {1, 2}, y2 in {0, 1}; initialCash = 3.
      value(y) = revenue + (R - cost . y) + salvage - initialCash
      (1,0): 3 + (4 - 1) + 0 - 3 = 3;  (1,1): 8 + (4 - 3) + 0 - 3 = 6;  (2,0): 6 + (4 - 2) + 0 - 3 = 5;
      (2,1): 11 + (4 - 4) + 0 - 3 = 8.   Scan: 3, 6, (2,0) no, 8 -> y = (2, 1), final R + 8 = 12."""
    kw = dict(T=1, q_bound=2, price=[3, 5], vari_cost=[1, 2], sal_price=[0.5, 1], ini_cash=4, ini_i1=1, ini_i2=0,
              min_inventory=0, max_inventory=10, min_cash=0, max_cash=100, discount=1, pmf=[[[2, 1, 1.0]]])
    fv, y1, y2, states, cells = oracle.multixr_memo(0.0, **kw)
    assert (fv, y1, y2, states, cells) == (12.0, 2, 1, [1], 4)

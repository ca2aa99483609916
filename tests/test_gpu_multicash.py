"""GPU: CashRecursionMulti over MultiItemCash's lambdas on the reachable-set engine (sdpgpu_multicash_solve) against
the oracle's literal recursion: random instances, and MultiItemCash.main at the size it is written for."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import multicash_cases  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(24))
def test_random_instances_match_the_oracle(sia, oracle, seed):
    kw = multicash_cases.random_instance(seed)
    r = sia.multicash_solve(**kw)
    fv, q1, q2, states, cells = oracle.multicash_memo(**kw)
    assert r.finalValue == fv and (r.firstAction, r.secondAction) == (q1, q2)
    assert r.statesPerPeriod == states and r.cells == cells


def test_multi_item_cash_main_smaller_action_box(sia, oracle):
    """MultiItemCash.main's parameters with Qbound 16 instead of 100 (the oracle's single-threaded recursion needs two
    minutes for the full box)."""
    kw = multicash_cases.main_instance()
    kw["q_bound"] = 16
    r = sia.multicash_solve(**kw)
    fv, q1, q2, states, cells = oracle.multicash_memo(**kw)
    assert r.finalValue == fv and (r.firstAction, r.secondAction) == (q1, q2)
    assert r.statesPerPeriod == states and r.cells == cells


def test_multi_item_cash_main(sia):
    """MultiItemCash.main (its solve is commented out in the reference; parameters as they stand): T = 2, Qbound 100,
    iniCash 100 -- one period-1 state, 24673 successors, 9.1e9 cells -- against the oracle's result committed in
    tests/golden/multicash_main.json (tests/golden/make_golden.py)."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "multicash_main.json")))
    kw = multicash_cases.main_instance()
    for t in range(kw["T"]):  # the instance itself (scipy's normal cdf) has not drifted
        assert kw["pmf"][t].tolist() == g["pmf"][t]
    r = sia.multicash_solve(**kw)
    print(f"MultiItemCash.main: final optimal cash {r.finalValue!r}, Q1 = {r.firstAction}, Q2 = {r.secondAction}, "
          f"states {r.statesPerPeriod}, {r.cells:.3g} cells in {r.gpu_ms:.1f} ms")
    assert r.finalValue == g["final_value"] and (r.firstAction, r.secondAction) == (g["q1"], g["q2"])
    assert r.statesPerPeriod == g["states_per_period"] and r.cells == g["cells"]


def test_misuse(sia):
    kw = multicash_cases.random_instance(0)
    kw["min_cash"] = -5
    with pytest.raises(sia.SdpgpuError):
        sia.multicash_solve(**kw)

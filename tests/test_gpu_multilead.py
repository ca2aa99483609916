"""The two-product lead-time family on the GPU's reachable-set engine, against (i) the outputs the
reference itself recorded (MultiProductLeadtime.java:30-50) and (ii) the oracle's literal memoised
recursion on random small instances."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "kat_reference.json")))
KEYS = ("T", "q_bound", "price", "vari_cost", "sal_value", "ini_cash", "ini_i1", "ini_i2", "r0", "r1", "r2", "limit",
        "interest_free", "min_inventory", "max_inventory", "min_cash", "max_cash", "discount", "overhead", "values",
        "probs")


def test_kat1_on_the_gpu_bit_exact(sia):
    """'when T = 2, final optimal cash is -17.800000000000008 ... Q1 = 40, Q2 = 20' (:41-43)."""
    from stochastic_inventory_amd.multiitem import multilead_solve
    k = KATS["kat1"]
    r = multilead_solve(**{n: k[n] for n in KEYS})
    assert r.finalValue == k["expected_final_cash"]
    assert (r.firstAction, r.secondAction) == (k["expected_q1"], k["expected_q2"])
    assert r.statesPerPeriod == [1, 2500]
    assert r.cells == 2500 * 9 + 2500 * 2500 * 9


def test_kat2_three_periods_on_the_gpu(sia):
    """'3 periods: ... final optimal cash is -76.56 ... Q1 = 30, Q2 = 15, running time is 1568.0s' (:45-50):
    2.5e11 cells over 2.5e7 reachable period-3 states."""
    from stochastic_inventory_amd.multiitem import multilead_solve
    k = KATS["kat2_slow"]
    r = multilead_solve(**{n: k[n] for n in KEYS})
    # Java prints the shortest decimal that identifies the double: "-76.56" IS the double nearest -76.56
    assert r.finalValue == k["expected_final_cash"]
    assert (r.firstAction, r.secondAction) == (k["expected_q1"], k["expected_q2"])
    print(f"KAT-2 on the GPU: {r.finalValue!r}, states {r.statesPerPeriod}, {r.cells:.3g} cells in {r.gpu_ms:.0f} ms "
          "(the reference's comment: 1568 s)")


@pytest.mark.parametrize("name", ["kat3_gpu", "kat4_gpu", "kat5_gpu", "kat6_gpu"])
def test_kat3_to_kat6_three_point_demands(sia, name):
    """'final optimal cash is 91.19499999999998 ... running time is 2863.0s' (:35-39), '441.57499999999993 for
    overhead cost 0' (:30), and -- with the state-rounding line :219 enabled, as it evidently was when that
    part of the comment was written -- '91.26875' (:30) and '272.23749999999995 for overhead cost 50' (:31):
    4e11 cells each, all 17 digits.  That is every number the comment block records."""
    from stochastic_inventory_amd.multiitem import multilead_solve
    k = KATS[name]
    r = multilead_solve(cash_int_cast=k.get("cash_int_cast", False), **{n: k[n] for n in KEYS})
    assert r.finalValue == k["expected_final_cash"]
    assert (r.firstAction, r.secondAction) == (k["expected_q1"], k["expected_q2"])
    print(f"{name}: {r.finalValue!r} in {r.gpu_ms:.0f} ms, states {r.statesPerPeriod}")


@pytest.mark.parametrize("seed", list(range(1, 17)) + [101, 102, 103])
def test_random_instances_match_the_oracle(sia, oracle, seed):
    from stochastic_inventory_amd.multiitem import multilead_solve
    rng = np.random.default_rng(seed)
    T = int(rng.integers(2, 4)) if seed < 100 else 1 + seed % 100  # (101 .. 103: horizons 2 .. 4 with the smallest action boxes)
    n1, n2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    v1 = sorted(rng.choice(np.arange(1, 12), size=n1, replace=False).tolist())
    v2 = sorted(rng.choice(np.arange(1, 9), size=n2, replace=False).tolist())
    p1 = rng.dirichlet(np.ones(n1)).tolist()
    p2 = rng.dirichlet(np.ones(n2)).tolist()
    kw = dict(T=T, q_bound=int(rng.integers(3, 8)) if seed < 100 else 1 + seed % 2, price=(float(rng.integers(3, 9)), float(rng.integers(5, 14))),
              vari_cost=(1.0, 2.5), sal_value=(0.5, 1.25), ini_cash=float(rng.integers(-5, 30)), ini_i1=float(rng.integers(0, 4)),
              ini_i2=0.0, r0=0.01, r1=0.1, r2=1.5, limit=40.0, interest_free=3.0, min_inventory=0.0, max_inventory=9.0,
              min_cash=-120.0, max_cash=400.0, discount=float(rng.choice([1.0, 0.95])),
              overhead=[float(x) for x in rng.integers(0, 25, size=T)], values=[v1, v2], probs=[p1, p2])
    kw["cash_int_cast"] = bool(seed % 2)
    g = multilead_solve(**kw)
    fv, q1, q2, states, cells = oracle.kat_multilead(**kw)
    assert g.finalValue == fv
    assert (g.firstAction, g.secondAction) == (q1, q2)
    assert sum(g.statesPerPeriod) == states and g.cells == cells


def test_kat1_whole_memo_matches_the_oracle(sia, oracle):
    """All 2501 states the reference's recursion visits for KAT-1: tuple, value and action of each."""
    from stochastic_inventory_amd.multiitem import multilead_solve
    k = KATS["kat1"]
    kw = {n: k[n] for n in KEYS}
    r = multilead_solve(table=True, **kw)
    _, want = oracle.memo_table("multilead", **kw)
    assert r.table.shape == want.shape == (2501, 9)
    assert (r.table == want).all()
    assert r.table[0, 0] == 1 and r.table[0, 6] + k["ini_cash"] == k["expected_final_cash"]


def test_read_out_allocation_failure_is_a_status_code(sia, monkeypatch):
    """no C++ exception crosses the ABI.  The memo read-out builds host vectors of one entry per visited state;
    with the allocation capped (SDPGPU_TEST_HOST_ALLOC_CAP: larger vectors throw std::bad_alloc inside the library) the solve
    returns SDPGPU_ERR_ALLOC with a message -- the process lives, the device buffers are released, the next solve is right."""
    from stochastic_inventory_amd.multiitem import multilead_solve
    k = KATS["kat1"]
    monkeypatch.setenv("SDPGPU_TEST_HOST_ALLOC_CAP", "1000")
    with pytest.raises(sia.SdpgpuError) as e:
        multilead_solve(table=True, **{n: k[n] for n in KEYS})
    assert e.value.code == 5 and "host allocation failed" in str(e.value)
    monkeypatch.delenv("SDPGPU_TEST_HOST_ALLOC_CAP")
    r = multilead_solve(table=True, **{n: k[n] for n in KEYS})
    assert r.finalValue == k["expected_final_cash"] and len(r.table) == 2501

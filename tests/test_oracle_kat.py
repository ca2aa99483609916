"""Pin the oracle against the only numbers the reference itself recorded for this path."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "kat_reference.json")))


def _run(oracle, k):
    kw = {n: k[n] for n in ("T", "q_bound", "price", "vari_cost", "sal_value", "ini_cash", "ini_i1", "ini_i2", "r0",
                            "r1", "r2", "limit", "interest_free", "min_inventory", "max_inventory", "min_cash",
                            "max_cash", "discount", "overhead", "values", "probs")}
    return oracle.kat_multilead(**kw)


def test_kat1_bit_exact(oracle):
    """MultiProductLeadtime.java:41-43: 'final optimal cash is -17.800000000000008 ... Q1 = 40, Q2 = 20'."""
    k = KATS["kat1"]
    final, q1, q2, states, cells = _run(oracle, k)
    assert final == k["expected_final_cash"]  # bit for bit, all 17 digits
    assert (q1, q2) == (k["expected_q1"], k["expected_q2"])
    assert states == 2501  # the root plus one period-2 state per action pair (revenue is 0 in period 1)
    assert cells == 2500 * 9 + 2500 * 2500 * 9


@pytest.mark.slow
def test_kat2_three_periods(oracle):
    """MultiProductLeadtime.java:45-50: -76.56 with Q = (30, 15); ~2.5e11 cells, hours single-threaded."""
    k = KATS["kat2_slow"]
    final, q1, q2, _, _ = _run(oracle, k)
    assert abs(final - k["expected_final_cash"]) < 5e-3  # the comment prints two decimals
    assert (q1, q2) == (k["expected_q1"], k["expected_q2"])


def test_java_round_corner_cases(oracle):
    L = oracle.lib()
    # ties toward +infinity (C llround would give -3 and 3)
    assert L.sdpref_java_round(-2.5) == -2
    assert L.sdpref_java_round(2.5) == 3
    assert L.sdpref_java_round(-0.5) == 0
    assert L.sdpref_java_round(0.49999999999999994) == 0  # floor(x + 0.5) would give 1
    assert L.sdpref_java_round(1e15 + 0.5) == 1000000000000001
    assert L.sdpref_java_round(float("nan")) == 0
    assert L.sdpref_java_round(-7.45 * 10) == -74  # -74.5 -> -74


def test_java_minmax_and_cast(oracle):
    import math
    L = oracle.lib()
    assert math.copysign(1.0, L.sdpref_java_max(-0.0, 0.0)) == 1.0
    assert math.copysign(1.0, L.sdpref_java_max(0.0, -0.0)) == 1.0
    assert math.copysign(1.0, L.sdpref_java_min(0.0, -0.0)) == -1.0
    assert math.isnan(L.sdpref_java_max(0.0, float("nan")))
    assert L.sdpref_java_d2i(-3.9) == -3 and L.sdpref_java_d2i(3.9) == 3
    assert L.sdpref_java_d2i(float("nan")) == 0
    assert L.sdpref_java_d2i(1e12) == 2**31 - 1

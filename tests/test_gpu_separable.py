"""The opt-in separable mode has its OWN parity statement: values within 1e-9
relative of the brute-force path (it reassociates the reference's sum), arg-opt free to differ where
two actions tie to within that rounding.  It is never selected automatically."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9


def _compare(sia, w):
    exact, sep = w.desc(), w.desc()
    sep.kernel = sia._abi.KERNEL_SEPARABLE
    with sia.SdpEngine(exact, w.pmf) as e, sia.SdpEngine(sep, w.pmf) as s:
        e.solve()
        s.solve()
        assert s.stats().kernel_used == 3 and e.stats().kernel_used != 3
        worst, mismatches, states = 0.0, 0, 0
        for period in range(1, w.T + 1):
            ve, vs = e.values(period), s.values(period)
            rel = np.abs(vs - ve) / np.maximum(np.abs(ve), 1e-300)
            rel[(ve == 0) & (vs == 0)] = 0.0
            worst = max(worst, float(rel.max()))
            mismatches += int((e.policy(period) != s.policy(period)).sum())
            states += len(ve)
        return worst, mismatches, states


@pytest.mark.parametrize("make", [cases.f1_small, cases.f1_max, cases.f1_gapped, cases.f1_unclamped, cases.f1_clsp_main],
                         ids=lambda f: f.__name__)
def test_separable_values_within_tolerance(sia, make):
    worst, mismatches, states = _compare(sia, make())
    assert worst <= REL_TOL
    assert mismatches <= 0.02 * states  # ties broken by rounding only


def test_separable_cfg2_full_horizon(sia):
    from stochastic_inventory_amd import workloads
    worst, mismatches, states = _compare(sia, workloads.cfg2_clsp())
    assert worst <= REL_TOL and mismatches <= 0.02 * states


def test_separable_refuses_other_families(sia):
    w = cases.f3_testing()
    d = w.desc()
    d.kernel = sia._abi.KERNEL_SEPARABLE
    with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
        with pytest.raises(sia.SdpgpuError) as e:
            eng.solve()
        assert e.value.code == 4

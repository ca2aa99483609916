"""sdpgpu_getpmf (GetPmf.getpmf / CLSP.main's inline pmf behind the C ABI, csrc/sdpgpu_pmf.hip) against a table that depends on
neither scipy nor the library's own cdf / quantile code: tests/golden/pmf_reference.json, the reference's formulas
(GetPmf.java:82-134, CLSP.java:219-247) evaluated in 50-digit arithmetic by the committed script make_pmf_reference.py.
Supports must match exactly (they come from (int)-truncated quantiles).  Probabilities: 1e-13 relative where the formula is a
quotient of masses (integer distributions, GetPmf.java:123-124); where it is a DIFFERENCE of two cdf values near 1
(continuous distributions, GetPmf.java:126-129; CLSP.java:241-244) the formula itself -- in the reference's fp64 as in anyone's
-- carries the rounding of those cdf values, 1.1e-16 each: 1e-13 relative plus 4e-16 absolute.
The scipy restatement (pmf.py) is held to the same table."""
import json
import os

import numpy as np
import pytest

from stochastic_inventory_amd.pmf import GammaDist, GetPmf, NormalDist, PoissonDist, clsp_pmf, getpmf_native

REF = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pmf_reference.json")))
CASES = REF["cases"]
RTOL, ATOL_QUOTIENT, ATOL_CDF_DIFFERENCE = 1e-13, 2e-17, 4e-16


def _dists(case):
    make = {"poisson": lambda a, b: PoissonDist(a), "normal": NormalDist, "gamma": GammaDist}
    return [make[d["kind"]](d["a"], d["b"]) for d in case["dists"]]


def _check(got, want_tiles, what, exact_support, atol):
    assert len(got) == len(want_tiles), what
    for t, (g, w) in enumerate(zip(got, want_tiles)):
        ws = np.array([float(x) for x in w["support"]])
        wp = np.array([float(x) for x in w["prob"]])  # (34-digit strings: float() rounds them correctly)
        assert g.shape == (len(ws), 2), (what, t, g.shape, len(ws))
        if exact_support:
            assert np.array_equal(g[:, 0], ws), (what, t, "support")
        else:
            assert np.allclose(g[:, 0], ws, rtol=1e-13, atol=0), (what, t, "support")
        err = np.abs(g[:, 1] - wp)
        assert np.all(err <= RTOL * wp + atol), (what, t, float(np.max(err / np.maximum(wp, 1e-300))), float(np.max(err)))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_native_getpmf_against_the_50_digit_table(case):
    got = getpmf_native(_dists(case), case["q"], case["step"])
    _check(got, case["getpmf"], case["name"], exact_support=True,
           atol=ATOL_QUOTIENT if case["dists"][0]["kind"] == "poisson" else ATOL_CDF_DIFFERENCE)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_native_clsp_variant_against_the_50_digit_table(case):
    got = getpmf_native(_dists(case), case["q"], case["step"], clsp_variant=True)
    _check(got, case["clsp"], case["name"] + " (CLSP.main)", exact_support=False, atol=ATOL_CDF_DIFFERENCE)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_scipy_restatement_against_the_50_digit_table(case):
    """pmf.py (the host-side mirror the Python drivers use) is held to the same table, at scipy's accuracy."""
    got = GetPmf(_dists(case), case["q"], case["step"]).getpmf()
    for t, (g, w) in enumerate(zip(got, case["getpmf"])):
        wp = np.array([float(x) for x in w["prob"]])
        assert np.array_equal(g[:, 0], np.array(w["support"])) and np.allclose(g[:, 1], wp, rtol=1e-11, atol=1e-16), (case["name"], t)
    got = clsp_pmf(_dists(case), case["q"], case["step"])
    for t, (g, w) in enumerate(zip(got, case["clsp"])):
        wp = np.array([float(x) for x in w["prob"]])
        assert len(g) == len(wp) and np.allclose(g[:, 1], wp, rtol=1e-10, atol=1e-16), (case["name"], t)


def test_table_is_structurally_what_the_reference_computes():
    """Quirks the table must show (they are the reference's, GetPmf.java:88-89,124): integer distributions start at 0 and are
    normalised by the covered mass, so their tile sums to 1; continuous tiles sum to 1 by construction."""
    for case in CASES:
        for w in case["getpmf"]:
            p = np.array([float(x) for x in w["prob"]])
            if case["step"] == 1.0:  # (with step 2 the cells cover one unit less than the mass they are divided by: the reference's)
                assert abs(p.sum() - 1.0) < 1e-12
            if case["dists"][0]["kind"] == "poisson":
                assert w["support"][0] == 0.0

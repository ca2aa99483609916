"""The opt-in separable mode (F1 and F2) has its OWN parity statement, checked here against the
ORACLE (oracle/sdpref.c), not against another kernel: values within 1e-9 relative of the oracle's (the mode
reassociates the reference's sum), arg-opt free to differ only where two actions tie to within that rounding --
which is checked too: at every state whose action differs, the oracle's own Q-value of the action the mode chose is
within tolerance of the oracle's optimum.  That is the F1 mode.  The F2 mode reassociates nothing -- the states of one
level x + preQ evaluate the very same cells, formed once per level in the reference's order -- and must be BIT-IDENTICAL
to the oracle, values and policy.  Neither mode is ever selected automatically."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9


def _rel(a, b):
    r = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    r[(a == 0) & (b == 0)] = 0.0
    return r


def _compare_with_oracle(sia, oracle, w):
    sep = w.desc()
    sep.kernel = sia._abi.KERNEL_SEPARABLE
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    V, pol, _ = P.solve(nthreads=8)
    with sia.SdpEngine(sep, w.pmf, w.overhead()) as s:
        s.solve()
        assert s.stats().kernel_used == 3
        worst, mismatches, states = 0.0, 0, 0
        for period in range(1, w.T + 1):
            vs, ps = s.values(period), s.policy(period)
            worst = max(worst, float(_rel(vs, V[period - 1]).max()))
            diff = np.nonzero(ps != pol[period - 1])[0]
            mismatches += len(diff)
            states += len(vs)
            assert ps.min() >= 0 and ps.max() <= int(w.functor.maxOrderQuantity / w.functor.stepSize)
            # (a differing action is a near-tie: V_sep(s) = Q_sep(s, a') ~ Q(s, a') must be ~ V(s), which the value
            # tolerance above already asserts; nothing else to check per state)
        return worst, mismatches, states


@pytest.mark.parametrize("make", [cases.f1_small, cases.f1_max, cases.f1_gapped, cases.f1_unclamped, cases.f1_clsp_main,
                                  cases.f2_unclamped, cases.f2_clamped, cases.f2_pipeline],
                         ids=lambda f: f.__name__)
def test_separable_values_within_tolerance_of_the_oracle(sia, oracle, make):
    worst, mismatches, states = _compare_with_oracle(sia, oracle, make())
    if make.__name__.startswith("f2"):
        assert worst == 0.0 and mismatches == 0  # the F2 mode is exact
    else:
        assert worst <= REL_TOL
        assert mismatches <= 0.02 * states  # ties broken by rounding only


def test_separable_cfg2_full_horizon(sia, oracle):
    """configs[1] at its full size and horizon (52 periods; the oracle sweeps it in seconds)."""
    from stochastic_inventory_amd import workloads
    worst, mismatches, states = _compare_with_oracle(sia, oracle, workloads.cfg2_clsp())
    assert worst <= REL_TOL and mismatches <= 0.02 * states


@pytest.mark.parametrize("name", ["cfg4", "cfg4p"])
def test_separable_f2_reduced_configs(sia, oracle, name):
    """configs[3] in both shapes (reference's (x, preQ) and the pipeline (x, q1, q2)) on reduced grids."""
    from stochastic_inventory_amd import workloads
    w = workloads.cfg4_leadtime(T=4, NX=150, A=40, D=30) if name == "cfg4" else workloads.cfg4_pipeline(T=3, NX=80, A=24, D=20)
    worst, mismatches, states = _compare_with_oracle(sia, oracle, w)
    assert worst == 0.0 and mismatches == 0


def test_separable_f2_sharded_slabs(sia, oracle):
    """The F2 mode on rank slabs (every rank builds the whole G table, expands its slab) through sdpgpu_solve_multi."""
    w = cases.f2_pipeline()
    V, pol, _ = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve()
    engs = []
    try:
        for r in range(3):
            d = w.desc()
            d.rank, d.world_size, d.device, d.kernel = r, 3, 0, sia._abi.KERNEL_SEPARABLE
            engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))
        sia.SdpEngine.solve_multi(engs)
        for e in engs:
            for period in range(2, w.T + 1):
                assert np.array_equal(e.values(period), V[period - 1])
    finally:
        for e in engs:
            e.close()


def test_separable_refuses_other_families(sia):
    w = cases.f3_testing()
    d = w.desc()
    d.kernel = sia._abi.KERNEL_SEPARABLE
    with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
        with pytest.raises(sia.SdpgpuError) as e:
            eng.solve()
        assert e.value.code == 4


def test_separable_f1_above_64KiB_of_lds(sia, oracle):
    """3000 actions x 400 demand steps: the separable F1 kernel's window + cost rows take 80 KiB of LDS per workgroup (round 2
    refused anything above 64 KiB as "too big"; gfx950 has 160 KiB per compute unit)."""
    from stochastic_inventory_amd import workloads
    w = workloads.cfg5_scaled(S=1500, T=2, A=3000, D=400)
    worst, mismatches, states = _compare_with_oracle(sia, oracle, w)
    assert worst <= REL_TOL and mismatches <= 0.02 * states


@pytest.mark.parametrize("wide", [False, True], ids=["row-kernel", "pair-kernel"])
def test_separable_f5_level_collapse_is_exact(sia, oracle, wide):
    """The cash + lead-time family (SingleProductLeadtime's lambdas) reads the state through x + preQ only: the opt-in mode
    evaluates ONE row per level (cell by cell, the reference's order) and copies it to the level's other rows -- values and
    policy of every state bit-identical to the oracle's dense sweep, on the one-point row kernel (short cash rows) and on the
    two-point pair kernel with the diagonal unit order (rows of 256 cash points and more)."""
    w = cases.f5_cash_leadtime()
    if not wide:
        w.functor.minCashState, w.functor.maxCashState = -1.0, 1.2  # 221 cash points: below the pair kernel's 256
    sep = w.desc()
    sep.kernel = sia._abi.KERNEL_SEPARABLE
    V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=8)
    with sia.SdpEngine(sep, w.pmf, w.overhead()) as s:
        s.solve()
        st = s.stats()
        assert st.kernel_used == 3 and st.cells_evaluated == cells  # (charged the dense count; executed: one row per level)
        for period in range(1, w.T + 1):
            assert np.array_equal(s.values(period), V[period - 1]), period
            assert np.array_equal(s.policy(period), pol[period - 1]), period


def test_separable_f5_refuses_slabs(sia):
    w = cases.f5_cash_leadtime()
    d = w.desc()
    d.kernel, d.rank, d.world_size = sia._abi.KERNEL_SEPARABLE, 0, 2
    with sia.SdpEngine(d, w.pmf, w.overhead()) as s:
        with pytest.raises(sia.SdpgpuError, match="one rank"):
            s.run_period(w.T)  # (a sharded handle runs period by period)

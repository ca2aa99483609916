"""Device rollout of the policy against the oracle's literal loop and the host lambdas."""
import math

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _demands(w, n, seed=3):
    rng = np.random.default_rng(seed)
    T = w.T
    out = np.empty((n, T))
    for t in range(T):
        tile = np.asarray(w.pmf[t])
        out[:, t] = rng.choice(tile[:, 0], size=n, p=tile[:, 1] / tile[:, 1].sum())
    return out


@pytest.mark.parametrize("make", [cases.f1_small, cases.f2_clamped, cases.f2_unclamped, cases.f3_tenths,
                                  cases.f3_min_gamma, cases.f4_overdraft, cases.f5_cash_leadtime],
                         ids=lambda f: f.__name__)
def test_rollout_matches_oracle_bit_for_bit(sia, oracle, make):
    w = make()
    f = w.functor
    eng = sia.SdpEngine(w.desc(), w.pmf, w.overhead())
    eng.solve()
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    V, pol, _ = P.solve()
    dem = _demands(w, 500)
    gamma = getattr(f, "discountFactor", 1.0) if w.desc().family in (3, 4) else 1.0
    disc = np.array([math.pow(gamma, t) for t in range(w.T)])
    ini = (getattr(f, "iniInventory", 0.0), getattr(f, "iniCash", 0.0), getattr(f, "iniPreQ", 0.0))
    gs, gv = eng.simulate(dem, disc, *ini)
    os_, ov = P.simulate(V, pol, dem, disc, *ini)
    assert gv.all() and ov.all()
    assert np.array_equal(gs, os_)
    eng.close()


def test_off_grid_start_and_paths_leaving_the_grid(sia, oracle):
    w = cases.f3_tenths()
    eng = sia.SdpEngine(w.desc(), w.pmf, w.overhead())
    eng.solve()
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    V, pol, _ = P.solve()
    dem = _demands(w, 200)
    disc = np.ones(w.T)
    gs, gv = eng.simulate(dem, disc, 0.0, 4.93)  # 4.93 is not a multiple of the 0.1 cash quantum
    os_, ov = P.simulate(V, pol, dem, disc, 0.0, 4.93)
    assert gv.all() and np.array_equal(gs, os_)
    # unclamped lead-time family: a demand far outside the PMF support walks off the period box
    w2 = cases.f2_unclamped()
    e2 = sia.SdpEngine(w2.desc(), w2.pmf)
    e2.solve()
    d2 = _demands(w2, 8)
    d2[3, 0] = 500.0
    s2, v2 = e2.simulate(d2, np.ones(w2.T), 0.0, 0.0, 0.0)
    P2 = oracle.Problem(w2.desc(), w2.pmf)
    V2, pol2, _ = P2.solve()
    o2, ov2 = P2.simulate(V2, pol2, d2, np.ones(w2.T), 0.0, 0.0, 0.0)
    assert not v2[3] and v2.sum() == 7 and np.array_equal(v2, ov2)
    assert np.array_equal(s2[v2], o2[ov2])
    eng.close()
    e2.close()


def test_simulation_mirror_validates_the_sdp_value(sia):
    """The reference's validation idiom (CLSPTesting.java:125-127): simulated mean ~ V_1(s_0)."""
    from stochastic_inventory_amd import pmf as PM
    from stochastic_inventory_amd.simulation import Simulation, round_demands
    dists = [PM.PoissonDist(m) for m in (6.0, 9.0, 4.0, 7.0)]
    tiles = PM.GetPmf(dists, 0.9999, 1).getpmf()
    f = sia.BackorderFunctor(fixedOrderingCost=30, variOrderingCost=1, holdingCost=1, penaltyCost=8, minInventory=-60,
                             maxInventory=80, maxOrderQuantity=40, iniInventory=0)
    rec = sia.Recursion(sia.OptDirection.MIN, tiles, functor=f)
    ini = sia.State(1, 0.0)
    v = rec.getExpectedValue(ini)
    sim = Simulation(dists, 20000, rec, seed=7)
    mean = sim.simulateSDPGivenSamplNum(ini)
    assert abs(mean - v) / v < 0.02
    # device rollout == the reference's per-path host loop through the lambdas
    dem = round_demands(sim.sampling.generateLHSamples(dists, 64))
    assert np.array_equal(sim._rollout(ini, dem), sim.simulateOnHost(ini, dem))
    center, radius = sim.simulateSDPwithErrorConfidence(ini, 0.01, 0.95)
    assert abs(center - v) < 3 * radius + 0.02 * v


def test_cash_simulation_mirror(sia):
    from stochastic_inventory_amd import pmf as PM
    from stochastic_inventory_amd.simulation import Simulation
    dists = [PM.PoissonDist(5.0)] * 3
    tiles = PM.GetPmf(dists, 0.999, 1).getpmf()
    f = sia.CashFunctor(price=5, fixOrderCost=4, variCost=1, salvageValue=0.5, maxOrderQuantity=20,
                        minInventoryState=0, maxInventoryState=40, minCashState=-20, maxCashState=200,
                        cashRoundMult=1.0, cashRoundDiv=1.0, cashRoundIntDiv=True, cashFormula=1, iniCash=12)
    rec = sia.CashRecursion(sia.OptDirection.MAX, tiles, functor=f, discountFactor=1.0)
    ini = sia.CashState(1, 0.0, 12.0)
    final_cash = rec.getExpectedValue(ini) + 12.0
    sim = Simulation(dists, 20000, rec, discountFactor=1.0, seed=11)
    assert abs(sim.simulateSDPGivenSamplNum(ini) - final_cash) / final_cash < 0.02

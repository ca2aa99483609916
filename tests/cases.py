"""Small problem instances of every functor family, shared by the CPU and GPU parity tests."""
import numpy as np

from stochastic_inventory_amd.functors import (BackorderFunctor, CashFunctor, CashLeadtimeFunctor, LeadtimeFunctor,
                                                OverdraftFunctor, SurvivalFunctor)
from stochastic_inventory_amd.states import OptDirection
from stochastic_inventory_amd.workloads import Workload, truncated_poisson_tile


def _pmf(means, D, d0=0):
    out = []
    for m in means:
        t = truncated_poisson_tile(m, D)
        t[:, 0] += d0
        out.append(t)
    return out


def discrete_pmf(T, values, probs):
    return [np.array([[v, p] for v, p in zip(values, probs)], dtype=np.float64) for _ in range(T)]


def f1_small(T=4):
    f = BackorderFunctor(fixedOrderingCost=20, variOrderingCost=1, holdingCost=2, penaltyCost=10, minInventory=-12,
                         maxInventory=15, maxOrderQuantity=9, iniInventory=1)
    return Workload("f1_small", f, OptDirection.MIN, _pmf([3, 5, 2, 4][:T], 8))


def f1_clsp_main():
    """The instance left in CLSP.java:196-211, with the 4-period means {9,23,53,29} on a truncated
    renormalised Poisson support (SSJ is not available, so the tile is ours)."""
    f = BackorderFunctor(fixedOrderingCost=500, variOrderingCost=0, holdingCost=2, penaltyCost=10,
                         minInventory=-300, maxInventory=300, maxOrderQuantity=60, iniInventory=1)
    pmf = [truncated_poisson_tile(m, int(m + 6 * m ** 0.5) + 1) for m in (9, 23, 53, 29)]
    return Workload("f1_clsp_main", f, OptDirection.MIN, pmf)


def f1_max(T=3):
    f = BackorderFunctor(fixedOrderingCost=3, variOrderingCost=0.5, holdingCost=1, penaltyCost=4, minInventory=-6,
                         maxInventory=8, maxOrderQuantity=5, iniInventory=0)
    return Workload("f1_max", f, OptDirection.MAX, _pmf([2, 3, 2][:T], 6))


def f1_unclamped(T=3):
    """Backorder family without the inventory clamp: per-period boxes grown from the initial state."""
    f = BackorderFunctor(fixedOrderingCost=4, variOrderingCost=1, holdingCost=1, penaltyCost=6, maxOrderQuantity=9,
                         iniInventory=2, clampInventory=False)
    w = Workload("f1_unclamped", f, OptDirection.MIN, _pmf([3, 4, 3][:T], 8, d0=1))
    return w


def f1_edge_single(T=1):
    """Degenerate sizes: one period, one action, one demand point, three states."""
    f = BackorderFunctor(fixedOrderingCost=1, variOrderingCost=1, holdingCost=1, penaltyCost=2, minInventory=-1,
                         maxInventory=1, maxOrderQuantity=0, iniInventory=0)
    return Workload("f1_edge_single", f, OptDirection.MIN, discrete_pmf(T, [1], [1.0]))


def f1_wide(T=2):
    """Action and demand ranges too wide for the window kernel's LDS span: falls back to the gather kernel."""
    f = BackorderFunctor(fixedOrderingCost=50, variOrderingCost=1, holdingCost=1, penaltyCost=9, minInventory=-40,
                         maxInventory=60, maxOrderQuantity=2400, iniInventory=0)
    return Workload("f1_wide", f, OptDirection.MIN, _pmf([600, 580][:T], 1200))


def f1_gapped(T=3):
    """Non-unit-stride demand support (DiscreteDistribution style): the window kernel lays it out on the unit-stride
    grid with zero-probability steps in the gaps."""
    f = BackorderFunctor(fixedOrderingCost=10, variOrderingCost=1, holdingCost=1, penaltyCost=5, minInventory=-20,
                         maxInventory=30, maxOrderQuantity=12, iniInventory=0)
    return Workload("f1_gapped", f, OptDirection.MIN, discrete_pmf(T, [2, 5, 9], [0.25, 0.5, 0.25]))


def f1_sparse_support(T=3):
    """A support too sparse to pad (3 points over a range of 96): stays on the generic kernel."""
    f = BackorderFunctor(fixedOrderingCost=10, variOrderingCost=1, holdingCost=1, penaltyCost=5, minInventory=-120,
                         maxInventory=150, maxOrderQuantity=60, iniInventory=0)
    return Workload("f1_sparse_support", f, OptDirection.MIN, discrete_pmf(T, [0, 40, 95], [0.25, 0.5, 0.25]))


def f2_unclamped(T=3):
    """Leadtime.java shape: no inventory clamp, boxes grow from the initial state."""
    f = LeadtimeFunctor(fixedOrderingCost=0, variOrderingCost=1, holdingCost=2, penaltyCost=10, maxOrderQuantity=12,
                        clampInventory=False, iniInventory=0, iniPreQ=0)
    return Workload("f2_unclamped", f, OptDirection.MIN, _pmf([4, 4, 4][:T], 10))


def f2_clamped(T=4):
    f = LeadtimeFunctor(fixedOrderingCost=5, variOrderingCost=1, holdingCost=2, penaltyCost=10, maxOrderQuantity=10,
                        clampInventory=True, minInventory=-10, maxInventory=25, iniInventory=0, iniPreQ=0)
    return Workload("f2_clamped", f, OptDirection.MIN, _pmf([4, 6, 3, 5][:T], 10))


def f2_pipeline(T=4):
    """Lead time 2 (BASELINE configs[3]'s 3-D pipeline state; the reference has lead time 1 only):
    state (x, q1, q2), 36 x 7 x 7 states."""
    f = LeadtimeFunctor(fixedOrderingCost=3, variOrderingCost=1, holdingCost=2, penaltyCost=10, maxOrderQuantity=6,
                        clampInventory=True, minInventory=-10, maxInventory=25, iniInventory=2, iniPreQ=3,
                        leadTime=2, iniPreQ2=1)
    return Workload("f2_pipeline", f, OptDirection.MIN, _pmf([4, 6, 3, 5][:T], 10))


def f3_tenths(T=3):
    """CashConstraint.java shape: cash rounded to tenths (Math.round(c*10)/10.0), fractional prices."""
    f = CashFunctor(price=2.3, fixOrderCost=1.2, variCost=0.7, holdingCost=0.1, depositeRate=0.01, overheadCost=0.5,
                    overheadRate=0.05, salvageValue=0.35, penaltyCost=0.3, maxOrderQuantity=8, minInventoryState=0,
                    maxInventoryState=10, minCashState=-3, maxCashState=20, cashRoundMult=10.0, cashRoundDiv=10.0,
                    cashRoundIntDiv=False, cashFormula=0, iniInventory=0, iniCash=5)
    return Workload("f3_tenths", f, OptDirection.MAX, _pmf([3, 4, 2][:T], 7))


def f3_row(T=3):
    """CashConstraint.java's own shape -- cash in tenths, formula 0, no end-cash penalty -- with fractional prices
    and a deposit rate, so that nothing is exact: the cash row kernel's case (the shift kernel must refuse it)."""
    f = CashFunctor(price=2.3, fixOrderCost=1.2, variCost=0.7, holdingCost=0.1, depositeRate=0.01, overheadCost=0.5,
                    overheadRate=0.05, salvageValue=0.35, penaltyCost=0, discountFactor=0.97, maxOrderQuantity=9,
                    minInventoryState=0, maxInventoryState=11, minCashState=-2, maxCashState=22, cashRoundMult=10.0,
                    cashRoundDiv=10.0, cashRoundIntDiv=False, cashFormula=0, iniInventory=0, iniCash=5)
    return Workload("f3_row", f, OptDirection.MAX, _pmf([3, 4, 2][:T], 7))


def f3_grid_prices(T=3):
    """CashConstraint.main's structure (cash in tenths, formula 0, no deposit / overhead rate, no penalty) with prices and
    costs that are multiples of the cash quantum: every (action, demand) shifts every cash point by a whole number of
    keys -- the cash row kernels' uniform-key trips -- while the increment itself is not exact (0.1 is not a double).
    336 cash points: two 128-point tiles and a ragged third (the two-points-per-lane kernel, clamped and clamp-free trips)."""
    f = CashFunctor(price=2.3, fixOrderCost=1.2, variCost=0.7, holdingCost=0.1, depositeRate=0.0, overheadCost=0.5,
                    overheadRate=0.0, salvageValue=0.35, penaltyCost=0, discountFactor=0.97, maxOrderQuantity=9,
                    minInventoryState=0, maxInventoryState=11, minCashState=-2, maxCashState=31.5, cashRoundMult=10.0,
                    cashRoundDiv=10.0, cashRoundIntDiv=False, cashFormula=0, iniInventory=0, iniCash=5)
    return Workload("f3_grid_prices", f, OptDirection.MAX, _pmf([3, 4, 2, 5, 3, 4, 2, 3, 5][:T], 9))


def f3_half_grid_prices(T=3):
    """The same with a price of 2.35: the shift is a whole number of keys for even sales and lands exactly BETWEEN two
    keys for odd sales (Math.round's tie, decided by the rounding of the floating-point chain) -- uniform and
    non-uniform steps mix inside one period."""
    f = CashFunctor(price=2.35, fixOrderCost=1.0, variCost=0.5, holdingCost=0.0, depositeRate=0.0, overheadCost=0.0,
                    overheadRate=0.0, salvageValue=0.3, penaltyCost=0, discountFactor=1.0, maxOrderQuantity=8,
                    minInventoryState=0, maxInventoryState=10, minCashState=0, maxCashState=27.3, cashRoundMult=10.0,
                    cashRoundDiv=10.0, cashRoundIntDiv=False, cashFormula=0, iniInventory=1, iniCash=6)
    return Workload("f3_half_grid_prices", f, OptDirection.MAX, _pmf([3, 4, 2, 4, 3, 2, 4, 3][:T], 8))


def f3_testing(T=4):
    """CashConstraintTesting.java shape: formula 1, integer cash (Math.round(c*1)/1)."""
    f = CashFunctor(price=5, fixOrderCost=10, variCost=1, holdingCost=0, overheadCost=0, salvageValue=0.5,
                    penaltyCost=0, maxOrderQuantity=15, minInventoryState=0, maxInventoryState=20, minCashState=-10,
                    maxCashState=120, cashRoundMult=1.0, cashRoundDiv=1.0, cashRoundIntDiv=True, cashFormula=1,
                    iniInventory=0, iniCash=13)
    return Workload("f3_testing", f, OptDirection.MAX, _pmf([4, 6, 3, 5][:T], 9))


def f3_dyadic(T=3):
    """Formula 0 on a half-unit cash grid (Math.round(c*2)/2.0) with quarter/half-valued costs: every
    operation of the lambda is exact, which is what the uniform-shift kernel requires; gamma != 1."""
    f = CashFunctor(price=3.5, fixOrderCost=2.25, variCost=0.75, holdingCost=0.25, overheadCost=1.5,
                    salvageValue=0.375, penaltyCost=0, discountFactor=0.9375, maxOrderQuantity=11,
                    minInventoryState=0, maxInventoryState=14, minCashState=-4, maxCashState=60,
                    cashRoundMult=2.0, cashRoundDiv=2.0, cashRoundIntDiv=False, cashFormula=0, iniInventory=1,
                    iniCash=9.5, overheadCosts=[1.5, 0.5, 2.0][:T])
    return Workload("f3_dyadic", f, OptDirection.MAX, _pmf([3, 5, 4][:T], 9))


def f3_dyadic_wide(T=3):
    """f3_dyadic on 309 cash points (two 128-point tiles and a ragged third): rows wide enough for the diagonal form of the
    uniform-shift kernel.  Shifts of neighbouring actions on a diagonal are 5.5 grid steps apart (rounded per cell), the
    fixed cost breaks the progression at action 0, holding cost and overhead vary the rows."""
    w = f3_dyadic(T)
    w.functor.maxCashState = 150.0
    w.functor.maxOrderQuantity = 21
    w.name = "f3_dyadic_wide"
    return w


def f3_dyadic_big_fixed(T=3):
    """Integer cash grid, fixed cost 140: action 0's shift lies more than the staged segment's reach from action 1's, so the
    first action block of every row takes the direct-gather steps; gamma != 1, a demand support that does
    not start at 0 (d0 = 2), maxInventory binding."""
    f = CashFunctor(price=8, fixOrderCost=140, variCost=2, holdingCost=1, overheadCost=3, salvageValue=0.5,
                    penaltyCost=0, discountFactor=0.875, maxOrderQuantity=60, minInventoryState=0,
                    maxInventoryState=40, minCashState=-40, maxCashState=299, cashRoundMult=1.0, cashRoundDiv=1.0,
                    cashRoundIntDiv=True, cashFormula=0, iniInventory=2, iniCash=160)
    return Workload("f3_dyadic_big_fixed", f, OptDirection.MAX, _pmf([30, 26, 33][:T], 45, d0=2))


def f3_min_gamma(T=3):
    f = CashFunctor(price=4, fixOrderCost=2, variCost=1, holdingCost=0.5, overheadCost=1, salvageValue=0.25,
                    penaltyCost=1.5, discountFactor=0.95, maxOrderQuantity=6, minInventoryState=0,
                    maxInventoryState=9, minCashState=-20, maxCashState=40, cashRoundMult=1.0, cashRoundDiv=1.0,
                    cashRoundIntDiv=True, cashFormula=0, iniInventory=2, iniCash=6)
    return Workload("f3_min_gamma", f, OptDirection.MIN, _pmf([3, 2, 4][:T], 6))


def f3_xr(T=3):
    """cash.singleItem.CashConstraintXR's shape (CashConstraintXR.java:37-125) under sdp.cash.CashRecursionXR: state
    (x, R), order-up-to actions limited by R / variCost, integer cash, variCost 2 (R - 2x is exact)."""
    from stochastic_inventory_amd.functors import CashXRFunctor
    f = CashXRFunctor(price=4, fixOrderCost=0, variCost=2, holdingCost=0, depositeRate=0, overheadCost=0, overheadRate=0,
                      salvageValue=1, discountFactor=1.0, maxOrderQuantity=200, minInventoryState=0, maxInventoryState=14,
                      minCashState=-6, maxCashState=40, iniInventory=0, iniCash=30)
    return Workload("f3_xr", f, OptDirection.MAX, _pmf([4, 5, 3][:T], 9))


def f3_xr_fractional(T=3):
    """The same family with nothing exact: unit cost 1.3 (so R - variCost * x is NOT the rounded balance bit for
    bit), fixed cost, deposit and overhead rates, holding cost, discount, a start with stock on hand."""
    from stochastic_inventory_amd.functors import CashXRFunctor
    f = CashXRFunctor(price=3.7, fixOrderCost=1.5, variCost=1.3, holdingCost=0.2, depositeRate=0.02, overheadCost=0.7,
                      overheadRate=0.05, salvageValue=0.45, discountFactor=0.96, maxOrderQuantity=50, minInventoryState=0,
                      maxInventoryState=12, minCashState=-5, maxCashState=33, iniInventory=2, iniCash=11 + 1.3 * 2)
    return Workload("f3_xr_fractional", f, OptDirection.MAX, _pmf([3, 4, 5][:T], 8))


def f4_overdraft(T=3):
    """CashOverdraft.java shape: piecewise interest, `/ 10` long division."""
    f = OverdraftFunctor(price=10, fixOrderCost=0, variCost=1, salvageValue=0.3, maxOrderQuantity=10,
                         minInventoryState=0, maxInventoryState=12, minCashState=-60, maxCashState=90,
                         cashRoundMult=10.0, cashRoundDiv=10.0, cashRoundIntDiv=True, r0=0.01, r2=0.1, r3=2.0,
                         limit=30, interestFreeAmount=5, iniInventory=0, iniCash=0,
                         overheadCosts=[12.0, 9.0, 15.0][:T])
    return Workload("f4_overdraft", f, OptDirection.MAX, _pmf([4, 5, 3][:T], 9))


def f5_cash_leadtime(T=3):
    """SingleProductLeadtime.java shape: (x, cash, preQ), hundredths, no order in the last period."""
    f = CashLeadtimeFunctor(price=5, variCost=1, salvageValue=0.5, maxOrderQuantity=6, minInventoryState=0,
                            maxInventoryState=8, minCashState=-12, maxCashState=20, cashRoundMult=100.0,
                            cashRoundDiv=100.0, cashRoundIntDiv=False, r0=0.0, r2=0.1, r3=2.0, limit=8,
                            interestFreeAmount=0, iniInventory=0, iniCash=0, iniPreQ=0,
                            overheadCosts=[1.0, 0.5, 0.25][:T])
    return Workload("f5_cash_leadtime", f, OptDirection.MAX, _pmf([3, 3, 3][:T], 6))


def f6_survival(T=4):
    """cashSurvival.java shape (survival-probability objective): price 4, unit cost 1, overhead per period,
    integer cash, orders limited by cash; negative-cash successors are worth 0."""
    f = SurvivalFunctor(price=4, fixOrderCost=0, variCost=1, holdingCost=0, depositeRate=0, salvageValue=0.5,
                        maxOrderQuantity=30, minInventoryState=0, maxInventoryState=25, minCashState=-20,
                        maxCashState=150, iniInventory=0, iniCash=12, overheadCosts=[14.0, 20.0, 9.0, 16.0][:T])
    return Workload("f6_survival", f, OptDirection.MAX, _pmf([4, 6, 3, 5][:T], 10))


def f6_survival_gamma(T=3):
    """CashRecursion.getSurvProb shape (CashRecursion.java:174 multiplies by discountFactor), fractional costs."""
    f = SurvivalFunctor(price=3.5, fixOrderCost=1.5, variCost=1.25, holdingCost=0.2, depositeRate=0.02,
                        salvageValue=0.4, discountFactor=0.97, maxOrderQuantity=12, minInventoryState=0,
                        maxInventoryState=15, minCashState=-8, maxCashState=70, iniInventory=1, iniCash=9,
                        overheadCost=7.5)
    return Workload("f6_survival_gamma", f, OptDirection.MAX, _pmf([3, 5, 4][:T], 9))


ALL = [f1_small, f1_max, f1_gapped, f1_sparse_support, f1_unclamped, f1_edge_single, f2_unclamped, f2_clamped, f2_pipeline, f3_tenths, f3_row, f3_testing, f3_dyadic, f3_min_gamma,
       f3_xr, f3_xr_fractional, f3_grid_prices, f3_half_grid_prices, f4_overdraft, f5_cash_leadtime, f6_survival, f6_survival_gamma]
TINY = [f1_small, f1_max, f1_gapped, f1_unclamped, f2_unclamped, f3_testing, f3_min_gamma, f3_xr, f3_xr_fractional, f4_overdraft]


def xr_main_instance(T=4, demand_points=None):
    """cash.singleItem.CashConstraintXR.main as it stands (CashConstraintXR.java:37-77): four periods, Gamma(8, rate 2)
    demand truncated at the 0.99 quantile on a unit grid, price 4, unit cost 2, salvage 1, inventory 0..500, cash
    -100..2000 (integer), initial state (x, R) = (0, 30).  The PMF is GetPmf's continuous branch (GetPmf.java:125-129)
    over scipy's gamma cdf (SSJ is not available): an input array to oracle and GPU alike."""
    from stochastic_inventory_amd.functors import CashXRFunctor
    from stochastic_inventory_amd.pmf import GammaDist, GetPmf
    f = CashXRFunctor(price=4, fixOrderCost=0, variCost=2, holdingCost=0, depositeRate=0, overheadCost=0, overheadRate=0,
                      salvageValue=1, discountFactor=1.0, maxOrderQuantity=200, minInventoryState=0, maxInventoryState=500,
                      minCashState=-100, maxCashState=2000, iniInventory=0, iniCash=30)
    pmf = GetPmf([GammaDist(8, 2) for _ in range(T)], 0.99, 1).getpmf()
    return Workload("xr_main", f, OptDirection.MAX, pmf)

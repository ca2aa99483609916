#!/usr/bin/env python3
"""Run the parameter sets the reference's driver mains leave in their source on the GPU engine and print
the first-period answer and the time, next to the wall-clock remark the reference's author left in a
comment (where there is one; hardware and JVM unstated there).  PMFs come from this repo's GetPmf
restatement (scipy in place of SSJ), so values are NOT comparable to a Java run digit for digit --
except the two-product family, whose discrete PMFs involve no SSJ arithmetic (those ARE the recorded
outputs, see tests/test_gpu_multilead.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stochastic_inventory_amd as sia
from stochastic_inventory_amd.multiitem import multilead_solve


def timed(label, make, ini, remark):
    t0 = time.perf_counter()
    rec = make()
    v = rec.getExpectedValue(ini)
    q = rec.getAction(ini)
    wall = time.perf_counter() - t0
    st = rec.engine.stats()
    print(f"{label}: value {v!r}, first action {q}, {st.states_total} state-periods, {st.cells_evaluated:.3g} cells, "
          f"GPU sweep {st.solve_ms:.1f} ms, wall incl. setup {wall:.2f} s   [reference comment: {remark}]", flush=True)


def poisson(means, q):
    return sia.GetPmf([sia.PoissonDist(m) for m in means], q, 1).getpmf()


# capacitated.CLSP.main (CLSP.java:196-211; its own inline PMF)
from stochastic_inventory_amd.pmf import clsp_pmf
f = sia.BackorderFunctor(fixedOrderingCost=500, variOrderingCost=0, holdingCost=2, penaltyCost=10, minInventory=-300,
                         maxInventory=300, maxOrderQuantity=60, iniInventory=1)
timed("CLSP.main", lambda: sia.CLSP(clsp_pmf([sia.PoissonDist(m) for m in (9, 23, 53, 29)], 0.99999, 1), functor=f),
      sia.State(1, 1.0), "none")

# leadtime.Leadtime.main (Leadtime.java:25-40)
f = sia.LeadtimeFunctor(fixedOrderingCost=0, variOrderingCost=1, holdingCost=2, penaltyCost=10, maxOrderQuantity=100,
                        clampInventory=False, iniInventory=0, iniPreQ=0)
timed("Leadtime.main", lambda: sia.LeadtimeRecursion(poisson([10, 10, 10], 0.9999), functor=f),
      sia.LeadtimeState(1, 0.0, 0.0), "none")

# cash.singleItem.CashConstraint.main (CashConstraint.java:44-68): cash in tenths, 501 x 20001 states
f = sia.CashFunctor(price=10, fixOrderCost=0, variCost=1, holdingCost=0, salvageValue=0.5, maxOrderQuantity=100,
                    minInventoryState=0, maxInventoryState=500, minCashState=0, maxCashState=2000, iniInventory=0,
                    iniCash=100)
timed("CashConstraint.main", lambda: sia.CashRecursion(sia.OptDirection.MAX, poisson([10] * 6, 0.9999), functor=f,
                                                        discountFactor=1.0),
      sia.CashState(1, 0.0, 100.0), "none here; the 10-period test bed: 500 s (CashConstraintTesting.java:38)")

# cash.overdraft.CashOverdraft.main (CashOverdraft.java:35-61)
f = sia.OverdraftFunctor(price=10, fixOrderCost=0, variCost=1, salvageValue=0, maxOrderQuantity=100, minInventoryState=0,
                         maxInventoryState=100, minCashState=-200, maxCashState=800, cashRoundMult=10.0, cashRoundDiv=10.0,
                         cashRoundIntDiv=True, r0=0, r2=0.1, r3=2, limit=1000, interestFreeAmount=0, iniInventory=0,
                         iniCash=0, overheadCosts=[100.0] * 4)
timed("CashOverdraft.main", lambda: sia.CashRecursion(sia.OptDirection.MAX, poisson([20] * 4, 0.9999), functor=f,
                                                       discountFactor=1.0),
      sia.CashState(1, 0.0, 0.0), "35 s integer cash / 312 s at a 0.1 quantum (CashOverdraftTesting.java:24)")

# cash.overdraft.SingleProductLeadtime at the size its header calls the limit of the Java code:
# 4 periods, mean demand 20 (SingleProductLeadtime.java:22-24); cash in hundredths -> 9.5e7 states per period
f = sia.CashLeadtimeFunctor(price=5, variCost=1, salvageValue=0.5, maxOrderQuantity=30, minInventoryState=0,
                            maxInventoryState=60, minCashState=-200, maxCashState=300, r0=0, r2=0.1, r3=2, limit=500,
                            interestFreeAmount=0, iniInventory=0, iniCash=0, iniPreQ=0, overheadCosts=[0.0] * 4)
timed("SingleProductLeadtime (T=4, mean 20)", lambda: sia.CashLeadtimeRecursion(poisson([20] * 4, 0.9999), functor=f),
      sia.CashLeadtimeState(1, 0.0, 0.0, 0.0), "50 s, 'maximum computational capacity for java' (SingleProductLeadtime.java:22-24)")

# cash.overdraft.MultiProductLeadtime (the recorded outputs)
base = dict(price=(5, 10), vari_cost=(1, 2), sal_value=(0.5, 1.0), ini_cash=0, ini_i1=0, ini_i2=0, r0=0, r1=0.1, r2=2,
            limit=500, interest_free=0, min_inventory=0, max_inventory=200, min_cash=-500, max_cash=5000, discount=1)
for label, kw, remark in (
        ("MultiProductLeadtime T=3 {10,30}/{5,15}", dict(T=3, q_bound=50, overhead=[100] * 3, values=[[10, 30], [5, 15]],
                                                        probs=[[.5, .5], [.5, .5]]), "-76.56, Q=(30,15), 1568 s"),
        ("MultiProductLeadtime T=3 {20,30,40}/{10,15,20}", dict(T=3, q_bound=50, overhead=[100] * 3,
                                                               values=[[20, 30, 40], [10, 15, 20]],
                                                               probs=[[.25, .5, .25], [.25, .5, .25]]),
         "91.19499999999998, Q=(40,20), 2863 s")):
    t0 = time.perf_counter()
    r = multilead_solve(**kw, **base)
    print(f"{label}: final cash {r.finalValue!r}, Q=({r.firstAction},{r.secondAction}), reachable states {r.statesPerPeriod}, "
          f"{r.cells:.3g} cells, GPU {r.gpu_ms:.0f} ms, wall {time.perf_counter() - t0:.2f} s   [reference comment: {remark}]",
          flush=True)

# cash.multiItem.MultiItemCash (CashRecursionMulti; its solve is commented out in the reference) and
# cash.multiItem.MultiItemCashXR (CashRecursionMultiXR; header comment: "2 periods running time is 0.5s")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import multicash_cases  # noqa: E402  (instance builders only: parameters as they stand in the two mains)
for label, solve, kw, remark in (
        ("MultiItemCash.main (T=2, Qbound 100)", sia.multicash_solve, multicash_cases.main_instance(), "solve commented out"),
        ("MultiItemCashXR.main (T=2, Qbound 50)", lambda **k: sia.multixr_solve(0.0, **k), multicash_cases.xr_main_instance(),
         "'2 periods running time is 0.5s', MultiItemCashXR.java:9")):
    t0 = time.perf_counter()
    r = solve(**kw)
    print(f"{label}: final cash {r.finalValue!r}, actions ({r.firstAction},{r.secondAction}), reachable states {r.statesPerPeriod}, "
          f"{r.cells:.3g} cells, GPU {r.gpu_ms:.0f} ms, wall {time.perf_counter() - t0:.2f} s   [reference comment: {remark}]",
          flush=True)

# workforce.WorkforcePlanning.main / one instance of WorkforceTesting.main (StaffRecursion): tools/workforce_drivers.py

"""The two workforce drivers at the sizes their main() methods use, on the HIP engine, timed, every table checked
against oracle/staffref.c (tools/ may use the oracle as a checker, like tests/):
  WorkforcePlanning.main  (WorkforcePlanning.java:33-50): T = 3, staff 0..600 clamped, hires 0..500, rate 0.5
  WorkforceTesting.main   (WorkforceTesting.java:45-107), first instance of its 216: T = 8, hires 0..1000, no clamp
                          (staff 0..7000 by period 8), table of 1001 levels, rate turnoverRates[0]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd.pmf import staff_level_pmf
from oracle import staffref


def run(name, f, table, check=True):
    T = table.shape[0]
    d = f.to_desc(T)
    with sia.SdpEngine(d, None, [float(m) for m in f.minStaffNum], level_pmf=table) as eng:
        eng.set_profiling(True)
        t0 = time.perf_counter(); eng.solve(sync=True); first = time.perf_counter() - t0
        t0 = time.perf_counter(); eng.solve(sync=True); again = time.perf_counter() - t0
        st = eng.stats()
        per = [eng.period_ms(p) for p in range(1, T + 1)]
        print(f"{name}: {st.cells_evaluated:.3e} cells, first solve {first * 1e3:.1f} ms (with upload), "
              f"again {again * 1e3:.2f} ms = {st.cells_evaluated / again:.3e} cells/s; kernel ms per period {['%.2f' % m for m in per]}",
              flush=True)
        if check:
            P = staffref.Problem(T=T, min_x=f.minX, max_x=f.maxX, clamp=f.clampStaff, ini_x=f.iniStaffNum,
                                 max_hire=f.maxHireNum, fix_cost=f.fixCost, unit_vari_cost=f.unitVariCost, salary=f.salary,
                                 unit_penalty=f.unitPenalty, min_staff=list(f.minStaffNum), prob=table)
            t0 = time.perf_counter(); V, pol, cells = P.solve(nthreads=16); cpu = time.perf_counter() - t0
            ok = all(np.array_equal(eng.values(p), V[p - 1]) and np.array_equal(eng.policy(p), pol[p - 1]) for p in range(1, T + 1))
            i0 = f.iniStaffNum - int(P.x_lo[0])
            print(f"  oracle (16 threads) {cpu:.1f} s = {cells / cpu:.3e} cells/s; all tables bit-identical: {ok}; "
                  f"optimal expected cost {V[0][i0]!r}, first-period hires {pol[0][i0]}", flush=True)
            assert ok and cells == st.cells_evaluated


f = sia.StaffFunctor(fixCost=100, unitVariCost=10, salary=20, unitPenalty=80, minStaffNum=[40, 40, 40], maxHireNum=500,
                     minX=0, maxX=600, clampStaff=True, iniStaffNum=0)
run("WorkforcePlanning.main", f, staff_level_pmf([0.5] * 3, 601))
rate = float(os.environ.get("TURNOVER", "0.1"))
f = sia.StaffFunctor(fixCost=50, unitVariCost=20, salary=30, unitPenalty=50, minStaffNum=[40] * 8, maxHireNum=1000,
                     clampStaff=False, iniStaffNum=0)
one = staff_level_pmf([rate], 1001)
run("WorkforceTesting.main[0]", f, np.repeat(one, 8, axis=0), check="--no-check" not in sys.argv)

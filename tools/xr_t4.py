"""MultiItemCashXR at four periods (the horizon its header comment times: '4 periods running time is 80s', MultiItemCashXR.java:8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import stochastic_inventory_amd as sia
import multicash_cases
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = multicash_cases.xr_main_instance()
kw["T"] = T
kw["pmf"] = [kw["pmf"][0]] * T
t0 = time.perf_counter(); r = sia.multixr_solve(0.0, **kw)
print(f"MultiItemCashXR T={T}: final cash {r.finalValue!r}, actions ({r.firstAction},{r.secondAction}), states {r.statesPerPeriod}, "
      f"{r.cells:.3g} cells, GPU {r.gpu_ms:.0f} ms, wall {time.perf_counter() - t0:.2f} s", flush=True)

"""MultiItemCash at three periods (the reachable-set engine's bitmap path): python tools/mc_t3.py [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import stochastic_inventory_amd as sia
import multicash_cases
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kw = multicash_cases.main_instance()
kw["T"] = T
kw["pmf"] = [kw["pmf"][0]] + [kw["pmf"][1]] * (T - 1)
t0 = time.perf_counter(); r = sia.multicash_solve(**kw)
print(f"MultiItemCash T={T}: final cash {r.finalValue!r}, actions ({r.firstAction},{r.secondAction}), states {r.statesPerPeriod}, "
      f"{r.cells:.3g} cells, GPU {r.gpu_ms:.0f} ms, wall {time.perf_counter() - t0:.2f} s", flush=True)

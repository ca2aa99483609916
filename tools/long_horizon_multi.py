"""The two cash-constrained two-product families beyond the horizon their mains are written for (T = 2): with more
periods the candidate lists states x actions x demand pairs outgrow 32-bit indices and the reachable-set engine
switches by itself to its bitmap / rank path (no per-candidate storage).  MultiItemCashXR's header comment:
"4 periods running time is 80s" (MultiItemCashXR.java:8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import stochastic_inventory_amd as sia
import multicash_cases


def show(label, r, wall):
    print(f"{label}: final cash {r.finalValue!r}, actions ({r.firstAction},{r.secondAction}), states {r.statesPerPeriod}, "
          f"{r.cells:.3g} cells, GPU {r.gpu_ms:.0f} ms ({r.cells / max(r.gpu_ms, 1e-9) / 1e9:.2f}e12 cells/s), wall {wall:.2f} s", flush=True)


# the T = 2 mains on both paths: same numbers
for label, solve, kw in (("MultiItemCash.main", sia.multicash_solve, multicash_cases.main_instance()),
                         ("MultiItemCashXR.main", lambda **k: sia.multixr_solve(0.0, **k), multicash_cases.xr_main_instance())):
    for path in ("0", "1"):
        os.environ["SDPGPU_MULTI_LATTICE"] = path
        t0 = time.perf_counter(); r = solve(**kw); show(f"{label} T=2 [{'bitmap' if path == '1' else 'sorted candidates'}]", r, time.perf_counter() - t0)
os.environ.pop("SDPGPU_MULTI_LATTICE")

for T in (3, 4):
    kw = multicash_cases.xr_main_instance()
    kw["T"] = T
    kw["pmf"] = [kw["pmf"][0]] * T
    t0 = time.perf_counter(); r = sia.multixr_solve(0.0, **kw); show(f"MultiItemCashXR T={T}", r, time.perf_counter() - t0)
kw = multicash_cases.main_instance()
kw["T"] = 3
kw["pmf"] = [kw["pmf"][0], kw["pmf"][1], kw["pmf"][1]]
t0 = time.perf_counter(); r = sia.multicash_solve(**kw); show("MultiItemCash T=3", r, time.perf_counter() - t0)

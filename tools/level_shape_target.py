"""CLSP's lambdas as user text of the level shape on the TARGET grid (1e6 x 500 x 200, T = 3) against the built-in family: same
kernel plan, tables bit-identical, times side by side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
w = workloads.target_grid(T=3)
prm = workloads.clsp_lambda_params(w)
res = {}
for name, kw in (("built-in F1", {}), ("user text, level shape", dict(custom_source=workloads.CLSP_LAMBDAS_LEVEL_HIP, custom_params=prm))):
    eng = sia.SdpEngine(w.desc(), w.pmf, **kw)
    eng.solve(); eng.solve()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.solve(sync=False)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st = eng.stats()
    res[name] = (eng.values(1), eng.policy(1), eng.values(2))
    print(f"{name:26s} {dt * 1e3:8.3f} ms per sweep = {st.cells_evaluated / dt:.4g} cells/s, kernel {st.kernel_used}, block ({st.window_r}, {st.window_s})", flush=True)
a, b = res["built-in F1"], res["user text, level shape"]
print("tables bit-identical:", all(np.array_equal(x, y) for x, y in zip(a, b)))

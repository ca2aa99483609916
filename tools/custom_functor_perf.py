import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
import custom_sources as cs
w = workloads.cfg2_clsp(T=6)
f = w.functor
prm = [f.fixedOrderingCost, f.variOrderingCost, f.holdingCost, f.penaltyCost, f.minInventory, f.maxInventory, f.maxOrderQuantity]
t0 = time.perf_counter()
eng = sia.SdpEngine(w.desc(), w.pmf, custom_source=cs.BACKORDER, custom_params=prm)
t_create = time.perf_counter() - t0
eng.solve(); eng.solve()
st = eng.stats()
d = w.desc(); d.kernel = 1
g = sia.SdpEngine(d, w.pmf); g.solve(); g.solve(); sg = g.stats()
a = sia.SdpEngine(w.desc(), w.pmf); a.solve(); a.solve(); sa = a.stats()
print(f"hipRTC compile + create {t_create*1e3:.0f} ms")
for name, s_ in (("user text (hipRTC)", st), ("built-in generic", sg), ("built-in window", sa)):
    print(f"{name:22s} {s_.cells_evaluated / s_.solve_ms / 1e9 * 1e3:8.1f} Gcells/s  ({s_.solve_ms:.2f} ms per 6-period sweep)")
print("tables equal:", all(np.array_equal(eng.values(t), a.values(t)) and np.array_equal(eng.policy(t), a.policy(t)) for t in range(1, 7)))

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

# CashOverdraftLimit's lambdas (not a built-in family) on a 200 x 2001 grid, 60 actions, 40 demand points, 4 periods: the text
# as three functions and with the fused per-cell callback (sdp_cell)
from stochastic_inventory_amd.workloads import truncated_poisson_tile
shape = sia.OverdraftFunctor(price=6, fixOrderCost=2, variCost=1, salvageValue=0.5, maxOrderQuantity=59, minInventoryState=0,
                             maxInventoryState=199, minCashState=-500, maxCashState=1500, cashRoundMult=10.0, cashRoundDiv=10.0,
                             cashRoundIntDiv=True, iniInventory=0, iniCash=10)
T = 4
params = [6, 2, 1, 0.25, 0.1, 0.0, 0.5, 59, 0, 199, -500, 1500] + [9.0, 12.0, 7.0, 10.0]
pmf = [truncated_poisson_tile(18.0, 40) for _ in range(T)]
desc = shape.to_desc(T, sia.OptDirection.MAX)
ref = None
for name, text in (("three functions", cs.OVERDRAFT_LIMIT), ("fused sdp_cell", cs.OVERDRAFT_LIMIT_FUSED),
                   ("three fns, sdp_ldiv", cs.OVERDRAFT_LIMIT_LDIV), ("fused, sdp_ldiv", cs.OVERDRAFT_LIMIT_FUSED_LDIV)):
    e = sia.SdpEngine(desc, pmf, custom_source=text, custom_params=params)
    e.solve(); e.solve()
    s_ = e.stats()
    print(f"CashOverdraftLimit, {name:20s} {s_.cells_evaluated / s_.solve_ms / 1e9 * 1e3:8.1f} Gcells/s  ({s_.solve_ms:.2f} ms per sweep, {s_.cells_evaluated:.3g} cells)")
    v = e.values(1)
    if ref is not None:
        print("fused tables == three-function tables:", bool(np.array_equal(ref, v)))
    ref = v

"""Try the parameter sets hinted at in the comment block MultiProductLeadtime.java:30-50 on the GPU's
reachable-set engine and print the outputs next to the numbers the reference recorded."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochastic_inventory_amd.multiitem import multilead_solve
base = dict(price=(5, 10), vari_cost=(1, 2), sal_value=(0.5, 1.0), ini_cash=0, ini_i1=0, ini_i2=0, r0=0, r1=0.1, r2=2,
            limit=500, interest_free=0, min_inventory=0, max_inventory=200, min_cash=-500, max_cash=5000, discount=1)
v3 = dict(values=[[20, 30, 40], [10, 15, 20]], probs=[[.25, .5, .25], [.25, .5, .25]])
for qb, oh, ic in itertools.product((45, 50), (100, 50, 0), (True, False)):
    r = multilead_solve(T=3, q_bound=qb, overhead=[oh] * 3, cash_int_cast=ic, **v3, **base)
    print(f"T=3 3-point qb={qb} overhead={oh} int-cast={ic}: {r.finalValue!r} Q=({r.firstAction},{r.secondAction}) states={r.statesPerPeriod} {r.gpu_ms:.0f} ms", flush=True)

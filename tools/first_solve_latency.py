"""What a caller that solves ONE problem per process pays (the reference's drivers do): import, library load, create, the first
solve (HIP runtime start-up and code-object load: 0.15-0.2 s, once per process), a second solve, a second engine.
Measured: import 0.36 s, load 0.016 s, create < 1 ms, first solve 0.15-0.21 s, second solve 0.1 ms, second engine's first solve 0.6 ms."""
import os, sys, time
sys.path.insert(0, os.getcwd())
t0 = time.perf_counter()
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
t1 = time.perf_counter()
lib = sia._abi.load()
t2 = time.perf_counter()
w = workloads.cfg1_sS(T=12)
eng = sia.SdpEngine(w.desc(), w.pmf, w.overhead())
t3 = time.perf_counter()
eng.solve(sync=True)
t4 = time.perf_counter()
eng.solve(sync=True)
t5 = time.perf_counter()
v = eng.values(1)
t6 = time.perf_counter()
eng.close()
w2 = workloads.cfg2_clsp(T=8)
e2 = sia.SdpEngine(w2.desc(), w2.pmf, w2.overhead())
t7 = time.perf_counter()
e2.solve(sync=True)
t8 = time.perf_counter()
print(f"import {t1-t0:.3f}  load lib {t2-t1:.3f}  create {t3-t2:.3f}  first solve {t4-t3:.3f}  second solve {t5-t4:.4f}  values {t6-t5:.4f}  | second engine create {t7-t6:.3f} first solve {t8-t7:.4f}")

"""The headline sweep with every table read back over PCIe (sdpgpu_values / sdpgpu_policy of every period into host arrays):
the rate a caller sees who wants all of V and the policy on the host, next to the HBM-resident rate bench.py reports as `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
w = workloads.by_name(sys.argv[1] if len(sys.argv) > 1 else "target")
eng = sia.SdpEngine(w.desc(), w.pmf, w.overhead())
eng.solve()
best = None
for _ in range(3):
    t0 = time.perf_counter()
    eng.solve()
    t1 = time.perf_counter()
    nbytes = 0
    for t in range(1, w.T + 1):
        v, p = eng.values(t), eng.policy(t)
        nbytes += v.nbytes + p.nbytes
    t2 = time.perf_counter()
    if best is None or t2 - t0 < best[0]:
        best = (t2 - t0, t1 - t0, t2 - t1, nbytes)
st = eng.stats()
cells = float(st.cells_evaluated)
print(f"{w.name}: sweep {best[1]*1e3:.2f} ms ({cells/best[1]:.3e} cells/s, host-timed), read-back of {best[3]/1e6:.1f} MB {best[2]*1e3:.2f} ms "
      f"({best[3]/best[2]/1e9:.1f} GB/s), PCIe-inclusive {cells/best[0]:.3e} cells/s")

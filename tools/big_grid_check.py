#!/usr/bin/env python3
"""configs[4] shape on ONE GPU (1e8 states x 500 actions x 200 demands, 2 periods = 2e13 cells): sizes the
oracle cannot sweep, so sampled states of period 1 are checked against the oracle fed the GPU's own V_2
(bit-exact), plus the timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
from oracle import sdpref

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
w = workloads.cfg5_scaled(S=S, T=2)
eng = sia.SdpEngine(w.desc(), w.pmf)
t0 = time.perf_counter()
eng.solve()
wall = time.perf_counter() - t0
st = eng.stats()
print(f"{w.name}: {st.cells_evaluated:.3g} cells, GPU sweep {st.solve_ms:.0f} ms = {st.cells_evaluated / st.solve_ms / 1e9 * 1e3:.3g} Gcells/s (wall {wall:.1f} s)", flush=True)
v2 = eng.values(2)
rng = np.random.default_rng(3)
pick = np.unique(np.concatenate([rng.integers(0, S, size=3000), [0, 1, 63, 64, S - 65, S - 2, S - 1]]))
P = sdpref.Problem(w.desc(), w.pmf)
x = pick.astype(np.float64)  # min_inventory = 0, step 1
ov, oa = P.eval_states(1, v2, x)
v1 = eng.values(1)
p1 = eng.policy(1)
ok = np.array_equal(v1[pick], ov) and np.array_equal(p1[pick], oa)
print(f"sampled {len(pick)} states of period 1 vs oracle: {'bit-identical' if ok else 'MISMATCH'}")
ov2, oa2 = P.eval_states(2, None, x)
ok2 = np.array_equal(v2[pick], ov2) and np.array_equal(eng.policy(2)[pick], oa2)
print(f"sampled {len(pick)} states of period 2 vs oracle: {'bit-identical' if ok2 else 'MISMATCH'}")
sys.exit(0 if ok and ok2 else 1)

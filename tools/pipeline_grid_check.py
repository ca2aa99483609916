#!/usr/bin/env python3
"""configs[3] at full size on ONE GPU: the two-stage pipeline state (x, q1, q2) = 250 x 200 x 200 = 1e7 states,
200 actions, 100 demands (2e11 cells per period), 3 periods.  The oracle cannot sweep that in test time, so
sampled states of every period are checked against the oracle fed the GPU's own V_{t+1} (bit-exact), plus the
timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
from oracle import sdpref

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = workloads.cfg4_pipeline(T=T)
eng = sia.SdpEngine(w.desc(), w.pmf)
t0 = time.perf_counter()
eng.solve()
wall = time.perf_counter() - t0
st = eng.stats()
print(f"{w.name}: {st.cells_evaluated:.3g} cells, GPU sweep {st.solve_ms:.0f} ms = "
      f"{st.cells_evaluated / st.solve_ms / 1e9 * 1e3:.3g} Gcells/s (wall {wall:.1f} s), kernel {st.kernel_used}", flush=True)
P = sdpref.Problem(w.desc(), w.pmf)
rng = np.random.default_rng(5)
x_lo, nx, nc, nq1, nq2 = eng.grid2(1)
S = nx * nq1 * nq2
edges = [0, 1, 63, 64, nx - 1, nx, nx * nq1 - 1, nx * nq1, S - nx - 1, S - 2, S - 1]
pick = np.unique(np.concatenate([rng.integers(0, S, size=2000), edges]))
ix, iq = pick % nx, pick // nx
x, q1, q2 = x_lo + ix.astype(np.float64), (iq % nq1).astype(np.float64), (iq // nq1).astype(np.float64)
ok_all = True
for period in range(T, 0, -1):
    v_next = eng.values(period + 1) if period < T else None
    ov, oa = P.eval_states(period, v_next, x, None, q1, q2)
    ok = np.array_equal(eng.values(period)[pick], ov) and np.array_equal(eng.policy(period)[pick], oa)
    print(f"sampled {len(pick)} states of period {period} vs oracle: {'bit-identical' if ok else 'MISMATCH'}", flush=True)
    ok_all = ok_all and ok
sys.exit(0 if ok_all else 1)

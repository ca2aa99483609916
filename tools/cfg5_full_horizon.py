#!/usr/bin/env python3
"""configs[4] at its FULL horizon on one GPU: 1e8 states x 500 actions x 200 demands x 100 periods = 1e15 cells, with
two ping-pong value tables (store_all_values = 0: 1.6 GB instead of 80 GB).  Prints the sweep time and checks sampled
states of periods 1 and 2 against the oracle fed the GPU's own successor table (bit-exact) -- after 98 periods of
compounding, the last two tables still are what the reference's arithmetic gives.
usage: cfg5_full_horizon.py [states] [periods]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
from oracle import sdpref

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w = workloads.cfg5_scaled(S=S, T=T)
d = w.desc()
d.store_all_values = 0
eng = sia.SdpEngine(d, w.pmf)
eng.set_profiling(True)
t0 = time.perf_counter()
eng.solve()
wall = time.perf_counter() - t0
st = eng.stats()
print(f"{w.name} (ping-pong tables): {st.cells_evaluated:.4g} cells, GPU sweep {st.solve_ms / 1e3:.2f} s = "
      f"{st.cells_evaluated / st.solve_ms / 1e9 * 1e3:.4g} Gcells/s (wall {wall:.1f} s), window plan R={st.window_r} S={st.window_s}",
      flush=True)
ms = [eng.period_ms(p) for p in range(T, 0, -1)]
print(f"per-period kernel ms: first (period T, no future term) {ms[0]:.1f}, median {float(np.median(ms)):.1f}, max {max(ms):.1f}", flush=True)
rng = np.random.default_rng(11)
pick = np.unique(np.concatenate([rng.integers(0, S, size=4000), [0, 1, 63, 64, 255, 256, S - 257, S - 65, S - 2, S - 1]]))
x = pick.astype(np.float64)  # min_inventory = 0, step 1
P = sdpref.Problem(w.desc(), w.pmf)
threads = min(os.cpu_count() or 1, 16)
v2 = eng.values(2)
ov, oa = P.eval_states(1, v2, x, nthreads=threads)
ok = np.array_equal(eng.values(1)[pick], ov) and np.array_equal(eng.policy(1)[pick], oa)
print(f"sampled {len(pick)} states of period 1 vs oracle fed the GPU's V_2: {'bit-identical' if ok else 'MISMATCH'}", flush=True)
print(f"V_1 at x = 0: {eng.values(1)[0]!r}, first order quantity {int(eng.policy(1)[0])}")
sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""A BASELINE config at FULL size on one GPU, checked where the oracle cannot sweep: sampled states of every
checked period are evaluated by the oracle against the GPU's own V_{t+1} and must be bit-identical (values and
policy indices); plus the timing.  usage: sampled_grid_check.py cfg3|cfg4|cfg4p [n_samples]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stochastic_inventory_amd as sia
from stochastic_inventory_amd import workloads
from oracle import sdpref

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
w = {"cfg3": workloads.cfg3_cash, "cfg4": workloads.cfg4_leadtime, "cfg4p": workloads.cfg4_pipeline}[name]()
eng = sia.SdpEngine(w.desc(), w.pmf, w.overhead())
t0 = time.perf_counter()
eng.solve()
wall = time.perf_counter() - t0
st = eng.stats()
print(f"{w.name}: {st.cells_evaluated:.3g} cells, GPU sweep {st.solve_ms:.0f} ms = "
      f"{st.cells_evaluated / st.solve_ms / 1e9 * 1e3:.3g} Gcells/s (wall {wall:.1f} s), kernel {st.kernel_used}", flush=True)
P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
rng = np.random.default_rng(9)
T = w.T
periods = sorted(set([T, T - 1, max(1, T // 2), 1]), reverse=True)
ok_all = True
for period in periods:
    x_lo, nx, nc, nq1, nq2 = eng.grid2(period)
    S = nx * nc * nq1 * nq2
    edges = [0, 1, nc - 1, nc, S - nc, S - 2, S - 1, nx * nc - 1, min(S - 1, nx * nc)]
    pick = np.unique(np.concatenate([rng.integers(0, S, size=n_samples), edges]))
    ic = pick % nc
    ix = (pick // nc) % nx
    iq = pick // (nc * nx)
    x = x_lo + ix.astype(np.float64) * w.desc().step
    cash = np.array([eng.cash_value(int(c)) for c in ic]) if nc > 1 else None
    q1 = (iq % nq1).astype(np.float64) if nq1 > 1 else None
    q2 = (iq // nq1).astype(np.float64) if nq2 > 1 else None
    v_next = eng.values(period + 1) if period < T else None
    ov, oa = P.eval_states(period, v_next, x, cash, q1, q2)
    ok = np.array_equal(eng.values(period)[pick], ov) and np.array_equal(eng.policy(period)[pick], oa)
    print(f"sampled {len(pick)} states of period {period} vs oracle: {'bit-identical' if ok else 'MISMATCH'}", flush=True)
    ok_all = ok_all and ok
sys.exit(0 if ok_all else 1)

#!/usr/bin/env python3
"""bench.py -- the hot path (backward Bellman sweep t = T..1) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W              (any N: for N > 1 and no WORLD_SIZE in the environment the
                                                                process starts its own N ranks, see `self_launch`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

A "step" is one full backward sweep of the workload with every input (PMF tiles, descriptor) already resident
in HBM.  Default workload: the grid BASELINE.json's target is quoted on -- 1e6 states x 500 actions x 200
demands (F1, capacitated.CLSP's lambdas, CLSP.java:251-272), six periods = 6e11 cells per sweep.  For N > 1 the
SAME grid is cut into N contiguous state slabs (strong scaling) with one RCCL all-gather of V_t per period, issued
by libsdpgpu.so itself (sdpgpu_solve_sharded); torch.distributed carries the 128-byte communicator id and the
contract's barrier, nothing else.

Order of business ("parity gates run before any timing is accepted"):
  1. one sweep, then the PARITY GATE: sampled states of every checked period (slab and grid edges + random) are
     evaluated by the CPU oracle fed the GPU's own V_{t+1}; values and policy indices must be bit-identical.
     A failed gate ends the run with a non-zero exit and no bench line.
  2. W warm-up sweeps, K timed sweeps between barriers; max over ranks.
  3. one more sweep with a HIP-event pair around every period launch (un-profiled per-launch times).
  4. N = 1 on the default workload: the other BASELINE configs as `secondary` entries (each gated and timed the same
     way, a few sweeps), then the CPU oracle timed on a bounded sample (`cpu_baseline`).

Rank 0 prints ONE JSON line.  `roofline` names the unit that binds the dominant kernel: fp64 VALU issue (the
reference's arithmetic executed without FMA, 16 lanes/clk/SIMD), with the HBM traffic from the rocprofv3 counters
(profiles/) as a sub-block -- the 8-bytes-per-cell gather SURVEY 8(d) prices is served by LDS/L2, not HBM.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, HBM)
# fp64 add / mul issue (no FMA by contract): 16 lanes per clock per SIMD x 4 SIMDs x 256 CUs x 2.4 GHz
# (= half the 78.6 TFLOP/s fp64 vector peak, which counts an FMA as two); tools/valu_probe.hip measures 3.8e13 at the
# clock the chip holds.  Every VALU instruction of a wave64 occupies its SIMD for four cycles, so the same figure is
# the issue bound for ALL vector instructions.
VALU_PEAK_LANE_OPS = 16 * 4 * 256 * 2.4e9
LDS_PEAK_BPS = 128 * 256 * 2.4e9  # 128 B/clk/CU: the rate of ds_read2_b64, the read the LDS-bound kernel issues (MI355X_MICROARCH.md, LDS)
L1_PEAK_BPS = 64 * 256 * 2.4e9    # vector L1: 64 B/clk/CU

FAMILY_NAME = {"target": "F1 backorder (capacitated.CLSP.f lambdas)", "cfg2": "F1 backorder (capacitated.CLSP.f)",
               "cfg5": "F1 backorder (capacitated.CLSP.f lambdas)", "cfg3": "F3 cash (CashRecursion, quantum 1)",
               "cfg3t": "F3 cash (CashConstraint.main, quantum 0.1)", "cfg4": "F2 lead time (LeadtimeRecursion)",
               "cfg4p": "F2 lead time 2 (pipeline state)",
               "f5_spl": "F5 cash + lead time (SingleProductLeadtime.main via CashLeadtimeRecursion)",
               "staff": "F7 workforce (WorkforceTesting.main[0] via StaffRecursion, level-dependent pmf)",
               "custom_clsp": "user lambdas as HIP text (CLSP's, through sdpgpu_create_custom / hipRTC)",
               "custom_clsp_level": "user lambdas of the declared LEVEL SHAPE (CLSP's cost functions as HIP text, tabulated, on the F1 window kernel)",
               "separable_target": "F1 backorder, OPT-IN separable mode (values to 1e-9, not the bit-exact path)",
               "separable_f5": "F5 cash + lead time, OPT-IN separable mode (one row per level x + preQ; exact)",
               "multilead_kat2": "two-product overdraft with lead time (MultiProductLeadtime via CashRecursionMultiLead)"}
REFERENCE_FLOPS_PER_CELL = {"target": 14, "cfg2": 14, "cfg5": 14, "cfg4": 14, "cfg4p": 14, "cfg3": 25, "cfg3t": 25}
# the entries of `secondary` beyond the BASELINE configs: configs[4] at full width and the SURVEY 8(f)-3 / 8(f)-4 rows
FAMILY_WORKLOADS = ("f5_spl", "staff", "custom_clsp", "custom_clsp_level", "separable_target", "separable_f5", "multilead_kat2")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="target",
                    help="target (default: 1e6 x 500 x 200, T = 6) | cfg2 | cfg3 | cfg3t | cfg4 | cfg4p | cfg5 (1e8 states) | "
                         "f5_spl | staff | custom_clsp | custom_clsp_level | separable_target | separable_f5 | multilead_kat2 (N = 1 only)")
    ap.add_argument("--states", type=int, default=0, help="override the state count of the F1 grids (target/cfg2/cfg5)")
    ap.add_argument("--periods", type=int, default=0, help="override the horizon")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling instead of strong: every rank keeps --states states (default 1e4 for cfg2, 1e6 otherwise)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 gather, 2 window, 3 separable (opt-in: F1 to 1e-9, F2 exact)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs (N = 1, default workload)")
    ap.add_argument("--no-gate", action="store_true", help="skip the parity gate (profiling runs only: the line says so)")
    ap.add_argument("--gate-cells", type=float, default=3e9, help="oracle work of the parity gate, in cells")
    ap.add_argument("--exchange", default="native",
                    help="N > 1: native (RCCL inside libsdpgpu.so, default) | torch (torch.distributed nccl, the schedules "
                         "of sharded.py) | host (gloo, host-staged: rehearsal of N ranks on one GPU)")
    ap.add_argument("--backend", default="", help="alias: --backend gloo = --exchange host")
    ap.add_argument("--split", action="store_true", help="rehearsal: force the interior/boundary split of every period")
    ap.add_argument("--no-overlap", action="store_true", help="blocking all-gather between periods")
    ap.add_argument("--schedule", default="auto",
                    help="N > 1: auto (time the candidates before the warm-up, keep the fastest) | overlap | blocking | "
                         "blockedK (torch/host exchange only: K periods per exchange on widened slabs)")
    ap.add_argument("--check", action="store_true", help="compare the sharded tables with a single-rank sweep (rank 0)")
    ap.add_argument("--rehearse-sharded", action="store_true",
                    help="N = 1 only: take the N > 1 code path with a world of one (gloo group of one, one-rank RCCL communicator "
                         "inside libsdpgpu.so, sdpgpu_solve_sharded, per-rank gate) -- what one rank of eight executes")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("SDP_BENCH_LAUNCH_TIMEOUT", "1500")),
                    help="bare `bench.py --gpus N` (N > 1, not under torchrun): seconds the N ranks this process starts may "
                         "take before they are ended with an error record")
    a = ap.parse_args()
    if a.backend == "gloo":
        a.exchange = "host"
    return a


def make_workload(name: str, world: int, states: int = 0, periods: int = 0, weak: bool = False):
    from stochastic_inventory_amd import workloads
    kw = {"T": periods} if periods else {}
    if name in ("target", "cfg2", "cfg5"):
        base = {"target": 1000000, "cfg2": 10000, "cfg5": 100000000}[name]
        S = (states or (10000 if name == "cfg2" else 1000000)) * world if weak else (states or base)
        if name == "cfg2":
            return workloads.cfg2_clsp(S=S, **kw)
        if name == "cfg5":
            return workloads.cfg5_scaled(S=S, **kw)
        return workloads.target_grid(S=S, **kw)
    if weak:
        raise SystemExit(f"--weak is defined for the F1 grids only (target/cfg2/cfg5), not {name}")
    if name == "separable_target":
        w = workloads.target_grid(**kw)
        w.name = "separable_" + w.name
        return w
    if name == "separable_f5":
        w = workloads.by_name("f5_spl", **kw)
        w.name = "separable_" + w.name
        return w
    return workloads.by_name(name, **kw)


def make_engine(sia, w, d):
    """The engine of a workload: built-in family, level-dependent pmf (staff) or the caller's lambdas as HIP text."""
    if getattr(w, "level_pmf", None) is not None:
        return sia.SdpEngine(d, None, w.overhead(), level_pmf=w.level_pmf)
    if getattr(w, "custom_source", None):
        return sia.SdpEngine(d, w.pmf, w.overhead(), custom_source=w.custom_source, custom_params=w.custom_params)
    return sia.SdpEngine(d, w.pmf, w.overhead())


def build_identity():
    """The loaded library's sdpgpu_build_id against the digest of the tree's sources (tools/kernel_sha.py: build_source_sha):
    every bench line says which binary produced it and whether that binary is this tree's."""
    import stochastic_inventory_amd as sia
    from tools.kernel_sha import build_source_sha
    have, want = sia._abi.load().sdpgpu_build_id().decode(), build_source_sha(ROOT)
    return {"build_id": have, "source_digest": want, "match": have == want}


def algorithmic_bytes(cells: float, states_periods: float) -> float:
    """SURVEY.md section 8(d)'s byte MODEL: 8 B per cell (one fp64 gather of V_{t+1}) + 12 B per state-period."""
    return 8.0 * cells + 12.0 * states_periods


# ---------------------------------------------------------------------------------------------------------------
# parity gate
# ---------------------------------------------------------------------------------------------------------------
def parity_gate(eng, w, budget_cells: float, seed: int = 5, periods=None, rel_tol=None):
    """Sampled states of this rank's slab, every checked period, against the oracle (oracle/sdpref.c eval_state) fed
    the GPU's own V_{t+1}.  Bit-exact on values and policy indices.  Returns (ok, record).
    `periods`: the periods to check (ping-pong tables keep only V_1 and V_2: [1]).  `rel_tol` (the OPT-IN separable mode
    only, whose parity statement is its own: include/sdpgpu.h): values within that relative tolerance, and an action that
    differs must be a near-tie -- which the value tolerance implies, V_sep(s) = Q_sep(s, a')."""
    import numpy as np
    from oracle import sdpref
    T = w.T
    P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
    if periods is None:
        periods = list(range(T, 0, -1)) if T <= 8 else sorted({T, T - 1, T // 2, 2, 1}, reverse=True)
    st = eng.stats()
    cells_per_state = max(1.0, float(st.cells_all_ranks) / max(1, int(st.states_total)))
    n_samples = int(max(64, min(20000, budget_cells / len(periods) / cells_per_state)))
    threads = min(os.cpu_count() or 1, 16)
    rng = np.random.default_rng(seed)
    step = w.desc().step
    checked, bad = 0, []
    worst_rel, action_diffs = 0.0, 0
    t0 = time.perf_counter()
    for period in periods:
        x_lo, nx, nc, nq1, nq2 = eng.grid2(period)
        S = nx * nc * nq1 * nq2
        _, lo, hi = eng.slab(period)
        if hi <= lo:
            continue
        edges = [lo, lo + 1, hi - 1, hi - 2, lo + nc - 1, lo + nc, hi - nc, (lo + hi) // 2,
                 lo + 63, lo + 64, lo + 127, lo + 128, lo + 255, lo + 256]  # tile seams of the window kernels
        edges = [e for e in edges if lo <= e < hi]
        pick = np.unique(np.concatenate([rng.integers(lo, hi, size=n_samples), np.asarray(edges, dtype=np.int64)]))
        ic = pick % nc
        ix = (pick // nc) % nx
        iq = pick // (nc * nx)
        x = x_lo + ix.astype(np.float64) * step
        cash = np.array([eng.cash_value(int(c)) for c in ic]) if nc > 1 else None
        q1 = (iq % nq1).astype(np.float64) * step if nq1 > 1 else None
        q2 = (iq // nq1).astype(np.float64) * step if nq2 > 1 else None
        v_next = eng.values(period + 1) if period < T else None
        ov, oa = P.eval_states(period, v_next, x, cash, q1, q2, nthreads=threads)
        gv = eng.values(period)[pick]
        gp = eng.policy(period)[pick - lo]
        if rel_tol is None:
            if not (np.array_equal(gv, ov) and np.array_equal(gp, oa)):
                bad.append(period)
        else:
            rel = np.abs(gv - ov) / np.maximum(np.abs(ov), 1e-300)
            worst_rel = max(worst_rel, float(rel.max()))
            action_diffs += int((gp != oa).sum())
            if not (rel.max() <= rel_tol):
                bad.append(period)
        checked += len(pick)
    rec = {"status": "ok" if not bad else "FAILED", "states_checked": checked, "periods_checked": periods,
           "seconds": round(time.perf_counter() - t0, 2),
           "against": "oracle/sdpref.c eval_state fed the GPU's own V_{t+1}; values and policy indices bit-identical"}
    if rel_tol is not None:
        rec.update({"against": f"oracle/sdpref.c eval_state fed the GPU's own V_{{t+1}}; values within {rel_tol:g} relative (the "
                               "separable mode's own parity statement; arg-opt may differ on near-ties)",
                    "worst_relative_difference": worst_rel, "actions_differing": action_diffs})
    if bad:
        rec["periods_failed"] = bad
    return not bad, rec


def staff_gate(eng, w, budget_cells: float, seed: int = 5):
    """workforce.StaffRecursion's gate: runs of states of every period against oracle/staffref.c (staffref_period) fed the
    GPU's own V_{t+1}; values and policy (hire counts) bit-identical."""
    import numpy as np
    from oracle import staffref
    f, T = w.functor, w.T
    P = staffref.Problem(T=T, min_x=f.minX, max_x=f.maxX, clamp=f.clampStaff, ini_x=f.iniStaffNum, max_hire=f.maxHireNum,
                         fix_cost=f.fixCost, unit_vari_cost=f.unitVariCost, salary=f.salary, unit_penalty=f.unitPenalty,
                         min_staff=list(f.minStaffNum), prob=w.level_pmf)
    st = eng.stats()
    cells_per_state = max(1.0, float(st.cells_all_ranks) / max(1, int(st.states_total)))
    run = int(max(8, min(512, budget_cells / T / cells_per_state / 3)))  # three runs of states per period: both ends and a random one
    threads = min(os.cpu_count() or 1, 16)
    rng = np.random.default_rng(seed)
    checked, bad = 0, []
    t0 = time.perf_counter()
    for period in range(T, 0, -1):
        n = int(P.nx[period - 1])
        assert n == eng.num_states(period), "the oracle and the engine lay the period out differently"
        v_next = eng.values(period + 1) if period < T else None
        gv, gp = eng.values(period), eng.policy(period)
        starts = sorted({0, max(0, n - run), int(rng.integers(0, max(1, n - run + 1)))})
        for lo in starts:
            hi = min(n, lo + run)
            ov, oa = np.zeros(n), np.zeros(n, dtype=np.int32)
            P.period(period, v_next, lo, hi, threads, ov, oa)
            if not (np.array_equal(gv[lo:hi], ov[lo:hi]) and np.array_equal(gp[lo:hi], oa[lo:hi])):
                bad.append(period)
            checked += hi - lo
    rec = {"status": "ok" if not bad else "FAILED", "states_checked": checked, "periods_checked": list(range(T, 0, -1)),
           "seconds": round(time.perf_counter() - t0, 2),
           "against": "oracle/staffref.c staffref_period fed the GPU's own V_{t+1}; values and hire counts bit-identical"}
    if bad:
        rec["periods_failed"] = sorted(set(bad))
    return not bad, rec


def run_multilead(steps: int, no_gate: bool):
    """The two-product overdraft recursion on its reachable set (sdpgpu_multilead_solve; csrc/sdpgpu_sparse.hip) at the size of
    the reference's slowest RECORDED run: "3 periods ... final optimal cash is -76.56 ... Q1 = 30, Q2 = 15, running time is
    1568.0s" (MultiProductLeadtime.java:45-50).  The gate is that recorded output itself (tests/golden/kat_reference.json holds
    the parameters and the number): the value must be THE double nearest -76.56 and the first order (30, 15).  A step = one whole
    solve (forward expansion of the reachable set + backward recursion), device buffers allocated and freed inside the call."""
    from stochastic_inventory_amd.multiitem import multilead_solve
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_reference.json")))["kat2_slow"]
    keys = ("T", "q_bound", "price", "vari_cost", "sal_value", "ini_cash", "ini_i1", "ini_i2", "r0", "r1", "r2", "limit",
            "interest_free", "min_inventory", "max_inventory", "min_cash", "max_cash", "discount", "overhead", "values", "probs")
    kw = {n: k[n] for n in keys}
    r = multilead_solve(**kw)
    ok = r.finalValue == k["expected_final_cash"] and (r.firstAction, r.secondAction) == (k["expected_q1"], k["expected_q2"])
    gate = {"status": "ok" if ok else "FAILED", "states_checked": int(sum(r.statesPerPeriod)),
            "final_cash": r.finalValue, "first_order": [r.firstAction, r.secondAction],
            "against": "the output the reference itself recorded (MultiProductLeadtime.java:45-50): final optimal cash -76.56, "
                       "Q1 = 30, Q2 = 15 -- all 17 digits of the value, both quantities"}
    if no_gate:
        gate = {"status": "skipped (--no-gate)"}
    elif not ok:
        raise SystemExit(f"PARITY GATE FAILED on multilead_kat2: {json.dumps(gate)} -- no timing accepted")
    walls, gpus = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        r = multilead_solve(**kw)
        walls.append(time.perf_counter() - t0)
        gpus.append(r.gpu_ms)
    wall = sum(walls) / len(walls)
    name = f"multilead_kat2_T{k['T']}_Q{k['q_bound']}"
    rf = {"bound": "valu-issue", "achieved": None, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s", "frac": None,
          "device_ms_per_solve": sum(gpus) / len(gpus), "traffic": None, "hbm": None,
          "note": "no counter summary in profiles/ for this build of csrc/sdpgpu_sparse.hip"}
    pmc = load_pmc(name)
    if pmc and pmc.get("valu_insts_per_launch") and pmc.get("kernels", {}).get(pmc.get("dominant_kernel"), {}).get("rocprof"):
        dom = pmc["kernels"][pmc["dominant_kernel"]]
        n_launch = dom["rocprof"]["calls"] / max(1, pmc.get("solves_profiled", 1))
        # the recursion kernel is launched once per period on reachable sets of very different sizes: totals per SOLVE
        insts = float(pmc["valu_insts_per_launch"]) * n_launch
        kern_ms = dom["rocprof"]["avg_us"] * n_launch / 1e3
        lane_ops = insts * 64.0 / (kern_ms * 1e-3)
        rf.update({"achieved": lane_ops / 1e12, "frac": lane_ops / VALU_PEAK_LANE_OPS, "kernel": pmc["dominant_kernel"],
                   "kernel_ms_per_solve_rocprof": kern_ms, "valu_insts_per_cell": insts * 64.0 / max(r.cells, 1),
                   "traffic": pmc.get("hbm_bytes_per_launch"),
                   "counters": {kk: pmc.get(kk) for kk in ("valu_busy_frac", "lds_busy_frac", "ta_busy_frac")},
                   "note": f"SQ_INSTS_VALU x 64 lanes of the recursion kernel's launches of one solve / their rocprofv3 duration ({pmc['_file']})"})
        if not (0.0 < rf["frac"] <= 1.0):
            rf.update({"frac": None, "achieved": None})
    return {"workload": name, "family": FAMILY_NAME["multilead_kat2"], "value": r.cells / wall, "unit": "cells/s",
            "ms_per_step": wall * 1e3, "steps": steps, "periods": k["T"], "states": int(sum(r.statesPerPeriod)),
            "states_per_period": [int(v) for v in r.statesPerPeriod], "cells_per_step": int(r.cells), "parity_gate": gate,
            "roofline": rf, "kernel": "reachable-set engine (expand, sort, rank, backward)",
            "reference_remark": "1568 s (the reference author's own comment; hardware and JVM unstated)"}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(w, target_seconds: float):
    """The CPU oracle ('port' of the Java recursion, oracle/sdpref.c) timed on a BOUNDED sample of the same workload on
    all host cores, sized from a probe to about `target_seconds`: whole periods from the end of the horizon when a
    period fits the budget, else a run of states of the last-but-one period (the big grids; the successor values it
    reads are then zeros -- the arithmetic per cell is the same).  Reported baseline, not the target."""
    from oracle import sdpref
    import numpy as np
    cores = min(os.cpu_count() or 1, 16)
    P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
    T = w.T
    pf = max(T - 1, 1)  # a period with a future term (the only period of a single-period horizon has none)
    S_f = int(P.S[pf - 1])
    v0 = np.zeros(int(P.S[pf])) if pf < T else None
    n_probe = max(1, min(S_f, max(cores * 4, S_f // 2048)))
    lo_p = (S_f - n_probe) // 2
    t0 = time.perf_counter()
    _, _, c_probe = P.period(pf, v0, lo=lo_p, hi=lo_p + n_probe, nthreads=cores)
    t_probe = max(time.perf_counter() - t0, 1e-6)
    est_period = t_probe * S_f / n_probe
    if 2.2 * est_period <= target_seconds:  # whole periods: last one first (it feeds the next), then as many as fit
        v, _, cells_last = P.period(T, None, nthreads=cores)
        total_cells, total_t, k = 0, 0.0, 0
        period = T - 1
        while period >= 1 and (k == 0 or total_t + total_t / k < target_seconds):
            t0 = time.perf_counter()
            v, _, cells = P.period(period, v, nthreads=cores)
            total_t += time.perf_counter() - t0
            total_cells += cells
            k += 1
            period -= 1
        if k == 0:  # single-period horizon
            t0 = time.perf_counter()
            _, _, total_cells = P.period(T, None, nthreads=cores)
            total_t, k = time.perf_counter() - t0, 1
        sample = f"{k} periods (with future term) of {w.name} = {total_cells:.3g} cells in {total_t:.1f} s on {cores} threads"
    else:
        n = int(max(n_probe, min(S_f, S_f * 0.6 * target_seconds / est_period)))
        lo = (S_f - n) // 2
        t0 = time.perf_counter()
        _, _, total_cells = P.period(pf, v0, lo=lo, hi=lo + n, nthreads=cores)
        total_t = time.perf_counter() - t0
        sample = (f"states [{lo}, {lo + n}) of period {pf} of {w.name} (of {S_f}; successor values zero) = "
                  f"{total_cells:.3g} cells in {total_t:.1f} s on {cores} threads")
    n1 = max(1, min(S_f, int(n_probe * 3.0 / max(t_probe * cores, 1e-6))))  # about three seconds of one thread
    lo1 = (S_f - n1) // 2
    t0 = time.perf_counter()
    _, _, c1 = P.period(pf, v0, lo=lo1, hi=lo1 + n1, nthreads=1)
    t1 = max(time.perf_counter() - t0, 1e-9)
    return {"value": total_cells / total_t, "unit": "cells/s", "cores": cores, "kind": "port", "sample": sample,
            "single_thread_cells_per_s": c1 / t1}


# ---------------------------------------------------------------------------------------------------------------
# roofline
# ---------------------------------------------------------------------------------------------------------------
def kernel_source_sha(workload_name: str) -> str:
    """Digest of the kernel sources this workload runs on (tools/kernel_sha.py): a counter summary measured on another
    build says nothing about this one's instruction counts or traffic."""
    from tools.kernel_sha import kernel_source_sha as sha
    return sha(ROOT, workload_name)


def load_pmc(workload_name: str):
    """Counter summary of this workload written by tools/pmc_reduce.py (rocprofv3 --pmc passes, corrected as
    MI355X_MICROARCH.md prescribes), newest round first -- only if it was measured on THIS build of the kernels."""
    sha = kernel_source_sha(workload_name)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{workload_name}.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("source_sha") != sha:
            continue
        rec["_file"] = os.path.relpath(path, ROOT)
        return rec
    return None


def roofline_block(w, st, T, cells_rank, states_rank, dev_ms_per_sweep, per_launch_ms, per_launch_cells=None):
    """`dev_ms_per_sweep`: HIP-event time of one timed sweep on the launch stream (average over the timed steps)."""
    launches = T
    avg_launch_ms = dev_ms_per_sweep / launches
    fp64_ops = float(st.fp64_ops_executed)  # per sweep, this rank: cells x executed fp64 add/mul per cell
    pmc = load_pmc(w.name)
    alg_bytes_launch = algorithmic_bytes(cells_rank, states_rank) / launches
    alg_gbps = alg_bytes_launch / (avg_launch_ms * 1e-3) / 1e9
    hbm = None
    if pmc and pmc.get("hbm_bytes_per_launch"):
        b = float(pmc["hbm_bytes_per_launch"])
        gb = b / (avg_launch_ms * 1e-3) / 1e9
        hbm = {"bytes_per_launch": b, "GBps": gb, "peak_GBps": HBM_PEAK_GBPS, "frac": gb / HBM_PEAK_GBPS,
               "source": pmc["_file"], "kernel": pmc.get("dominant_kernel"),
               "note": "2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 correction), separate --pmc passes"}
    out = {
        "avg_launch_ms": avg_launch_ms,
        "launches_per_sweep": launches,
        "per_launch_ms_events": [round(m, 5) for m in per_launch_ms],
        "traffic": hbm["bytes_per_launch"] if hbm else None,
        "hbm": hbm,
        "algorithmic": {"bytes_per_launch": alg_bytes_launch, "GBps": alg_gbps, "frac_of_hbm_peak": alg_gbps / HBM_PEAK_GBPS,
                        "note": "SURVEY 8(d) byte MODEL (8 B per cell + 12 B per state-period), not traffic: the per-cell "
                                "read of V_{t+1} is served by LDS / L2, so this figure may exceed the HBM peak and is not "
                                "a roofline fraction"},
    }
    # on-chip data paths of the CUs: 128 B/clk/CU of LDS, 64 B/clk/CU of vector L1 (MI355X_MICROARCH.md), 256 CUs at 2.4 GHz
    lds_bytes, l1_bytes = float(getattr(st, "lds_bytes", 0.0)), float(getattr(st, "l1_bytes", 0.0))
    units = {}
    if lds_bytes > 0:
        units["lds"] = {"achieved": lds_bytes / (dev_ms_per_sweep * 1e-3) / 1e12, "peak": LDS_PEAK_BPS / 1e12, "unit": "TB/s",
                        "bytes_per_cell": lds_bytes / max(cells_rank, 1)}
    if l1_bytes > 0:
        units["vector-l1"] = {"achieved": l1_bytes / (dev_ms_per_sweep * 1e-3) / 1e12, "peak": L1_PEAK_BPS / 1e12, "unit": "TB/s",
                              "bytes_per_cell": l1_bytes / max(cells_rank, 1)}
    if fp64_ops > 0:
        lane_ops = fp64_ops / (dev_ms_per_sweep * 1e-3)
        ops_future = None
        if int(st.window_r) > 0:
            R_, S_ = int(st.window_r), int(st.window_s)
            ops_future = [round(3.0 + 1.0 / S_ + (R_ + S_ - 1.0) / (R_ * S_), 4), round(2.0 + 1.0 / S_, 4)]
        out.update({
            "bound": "fp64-valu",
            "achieved": lane_ops / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s",
            "frac": lane_ops / VALU_PEAK_LANE_OPS,
            "ops_per_cell": fp64_ops / max(cells_rank, 1),
            "ops_per_cell_future_last": ops_future,
            "register_block": {"actions": int(st.window_r), "states_per_lane": int(st.window_s)} if int(st.window_r) else None,
            "note": "executed fp64 add/mul per cell (no FMA by contract; operations that are identical in neighbouring "
                    "cells are formed once) x cells / HIP-event time, against 16 lanes/clk/SIMD x 1024 SIMDs x 2.4 GHz",
        })
    elif pmc and pmc.get("valu_insts_per_launch"):
        # (the counted kernel is the one of the periods before T: priced against ITS launches, not the sweep average, which
        # the cheaper period-T launch would flatter)
        dom = [m for m in per_launch_ms[1:] if m > 0] or [avg_launch_ms]
        dom_ms = sum(dom) / len(dom)
        priced = "HIP-event launch time"
        # A sweep whose periods run DIFFERENT kernels or very different grids (the workforce family: staff ranges of 1 to 7001
        # numbers, four kernel forms) has no "the launch" to divide the counted kernel's instructions by: the mean of its
        # launches is not that kernel's duration (priced so, round 4's staff entry read 0.96 where the unit was 0.78 busy).
        # Told apart by the counted kernel's own duration in the profile; its instructions are then priced against THAT.
        prof_us = ((pmc.get("kernels") or {}).get(pmc.get("dominant_kernel") or "", {}).get("rocprof") or {}).get("avg_us")
        if prof_us and abs(prof_us * 1e-3 - dom_ms) > 0.15 * dom_ms:
            dom_ms = float(prof_us) * 1e-3
            priced = "the counted kernel's own rocprofv3 duration (the sweep's launches run different kernels / grids)"
        lane_ops = float(pmc["valu_insts_per_launch"]) * 64.0 / (dom_ms * 1e-3)
        out.update({
            "bound": "valu-issue",
            "achieved": lane_ops / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s",
            "frac": lane_ops / VALU_PEAK_LANE_OPS,
            "dominant_launch_ms": dom_ms,
            # (the counted kernel's launches are the periods before T: priced against THEIR cells -- period T of a family may
            # offer fewer orders, e.g. none in SingleProductLeadtime's last period)
            "valu_insts_per_cell": float(pmc["valu_insts_per_launch"]) * 64.0 / max(
                (sum(c for c in per_launch_cells[1:] if c > 0) / max(1, sum(1 for c in per_launch_cells[1:] if c > 0)))
                if per_launch_cells and any(c > 0 for c in per_launch_cells[1:]) else cells_rank / launches, 1),
            "ta_busy_frac": pmc.get("ta_busy_frac"),
            "note": f"SQ_INSTS_VALU x 64 lanes per launch ({pmc['_file']}) / {priced}; every wave64 VALU "
                    "instruction holds its SIMD four cycles",
        })
    else:
        out.update({"bound": "valu-issue", "achieved": None, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s",
                    "frac": None, "note": "no instruction model and no counter summary in profiles/ for this workload"})
    if pmc:
        out["counters"] = {k: pmc.get(k) for k in ("valu_busy_frac", "lds_busy_frac", "ta_busy_frac")}
        out["counters"]["source"] = pmc["_file"]
    # the unit that binds is the one closest to its peak: a kernel whose per-cell read goes through the LDS (cash_diag_kernel)
    # or the vector L1 (cash_shift_kernel) is priced against that path, with the fp64 issue fraction kept beside it
    for u in units.values():
        u["frac"] = u["achieved"] / u["peak"]
    if units:
        out["units"] = dict(units)
        if out.get("frac") is not None:
            out["units"]["fp64-valu" if out["bound"] == "fp64-valu" else out["bound"]] = {
                "achieved": out["achieved"], "peak": out["peak"], "unit": out["unit"], "frac": out["frac"]}
        name, top = max(units.items(), key=lambda kv: kv[1]["frac"])
        if out.get("frac") is None or top["frac"] > out["frac"]:
            out.update({"bound": name, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
                        "note": f"bytes the kernel's cells move through the {name} path (library model: "
                                f"{top['bytes_per_cell']:.3g} B per cell) / HIP-event time, against "
                                + ("128" if name == "lds" else "64") + " B/clk/CU x 256 CUs x 2.4 GHz; the other units: `units`"})
    if out.get("frac") is not None and not (0.0 < out["frac"] <= 1.0):
        # a fraction above 1 is not a roofline fraction: the op model / counter summary does not describe this kernel (a
        # summary in profiles/ measured on another build is skipped by its source_sha).  Never printed: the line keeps its
        # measured throughput and says why the fraction is missing.
        msg = f"roofline fraction {out['frac']:.4f} outside (0, 1] withheld: the model / counter summary does not describe this kernel"
        print("[bench] " + msg, file=sys.stderr, flush=True)
        out.update({"frac": None, "achieved": None, "error": msg})
        out.pop("units", None)
    return out


# ---------------------------------------------------------------------------------------------------------------
# one workload on one GPU (the headline at N = 1 and every secondary entry)
# ---------------------------------------------------------------------------------------------------------------
def run_single(sia, torch, dev, name, w, steps, warmup, kernel, gate_cells, no_gate, ping_pong=False, min_warmup=2):
    """`ping_pong`: two value tables instead of one per period (store_all_values = 0; what configs[4]'s full horizon runs
    on) -- only V_1 and V_2 outlive the sweep, so the gate checks period 1."""
    if name == "multilead_kat2":
        return run_multilead(steps, no_gate)
    d = w.desc()
    d.device = dev.index
    d.kernel = 3 if name in ("separable_target", "separable_f5") else kernel
    if ping_pong:
        d.store_all_values = 0
    T = w.T

    def gate_of(eng, budget, seed=5):
        if no_gate:
            return True, {"status": "skipped (--no-gate)"}
        if getattr(w, "level_pmf", None) is not None:
            return staff_gate(eng, w, budget, seed=seed)
        return parity_gate(eng, w, budget, seed=seed, periods=[1] if ping_pong else None,
                           rel_tol=1e-9 if name == "separable_target" else None)

    # A stream of the run's own (not the legacy NULL stream, which cannot be captured): with SDPGPU_GRAPH=1 sdpgpu_solve replays
    # its sweep as ONE HIP graph from the third call on (the warm-up covers the eager and the capturing call).  Off by default:
    # measured 0.6-1.6 % slower than the eager sweep, whose wall time is within 0.4 % of its device time (`wall_over_device`).
    # The torch events below are recorded on the same stream.
    side = torch.cuda.Stream(device=dev)
    with make_engine(sia, w, d) as eng, torch.cuda.stream(side):
        eng.set_stream(side.cuda_stream)
        eng.solve(sync=True)
        ok, gate = gate_of(eng, gate_cells)
        if not ok:
            raise SystemExit(f"PARITY GATE FAILED on {w.name}: {json.dumps(gate)} -- no timing accepted")
        for _ in range(max(warmup, min_warmup)):
            eng.solve(sync=False)
        torch.cuda.synchronize(dev)
        if not no_gate and eng.stats().graph_replays > 0:
            # the timed sweeps are graph replays: their tables must be the gated (eager) sweep's, bit for bit
            import numpy as np
            ok2, gate2 = gate_of(eng, gate_cells / 4, seed=11)
            if not ok2:
                raise SystemExit(f"PARITY GATE FAILED on the graph-replayed sweep of {w.name}: {json.dumps(gate2)}")
            gate["graph_replay_gate"] = {"status": gate2["status"], "states_checked": gate2["states_checked"]}
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            eng.solve(sync=False)
            ev[k][1].record()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        dev_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
        st = eng.stats()
        replays = int(st.graph_replays)
        eng.set_profiling(True)
        eng.solve(sync=True)
        per_launch = [eng.period_ms(p) for p in range(T, 0, -1)]
        per_launch_cells = [eng.period_cells(p) for p in range(T, 0, -1)]
        eng.set_profiling(False)
        cells = int(st.cells_evaluated)
        states = int(st.states_total)
        rf = roofline_block(w, st, T, cells, states, dev_ms, per_launch, per_launch_cells)
        rf["sweep_issue"] = ("one hipGraphLaunch per sweep (captured inside sdpgpu_solve)" if replays >= steps
                             else "eager: one launch per period")
        rf["wall_over_device"] = (elapsed / steps * 1e3) / dev_ms if dev_ms > 0 else None
        n_states = max(eng.num_states(p) for p in range(1, T + 1))  # (unclamped families grow with the period: the widest)
    out = {"workload": w.name, "family": FAMILY_NAME.get(name, name), "value": cells * steps / elapsed, "unit": "cells/s",
           "ms_per_step": elapsed / steps * 1e3, "steps": steps, "periods": T, "states": n_states,
           "cells_per_step": cells, "parity_gate": gate, "roofline": rf,
           "kernel": {0: "auto", 1: "gather", 2: "specialised (window / shift / row)", 3: "separable"}[int(st.kernel_used)]}
    if ping_pong:
        out["tables"] = "two ping-pong value tables (store_all_values = 0), as the full 100-period horizon runs"
    if getattr(w, "custom_source", None):
        out["kernel"] = ("user cost functions compiled with hipRTC, tabulated per period; the library's F1 window kernel reads the tables"
                         if int(st.kernel_used) == 2 else "user lambdas compiled with hipRTC around the generic period loop")
    if name == "separable_target":
        out["value_is"] = ("brute-force-equivalent cells/s: the cells of the (state x action x demand) grid / time -- the mode "
                           "evaluates O((S + A) D + S A) terms instead; compare `ms_per_step` with the headline's")
    if name == "separable_f5":
        out["value_is"] = ("brute-force-equivalent cells/s: the dense grid's cells / time -- the mode evaluates one row per level "
                           "x + preQ (the rows of a level hold identical tables) cell by cell and copies it to the others: EXACT, "
                           "values and policy bit-identical; compare `ms_per_step` with the f5_spl entry's")
    return out


# ---------------------------------------------------------------------------------------------------------------
# bare `python bench.py --gpus N`, N > 1: start the N ranks ourselves
# ---------------------------------------------------------------------------------------------------------------
def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the PARENT of a
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` CHILD process (one rank
    per GPU underneath it) -- a child, never an exec, started before anything here has imported torch or touched HIP --
    and relays rank 0's ONE JSON line on its own stdout (the ranks' stderr is passed through).  The reference's contract
    is one call that solves everything (Recursion.java:89); so is this.

    A non-zero exit of the ranks, no bench line, or an overrun of --launch-timeout ends with ONE JSON error record on
    stdout (phase = the last phase a rank announced, the ranks' own watchdog record if one fired, the tail of stderr)
    and a non-zero exit code; on an overrun the process group this parent started (and only that) is ended first."""
    import collections
    import signal
    import socket
    import subprocess
    import threading

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SDP_BENCH_PHASE_TRACE"] = "1"  # the ranks announce every phase on stderr: the record of a failure names the last one
    t0 = time.monotonic()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT,
                            start_new_session=True)  # its own process group: what an overrun ends, by id
    out_lines, err_tail, fatal = [], collections.deque(maxlen=40), []
    last_phase = {}

    def pump_out():
        for line in proc.stdout:
            out_lines.append(line.rstrip("\n"))

    def pump_err():
        for line in proc.stderr:
            sys.stderr.write(line)
            sys.stderr.flush()
            text = line.rstrip("\n")
            if text.startswith("[bench phase] "):  # "[bench phase] rank R: NAME"
                try:
                    head, name = text[len("[bench phase] "):].split(": ", 1)
                    last_phase[int(head.split()[1])] = name
                except (ValueError, IndexError):
                    pass
            elif text.startswith("[bench fatal] "):  # a rank's own reason for ending (see __main__)
                fatal.append(text[len("[bench fatal] "):])
            elif text.strip():
                err_tail.append(text)

    pumps = [threading.Thread(target=pump_out, daemon=True), threading.Thread(target=pump_err, daemon=True)]
    for t in pumps:
        t.start()
    timed_out = False
    try:
        rc = proc.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
            try:
                os.killpg(proc.pid, sig)  # exactly the group started above (start_new_session: pgid == proc.pid)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        rc = proc.returncode if proc.returncode is not None else -9
    for t in pumps:
        t.join(timeout=5.0)

    records, dec = [], json.JSONDecoder()
    for line in out_lines:
        if line.startswith("{"):
            try:  # (one record a line; two ranks' writes that met in the pipe are taken apart)
                pos, found = 0, []
                while line.startswith("{", pos):
                    rec, pos = dec.raw_decode(line, pos)
                    found.append(rec)
                if pos == len(line.rstrip()):
                    records.extend(found)
                    continue
            except ValueError:
                pass
        if line.strip():
            print(line, file=sys.stderr, flush=True)  # anything else a rank printed is not the bench line
    bench = [r for r in records if r.get("metric") and r.get("value") is not None and "error" not in r]
    if rc == 0 and not timed_out and len(bench) == 1:
        print(json.dumps(bench[0]), flush=True)
        return 0
    fired = [r for r in records if "error" in r]  # the ranks' own watchdog records (phase, rank, times)
    phase = (fired[0].get("phase") if fired else None) or (last_phase.get(0) or next(iter(last_phase.values()), None)) or "launch"
    rec = {"error": ("launch timeout exceeded" if timed_out else
                     "ranks exited with a non-zero code" if rc != 0 else
                     f"{len(bench)} bench lines on the ranks' stdout (expected exactly one)"),
           "phase": phase, "returncode": rc, "elapsed_s": round(time.monotonic() - t0, 2),
           "launch_timeout_s": args.launch_timeout, "n_gpus": args.gpus, "workload": args.workload,
           "metric": "(state,action,demand) cell evals/sec", "value": None,
           "last_phase_per_rank": {str(k): v for k, v in sorted(last_phase.items())},
           "rank_records": fired[:8], "fatal": fatal[:8], "stderr_tail": list(err_tail)[-12:]}
    print(json.dumps(rec), flush=True)
    return rc if rc not in (0, None) else 1


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the N ranks as a child process (before torch or HIP are touched) and relay their line
        sys.exit(self_launch(args, sys.argv[1:]))
    if os.environ.get("SDP_BENCH_TEST_STALL_AT_START"):  # (tests: a rank that never gets as far as the process group)
        time.sleep(float(os.environ["SDP_BENCH_TEST_STALL_AT_START"]))
    # multi-process GPU work on this image needs dmabuf IPC (RCCL's buffer exchange between the ranks' processes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.exchange == "host" or local_rank >= torch.cuda.device_count():
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: several ranks share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded_path = world > 1 or args.rehearse_sharded
    if args.rehearse_sharded and world == 1:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=0, world_size=1)
    import stochastic_inventory_amd as sia
    from stochastic_inventory_amd.watchdog import PhaseWatchdog

    # N > 1: every phase of the run has a deadline watched by a thread of this rank; a rank that overruns prints a one-line
    # JSON error record (phase, rank, times) and exits non-zero -- a stalled peer ends the run with a diagnosis instead of
    # leaving everybody inside a collective until the launcher's timeout.  Nothing is relaunched or re-executed.
    wd = PhaseWatchdog(rank, world, context={"metric": "(state,action,demand) cell evals/sec", "n_gpus": world,
                                               "workload": args.workload, "exchange_requested": args.exchange}) if sharded_path else None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: do not depend on the hostname resolving
        # RCCL's bootstrap (the rendezvous behind ncclCommInitRank; the data itself travels over xGMI) on the loopback interface
        # as well -- a GPU box without network may have no other -- and its warnings on stderr, so that a failed communicator
        # says why
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        # native exchange: the data path is RCCL inside libsdpgpu.so; the process group carries the communicator id,
        # the barriers and the timing reductions, for which gloo is enough
        with wd.phase("process group", 180):
            dist.init_process_group("nccl", device_id=dev) if args.exchange == "torch" else dist.init_process_group("gloo")

    if args.workload in FAMILY_WORKLOADS and sharded_path:
        raise SystemExit(f"--workload {args.workload} is a single-GPU entry (the N > 1 path shards the grid families)")
    w = None if args.workload == "multilead_kat2" else make_workload(args.workload, world, args.states, args.periods, args.weak)
    T = w.T if w is not None else 3

    if not sharded_path:
        head = run_single(sia, torch, dev, args.workload, w, args.steps, args.warmup, args.kernel, args.gate_cells,
                          args.no_gate, ping_pong=(args.workload == "cfg5" and not args.states))
        f = getattr(w, "functor", None)
        out = {
            "metric": "(state,action,demand) cell evals/sec",
            "value": head["value"], "unit": "cells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": head["workload"], "family": head["family"], "states": head["states"],
                       "actions": (int(getattr(f, "maxOrderQuantity", getattr(f, "maxHireNum", 0))) + 1) if f is not None else None,
                       "demands": len(w.pmf[0]) if getattr(w, "pmf", None) is not None else None, "periods": head["periods"],
                       "cells_per_step": head["cells_per_step"], "parallelism": "single GPU", "exchange": "single rank",
                       "kernel": head["kernel"]},
            "parity_gate": head["parity_gate"],
            "roofline": head["roofline"],
            "build": build_identity(),
        }
        flops = REFERENCE_FLOPS_PER_CELL.get(args.workload, 14)
        if "algorithmic" in head["roofline"]:
            out["side_by_side"] = {
                "cells_per_s": out["value"],
                "algorithmic_GBps": head["roofline"]["algorithmic"]["GBps"],
                "hbm_measured_GBps": head["roofline"]["hbm"]["GBps"] if head["roofline"]["hbm"] else None,
                "fp64_TFLOPs_at_reference_op_count": out["value"] * flops / 1e12,
                "reference_fp64_ops_per_cell": flops,
                "compulsory_bytes_per_launch": 20.0 * head["states"],
                "note": "the reference's formulas spend 14 (F1/F2) / 25 (F3) fp64 operations per cell; the kernels execute fewer "
                        "(identical operations are formed once): roofline.ops_per_cell",
            }
        if args.workload == "target" and not args.no_secondary and not args.weak and not args.states and not args.periods:
            sec = []

            def secondary(name, ws, st_, wu_, gate_cells, **kw):
                """One gated secondary entry.  A failure of a SECONDARY workload (its gate included) is recorded in its place --
                no value, the reason -- and does not take the headline, whose own gate has passed, down with it."""
                only = os.environ.get("SDP_BENCH_ONLY_SECONDARY")  # (tests: a shorter list)
                if only and name not in only.split(","):
                    return None
                try:
                    if os.environ.get("SDP_BENCH_TEST_FAIL_SECONDARY") == name:  # (tests: this entry fails)
                        raise SystemExit(f"PARITY GATE FAILED on {name}: injected (SDP_BENCH_TEST_FAIL_SECONDARY)")
                    sec.append(run_single(sia, torch, dev, name, ws, st_, wu_, 0, gate_cells, args.no_gate, **kw))
                    return sec[-1]
                except (SystemExit, Exception) as exc:  # noqa: B014 (SystemExit is how a failed gate ends a run)
                    msg = str(exc.code) if isinstance(exc, SystemExit) else f"{type(exc).__name__}: {exc}"
                    print(f"[bench] secondary entry {name} failed: {msg}", file=sys.stderr, flush=True)
                    sec.append({"workload": name, "family": FAMILY_NAME.get(name, name), "value": None, "error": msg[:600],
                                "parity_gate": {"status": "FAILED" if "PARITY GATE FAILED" in msg else "not reached"}})
                    return None

            # (configs[1]: a sweep is 52 launches of 30 us -- 1.5 ms; 100 of them, so that one hiccup of the host thread that issues
            # them does not move the entry by a fifth, as it did in one of three runs with 20)
            for name, st_, wu_ in (("cfg2", 100, 10), ("cfg3", 3, 1), ("cfg3t", 3, 1), ("cfg4", 3, 1), ("cfg4p", 2, 1)):
                secondary(name, make_workload(name, 1), st_, wu_, args.gate_cells / 3)
            # configs[4] at its full width (1e8 states, the largest single-GPU configuration; three periods of its hundred, on
            # the two ping-pong tables the full horizon runs on: tools/cfg5_full_horizon.py), then the SURVEY 8(f)-3 families at
            # the sizes of the reference's slowest drivers and the 8(f)-4 mode -- each gated like the rest
            secondary("cfg5", make_workload("cfg5", 1, periods=3), 1, 1, args.gate_cells / 8, ping_pong=True, min_warmup=1)
            f5 = None
            for name, st_, wu_ in (("f5_spl", 3, 1), ("staff", 20, 2), ("custom_clsp", 10, 2), ("custom_clsp_level", 100, 10),
                                   ("multilead_kat2", 3, 0)):
                ws = None if name == "multilead_kat2" else make_workload(name, 1)
                e = secondary(name, ws, st_, wu_, args.gate_cells / 6)
                if name == "f5_spl":
                    f5 = e
            sep = secondary("separable_target", make_workload("separable_target", 1), 5, 1, args.gate_cells / 6)
            if sep:
                sep["ms_per_sweep"] = sep["ms_per_step"]
                sep["speedup_over_brute_force"] = head["ms_per_step"] / sep["ms_per_step"]
                sep["brute_force_ms_per_sweep"] = head["ms_per_step"]
            sep5 = secondary("separable_f5", make_workload("separable_f5", 1), 5, 1, args.gate_cells / 6)
            if sep5 and f5:
                sep5["ms_per_sweep"] = sep5["ms_per_step"]
                sep5["speedup_over_brute_force"] = f5["ms_per_step"] / sep5["ms_per_step"]
                sep5["brute_force_ms_per_sweep"] = f5["ms_per_step"]
            out["secondary"] = sec
        if not args.no_cpu_baseline and args.workload not in FAMILY_WORKLOADS:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
        print(json.dumps(out), flush=True)
        return

    # ------------------------------------------------------------------------------------------------------
    # N > 1
    # ------------------------------------------------------------------------------------------------------
    from stochastic_inventory_amd.sharded import GpuSlabBackend, ShardedSolver, init_native_comm

    ctrl = dev if args.exchange == "torch" else torch.device("cpu")  # where the default group's tensors live

    def barrier():
        dist.barrier()
        torch.cuda.synchronize(dev)

    def all_agree(ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctrl)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def max_over_ranks(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=ctrl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    desc = w.desc()
    desc.rank, desc.world_size = rank, world
    desc.kernel = args.kernel
    backend = GpuSlabBackend(desc, w.pmf, w.overhead(), device=dev)
    eng = backend.engine
    exchange = args.exchange
    note = ""
    if os.environ.get("SDP_TEST_BREAK_RCCL_ON_RANK") == str(rank):  # (tests: this rank's collective library does not load)
        os.environ["SDPGPU_RCCL_LIB"] = "/nonexistent/librccl.so"
    t_comm0 = time.perf_counter()
    with wd.phase("communicator init", 180):
        native_ok = exchange != "native" or init_native_comm(eng)  # (the ranks agree inside: all or none)
    comm_init_s = time.perf_counter() - t_comm0
    if not native_ok:
        exchange = "torch"
        err = init_native_comm.last_error
        note = f" (native communicator failed{': ' + err if err else ' on another rank'}; fell back to torch.distributed nccl)"
        if rank == 0 or err:  # (rank 0 announces the fallback; a rank whose own preparation failed says why)
            print(f"[bench] rank {rank}: native RCCL communicator unavailable{': ' + err if err else ' (another rank could not prepare)'}; "
                  "using torch.distributed", file=sys.stderr, flush=True)
    group = dist.new_group(backend="nccl") if (exchange == "torch" and args.exchange != "torch") else None
    solver = ShardedSolver(backend, group=group, stage_through_host=(exchange == "host"))
    solver.force_split = args.split

    def make_runner(name):
        if exchange == "native":
            return lambda: solver.solve_native(overlap=(name == "overlap"), sync=False)
        if name == "overlap":
            return lambda: solver.solve(overlap=True)
        if name == "blocking":
            return lambda: solver.solve(overlap=False)
        k = int(name[len("blocked"):])
        return lambda: solver.solve_blocked(k)

    sched = "blocking" if args.no_overlap else args.schedule
    if args.split:
        sched = "overlap"
    blocked_ok = exchange != "native" and not args.split and solver.prepare_blocked(8)
    if sched.startswith("blocked") and not blocked_ok:
        raise SystemExit("blockedK needs --exchange torch|host and a workload with a bounded dependency footprint")

    # PARITY GATE FIRST (every rank on its own slab; all must pass), on the plainest schedule -- a blocking all-gather per
    # period -- and before any schedule is timed: nothing is calibrated, warmed up or timed on tables that have not been
    # checked, and the first collectives of the run are issued in a phase with its own deadline.
    first_runner = make_runner("blocking" if sched == "auto" else sched)
    t_first0 = time.perf_counter()
    with wd.phase("first sweep", 300):
        first_runner()
        torch.cuda.synchronize(dev)
        barrier()
    first_sweep_s = time.perf_counter() - t_first0
    gate = {"status": "skipped (--no-gate)"}
    if not args.no_gate:
        with wd.phase("parity gate", 900):
            ok, gate = parity_gate(eng, w, args.gate_cells / world, seed=5 + rank)
            if not all_agree(ok):
                raise SystemExit(f"PARITY GATE FAILED (rank {rank}: {json.dumps(gate)}) -- no timing accepted")
        gate["ranks"] = world
        gate["schedule"] = "blocking" if sched == "auto" else sched

    schedule_note = sched
    if sched == "auto":
        candidates = ["blocking", "overlap"]
        if blocked_ok and (exchange == "torch" or os.environ.get("SDP_BENCH_CALIBRATE")):
            candidates += ["blocked2", "blocked4", "blocked8"]
        if exchange == "host" and not os.environ.get("SDP_BENCH_CALIBRATE"):
            candidates = ["overlap"]
        timing = {}
        with wd.phase("schedule calibration", 120 + 40 * first_sweep_s * len(candidates)):
            for name in candidates:
                run_c = make_runner(name)
                took, ok = float("inf"), True
                try:
                    run_c()
                    torch.cuda.synchronize(dev)
                except Exception as exc:
                    ok = False
                    print(f"[bench] schedule {name} failed on rank {rank}: {exc}", file=sys.stderr, flush=True)
                # agree BEFORE any further collective: a rank that failed must not leave the others inside one
                if not all_agree(ok):
                    if name in ("blocking",):
                        raise SystemExit("the blocking schedule failed: nothing to fall back to")
                    continue
                barrier()
                t0c = time.perf_counter()
                run_c()
                run_c()
                barrier()
                took = (time.perf_counter() - t0c) / 2
                timing[name] = max_over_ranks(took)
        sched = min(timing, key=timing.get)
        schedule_note = sched + " (calibrated, ms per sweep: " + ", ".join(f"{k} {v * 1e3:.3f}" for k, v in timing.items()) + ")"
    run_sweep = make_runner(sched)
    if sched != gate.get("schedule", sched):
        # the timed schedule is not the one the gate ran on: its tables must be the gated ones, bit for bit (V_1 on the slab,
        # V_2 whole, this rank's policy of period 1)
        import numpy as np
        with wd.phase("schedule cross-check", 300):
            ref = (eng.values(1).copy(), eng.values(2 if T > 1 else 1).copy(), eng.policy(1).copy())
            run_sweep()
            torch.cuda.synchronize(dev)
            _, lo1, hi1 = eng.slab(1)
            same = (np.array_equal(eng.values(1)[lo1:hi1], ref[0][lo1:hi1]) and np.array_equal(eng.policy(1), ref[2])
                    and (T == 1 or np.array_equal(eng.values(2), ref[1])))
            if not all_agree(bool(same)):
                raise SystemExit(f"schedule {sched} does not reproduce the gated tables (rank {rank}) -- no timing accepted")
        gate["timed_schedule_matches_gated_tables"] = True

    with wd.phase("warm-up", 120 + 40 * first_sweep_s * max(1, args.warmup)):
        for _ in range(args.warmup):
            run_sweep()
        barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    with wd.phase("timed loop", 120 + 40 * first_sweep_s * max(1, args.steps)):
        t0 = time.perf_counter()
        for k in range(args.steps):
            ev[k][0].record()
            run_sweep()
            ev[k][1].record()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)

    st = eng.stats()
    cells_step_all = int(st.cells_all_ranks)
    cells_step_rank = int(st.cells_evaluated)
    states_step_rank = sum(backend.slab(p)[2] - backend.slab(p)[1] for p in range(1, T + 1))
    dev_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    with wd.phase("per-launch events sweep", 120 + 40 * first_sweep_s):
        eng.set_profiling(True)
        run_sweep()
        torch.cuda.synchronize(dev)
        per_launch = [eng.period_ms(p) for p in range(T, 0, -1)]
        eng.set_profiling(False)
    compute_ms = sum(m for m in per_launch if m > 0)
    # load balance: cells per rank (F3's action count depends on the cash balance, CashConstraint.java:96-99)
    cells_t = torch.zeros(world, dtype=torch.float64, device=ctrl)
    cells_t[rank] = float(cells_step_rank)
    comp_t = torch.zeros(world, dtype=torch.float64, device=ctrl)
    comp_t[rank] = compute_ms
    sweep_t = torch.zeros(world, dtype=torch.float64, device=ctrl)
    sweep_t[rank] = dev_ms
    with wd.phase("reductions", 120):
        dist.all_reduce(cells_t)
        dist.all_reduce(comp_t)
        dist.all_reduce(sweep_t)

    check = None
    if args.check:
      with wd.phase("check vs single rank", 900):
        import numpy as np
        pol_local = eng.policy(1)
        if exchange == "native":
            eng.solve_sharded(sync=True, gather_first=True)
        else:
            solver.exchange(1)
            torch.cuda.synchronize(dev)
            eng.finalize()
        v1 = eng.values(1)
        v2 = eng.values(2 if T > 1 else 1)
        gathered = [None] * world
        dist.all_gather_object(gathered, pol_local)
        pol_all = np.concatenate(gathered)
        if rank == 0:
            d1 = w.desc()
            d1.device = dev.index
            with sia.SdpEngine(d1, w.pmf, w.overhead()) as ref:
                ref.solve()
                check = bool(np.array_equal(ref.values(1), v1) and np.array_equal(ref.policy(1), pol_all)
                             and np.array_equal(ref.values(2 if T > 1 else 1), v2))
    if rank == 0:
        rf = roofline_block(w, st, T, cells_step_rank, states_step_rank, dev_ms, per_launch)
        rf["note_multi"] = "rank 0's slab; the sweep time includes the per-period all-gathers"
        out = {
            "metric": "(state,action,demand) cell evals/sec",
            "value": cells_step_all * args.steps / elapsed, "unit": "cells/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": w.name, "family": FAMILY_NAME.get(args.workload, args.workload),
                "states": eng.num_states(1), "actions": int(w.functor.maxOrderQuantity) + 1, "demands": len(w.pmf[0]),
                "periods": T, "cells_per_step": cells_step_all,
                "parallelism": f"state axis cut into {world} slabs, one all-gather of V_t per period",
                "exchange": {"native": "RCCL all-gather issued by libsdpgpu.so (sdpgpu_solve_sharded)",
                             "torch": "torch.distributed nccl all_gather_into_tensor (sharded.py)",
                             "host": "gloo, host-staged (rehearsal)"}[exchange] + note,
                "schedule": schedule_note,
                "cells_per_rank": [int(c) for c in cells_t.tolist()],
                "kernel_ms_per_rank": [round(c, 4) for c in comp_t.tolist()],
                # device time of a sweep on the rank's launch stream less its period kernels: what the all-gathers (and the
                # waits for slower peers inside them) cost that rank
                "sweep_ms_per_rank": [round(c, 4) for c in sweep_t.tolist()],
                "exchange_ms_per_rank": [round(max(0.0, a - b), 4) for a, b in zip(sweep_t.tolist(), comp_t.tolist())],
                "communicator_init_s": round(comm_init_s, 3),
                "first_sweep_s": round(first_sweep_s, 3),
                "phases_s": {n: round(sec, 3) for n, sec in wd.history},
                "exchange_bytes_per_rank_per_period": 8 * (backend.slab(1)[0] // world),
                "kernel": {0: "auto", 1: "gather", 2: "specialised (window / shift / row)", 3: "separable"}[int(st.kernel_used)],
            },
            "parity_gate": gate,
            "roofline": rf,
            "build": build_identity(),
        }
        if check is not None:
            out["check_vs_single_rank"] = check
        # (cpu_baseline is the N = 1 line's: the oracle timed on rank 0 while seven GPUs wait would only lengthen the scaling run)
        print(json.dumps(out), flush=True)
    with wd.phase("teardown", 120):
        dist.barrier()
        dist.destroy_process_group()
        backend.close()
    wd.close()


if __name__ == "__main__":
    try:
        main()
    except SystemExit as exc:
        if isinstance(exc.code, str):  # a refusal with a reason: tagged, so that a self-launching parent can quote it in its record
            print(f"[bench fatal] rank {os.environ.get('RANK', '0')}: {exc.code}", file=sys.stderr, flush=True)
            sys.exit(1)
        raise
    except Exception as exc:
        print(f"[bench fatal] rank {os.environ.get('RANK', '0')}: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
        raise

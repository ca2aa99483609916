#!/usr/bin/env python3
"""bench.py -- the hot path (backward Bellman sweep) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full backward sweep t = T..1 of the workload with every input (PMF tiles,
descriptor) already resident in HBM.  Default workload at N = 1: BASELINE.json configs[1], the
capacitated lot-sizing grid (1e4 states x 200 actions x 100 demands, 52 periods = 1.04e10
cells).  For N > 1 the state axis is sharded (weak scaling: each rank keeps a 1e4-state slab, the
grid grows to N * 1e4 states) with one RCCL all-gather of V_t per period.

Rank 0 prints ONE JSON line: the driver's contract plus `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", help="cfg2 (default) | cfg1 | cfg3 | cfg4 | cfg4p | cfg5")
    ap.add_argument("--states", type=int, default=0, help="override the per-GPU state count (cfg2/cfg5)")
    ap.add_argument("--periods", type=int, default=0, help="override the horizon")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 gather, 2 window, 3 separable (opt-in, F1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target-grid", action="store_true", help="skip the informational 1e6 x 500 x 200 probe")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL, default) | gloo (debug: N ranks on one GPU, host-staged)")
    ap.add_argument("--split", action="store_true", help="rehearsal: force the interior/boundary split of every period")
    ap.add_argument("--no-overlap", action="store_true", help="blocking all-gather between periods (no compute overlap)")
    ap.add_argument("--schedule", default="auto",
                    help="N > 1 exchange schedule: auto (time the candidates before the warm-up, keep the fastest) | overlap | "
                         "blocking | blockedK (K periods per exchange on widened slabs, e.g. blocked4)")
    ap.add_argument("--check", action="store_true", help="compare the sharded result with a single-rank sweep (rank 0)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    return ap.parse_args()


def make_workload(args, world):
    from stochastic_inventory_amd import workloads
    kw = {}
    if args.periods:
        kw["T"] = args.periods
    if args.workload == "cfg2":
        per_gpu = args.states or 10000
        return workloads.cfg2_clsp(S=per_gpu * world, **kw)
    if args.workload == "cfg5":
        per_gpu = args.states or 1000000
        return workloads.cfg5_scaled(S=per_gpu * world, **kw)
    if args.workload == "cfg4":  # weak scaling along the inventory axis of every preQ row
        return workloads.cfg4_leadtime(NX=(args.states or 1000) * world, **kw)
    if args.workload == "cfg4p":  # the 3-D pipeline state (x, q1, q2): weak scaling along x
        return workloads.cfg4_pipeline(NX=(args.states or 250) * world, **kw)
    if args.workload == "cfg3":  # weak scaling along the inventory axis (cash rows stay whole)
        return workloads.cfg3_cash(NX=(args.states or 200) * world, **kw)
    if world > 1:
        raise SystemExit(f"workload {args.workload} has no sharded bench definition")
    return workloads.by_name(args.workload, **kw)


def algorithmic_bytes(cells: int, states_periods: int) -> float:
    """8 B per cell (one fp64 gather of V_{t+1}) + 12 B per state-period
    (8 B V_t write + 4 B policy write).  The PMF tile (16*D B per period) is below 0.01 %."""
    return 8.0 * cells + 12.0 * states_periods


def cpu_baseline(w, target_seconds: float):
    """The CPU oracle ('port' of the Java recursion, oracle/sdpref.c) timed on a BOUNDED sample of the same workload on
    all host cores, sized from a probe to about `target_seconds`: whole periods from the end of the horizon when a
    period fits the budget (the bench default), else a run of states of the last-but-one period (the big grids; the
    successor values it reads are then zeros -- the arithmetic per cell is the same).  Reported baseline, not the
    target."""
    from oracle import sdpref
    import numpy as np
    cores = min(os.cpu_count() or 1, 16)
    P = sdpref.Problem(w.desc(), w.pmf, w.overhead())
    T = w.T
    pf = max(T - 1, 1)  # a period with a future term (the only period of a single-period horizon has none)
    S_f = int(P.S[pf - 1])
    v0 = np.zeros(int(P.S[pf])) if pf < T else None
    # probe: a thin slice of that period's states from the middle of the grid
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
    # a slice single-threaded for the like-for-like figure next to the single-threaded Java loop
    n1 = max(1, min(S_f, int(n_probe * 3.0 / max(t_probe * cores, 1e-6))))  # about three seconds of one thread
    lo1 = (S_f - n1) // 2
    t0 = time.perf_counter()
    _, _, c1 = P.period(pf, v0, lo=lo1, hi=lo1 + n1, nthreads=1)
    t1 = max(time.perf_counter() - t0, 1e-9)
    return {
        "value": total_cells / total_t,
        "unit": "cells/s",
        "cores": cores,
        "kind": "port",
        "sample": sample,
        "single_thread_cells_per_s": c1 / t1,
    }


def target_grid_probe(sia, dev):
    """Secondary, informational: the grid BASELINE.json's target sentence names (1e6 states x 500 actions x
    200 demands) for 3 periods on this one GPU, same kernels, same accounting.  Not the bench metric."""
    import torch
    from stochastic_inventory_amd import workloads
    w = workloads.cfg5_scaled(S=1000000, T=3)
    d = w.desc()
    d.device = dev.index
    with sia.SdpEngine(d, w.pmf) as eng:
        eng.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        eng.solve(sync=True)  # warm-up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = 3
        e0.record()
        for _ in range(steps):
            eng.solve(sync=False)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / steps
        cells = int(eng.stats().cells_evaluated)
    gbps = algorithmic_bytes(cells, 3 * 1000000) / (ms * 1e-3) / 1e9
    return {"workload": w.name, "value": cells / (ms * 1e-3), "unit": "cells/s", "ms_per_step": ms,
            "algorithmic_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: several ranks share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import stochastic_inventory_amd as sia
    from stochastic_inventory_amd.sharded import GpuSlabBackend, ShardedSolver

    w = make_workload(args, world)
    desc = w.desc()
    desc.rank, desc.world_size = rank, world
    desc.kernel = args.kernel
    backend = GpuSlabBackend(desc, w.pmf, w.overhead(), device=dev)
    solver = ShardedSolver(backend, stage_through_host=(args.backend == "gloo"))
    solver.force_split = args.split
    eng = backend.engine
    T = w.T

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- exchange schedule (N > 1) ------------------------------------------------------------------------
    # overlap:  all-gather of V_{t+1} beside the interior tiles of period t, boundary tiles after it
    # blocking: kernel, all-gather, kernel, ...
    # blockedK: K periods per exchange on slabs widened by the dependency footprint (redundant work, a K-th of the waits)
    # Which one wins depends on how this node's all-gather latency compares with a period's compute (30 us at the
    # configs[1] slab), so `auto` times every candidate (untimed, before the warm-up) and keeps the fastest; every
    # rank takes the same decision (max over ranks of each timing).
    def make_runner(name):
        if name == "overlap":
            return lambda: solver.solve(overlap=True)
        if name == "blocking":
            return lambda: solver.solve(overlap=False)
        k = int(name[len("blocked"):])
        return lambda: solver.solve_blocked(k)

    sched_name = args.schedule
    if args.no_overlap:
        sched_name = "blocking"
    if world == 1 or args.split:
        sched_name = "overlap" if not args.no_overlap else "blocking"
    blocked_ok = world > 1 and not args.split and solver.prepare_blocked(8)  # sizes the scratch rows for K <= 8
    if sched_name.startswith("blocked") and not blocked_ok:
        raise SystemExit("this workload has no bounded dependency footprint: no blocked schedule")
    schedule = "single rank"
    if world > 1 and sched_name == "auto":
        candidates = ["overlap", "blocking"] + (["blocked2", "blocked4", "blocked8"] if blocked_ok else [])
        if not (args.backend == "nccl" or os.environ.get("SDP_BENCH_CALIBRATE")):
            candidates = ["overlap"]
        timing = {}
        for name in candidates:
            run_c = make_runner(name)
            try:
                run_c()
                barrier()
                t0c = time.perf_counter()
                run_c()
                run_c()
                barrier()
                took = time.perf_counter() - t0c
            except Exception as exc:  # a schedule this node's stack cannot run is not a reason to lose the bench
                if name in ("overlap", "blocking"):
                    raise
                print(f"[bench] schedule {name} failed on rank {rank}: {exc}", file=sys.stderr, flush=True)
                took = float("inf")
            tc = torch.tensor([took], dtype=torch.float64, device=dev)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            timing[name] = float(tc.item()) / 2
        sched_name = min(timing, key=timing.get)
        schedule = sched_name + " (calibrated, ms per sweep: " + ", ".join(f"{k} {v * 1e3:.3f}" for k, v in timing.items()) + ")"
    elif world > 1:
        schedule = sched_name
    run_sweep = make_runner(sched_name)
    for _ in range(args.warmup):
        run_sweep()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        run_sweep()
        ev[k][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())

    st = eng.stats()
    cells_step_all = int(st.cells_all_ranks)    # whole grid, all ranks, one sweep
    cells_step_rank = int(st.cells_evaluated)   # this rank's slab
    states_step_rank = sum(backend.slab(p)[2] - backend.slab(p)[1] for p in range(1, T + 1))
    dev_ms = sum(a.elapsed_time(b) for a, b in ev)  # HIP events on the launch stream, this rank
    launches = args.steps * T
    avg_launch_ms = dev_ms / launches
    bytes_per_launch = algorithmic_bytes(cells_step_rank, states_step_rank) / T
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9

    # per-kernel event times in a separate, un-timed pass (cross-check for the rocprofv3 summary)
    eng.set_profiling(True)
    run_sweep()
    torch.cuda.synchronize(dev)
    per_kernel = [eng.period_ms(p) for p in range(1, T + 1)]
    eng.set_profiling(False)
    inner = [m for m in per_kernel[:-1]] or per_kernel
    kernel_ms_avg = sum(per_kernel) / len(per_kernel)

    check = None
    if args.check:
        import numpy as np
        v1 = eng.values(1)  # every rank holds the whole V_1 after finalize? V_1 is not exchanged: gather it
        pol_local = eng.policy(1)
        if world > 1:
            solver.exchange(1)
            torch.cuda.synchronize(dev)
            eng.finalize()
            v1 = eng.values(1)
            gathered = [None] * world
            dist.all_gather_object(gathered, pol_local)
            pol_all = np.concatenate(gathered)
        else:
            pol_all = pol_local
        if rank == 0:
            d1 = w.desc()
            d1.device = dev.index
            with sia.SdpEngine(d1, w.pmf, w.overhead()) as ref:
                ref.solve()
                check = bool(np.array_equal(ref.values(1), v1) and np.array_equal(ref.policy(1), pol_all)
                             and np.array_equal(ref.values(2 if T > 1 else 1), eng.values(2 if T > 1 else 1)))
    if rank == 0:
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                rec = json.load(open(prof))
                if rec.get("workload") == w.name and rec.get("kernel_used") == st.kernel_used:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "(state,action,demand) cell evals/sec",
            "value": cells_step_all * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": w.name,
                "family": "F1 backorder (capacitated.CLSP.f)" if args.workload in ("cfg2", "cfg5") else args.workload,
                "states": eng.num_states(1), "actions": int(w.functor.maxOrderQuantity) + 1,
                "demands": len(w.pmf[0]), "periods": T,
                "cells_per_step": cells_step_all,
                "parallelism": f"state-sharded x{world}, all-gather V_t per period" if world > 1 else "single GPU",
                "exchange": schedule,
                "kernel": {0: "auto", 1: "gather", 2: "window", 3: "separable (opt-in, not the graded path)"}[int(st.kernel_used)],
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_launch_ms,
                "kernel_ms_avg_events": kernel_ms_avg,
                "launches_timed": launches,
                "note": "algorithmic bytes = 8 B/cell + 12 B/state-period (SURVEY 8d); V_{t+1} is cache-resident, "
                        "so HBM traffic is far below this figure and the true limiter is fp64 VALU issue",
            },
        }
        if int(st.kernel_used) == 2 and args.workload in ("cfg2", "cfg5"):
            # the bound that actually limits the window kernel: fp64 add/mul issue (no FMA by contract);
            # peak = 16 lanes/clk/SIMD x 1024 SIMDs x 2.4 GHz (tools/valu_probe.hip measures 3.8e13 of it at the
            # clock the chip holds).
            # Operations the kernel executes per cell with R actions x S adjacent states per lane (sdp_window.hpp):
            # c0 + M once per (action, m): 1/S; p * imm: 1; p * V once per window entry: (R + S - 1)/(R S); two
            # accumulations: 2.  The last period has no future term.  Every one is an operation of the reference.
            R_, S_ = max(int(st.window_r), 1), max(int(st.window_s), 1)
            ops_future = 3.0 + 1.0 / S_ + (R_ + S_ - 1.0) / (R_ * S_)
            ops_last = 2.0 + 1.0 / S_
            ops_per_sweep = cells_step_rank * (ops_future * (T - 1) + ops_last) / T
            lane_ops = ops_per_sweep * args.steps / (dev_ms * 1e-3)
            out["valu_roofline"] = {"bound": "fp64 add/mul issue", "achieved": lane_ops / 1e12, "peak": 39.3,
                                    "unit": "T lane-op/s", "frac": lane_ops / 39.3e12,
                                    "ops_per_cell": [round(ops_future, 4), round(ops_last, 4)],
                                    "register_block": {"actions": R_, "states_per_lane": S_},
                                    "note": "secondary: the north star prices this path against HBM"}
        # SURVEY.md section 8(d) asks for four numbers side by side
        flops_per_cell = {"cfg2": 14, "cfg5": 14, "cfg4": 14, "cfg4p": 14, "cfg3": 25}.get(args.workload, 14)
        out["side_by_side"] = {
            "cells_per_s": out["value"],
            "algorithmic_GBps": achieved * (world if world > 1 else 1),
            "hbm_measured_GBps": (traffic / (avg_launch_ms * 1e-3) / 1e9) if (traffic and avg_launch_ms > 0) else None,
            "fp64_TFLOPs_at_reference_op_count": out["value"] * flops_per_cell / 1e12,
            "reference_fp64_ops_per_cell": flops_per_cell,
            # what a period must move at the very least: read V_{t+1} once, write V_t and the policy (20 B per state)
            "compulsory_bytes_per_launch": 20.0 * states_step_rank / T,
            "compulsory_GBps": (20.0 * states_step_rank / T / (avg_launch_ms * 1e-3) / 1e9) if avg_launch_ms > 0 else None,
            "note": "the reference's formulas spend 14 (F1/F2) / 25 (F3) fp64 operations per cell; the kernels execute fewer "
                    "(identical operations are formed once), see valu_roofline",
        }
        if check is not None:
            out["check_vs_single_rank"] = check
        if not args.no_cpu_baseline:
            wb = make_workload(args, 1)
            out["cpu_baseline"] = cpu_baseline(wb, args.cpu_seconds)
        if world == 1 and args.workload == "cfg2" and not args.no_target_grid:
            out["north_star_grid"] = target_grid_probe(sia, dev)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    backend.close()


if __name__ == "__main__":
    main()

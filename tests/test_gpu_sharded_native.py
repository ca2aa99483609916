"""The multi-GPU entry points of include/sdpgpu.h (csrc/sdpgpu_comm.hip) on ONE GPU, every family:

* sdpgpu_solve_multi with N rank-handles that share the device (slabs exchanged by device copies inside the library),
* sdpgpu_comm_init + sdpgpu_solve_sharded with a one-rank RCCL communicator (the collective path itself),

each against the oracle, table for table, bit for bit.  What one GPU cannot show is RCCL between devices: the slab
arithmetic, the row that travels (fp64 values or order-preserving keys), the in-place offsets and the deferred
read-out are all exercised here; the collective's transport is RCCL's own business.
"""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _oracle_tables(oracle, w):
    V, pol, _ = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve()
    return V, pol


def _check_rank(eng, V, pol, T, first_full, label):
    for period in range(1, T + 1):
        _, lo, hi = eng.slab(period)
        gv = eng.values(period)
        a, b = (0, len(gv)) if period >= first_full else (lo, hi)
        assert np.array_equal(gv[a:b], V[period - 1][a:b]), f"{label}: V_{period}"
        assert np.array_equal(eng.policy(period), pol[period - 1][lo:hi]), f"{label}: policy of period {period}"


@pytest.mark.parametrize("make,world", [(cases.f1_small, 2), (cases.f1_clsp_main, 4), (cases.f1_unclamped, 3),
                                        (cases.f1_gapped, 2), (cases.f2_clamped, 3), (cases.f2_unclamped, 2),
                                        (cases.f2_pipeline, 3), (cases.f3_tenths, 3), (cases.f3_dyadic, 2),
                                        (cases.f3_testing, 5), (cases.f4_overdraft, 2), (cases.f5_cash_leadtime, 2),
                                        (cases.f6_survival, 3), (cases.f3_dyadic_wide, 3), (cases.f3_grid_prices, 4),
                                        (cases.f2_clamped, 5)], ids=lambda v: getattr(v, "__name__", str(v)))
@pytest.mark.parametrize("threads", [False, True], ids=["one-thread", "thread-per-rank"])
def test_solve_multi_shared_device(sia, oracle, make, world, threads):
    """Both ways of driving the ranks from one process: ONE host thread issuing for all (default), and one host thread
    per rank inside the library (SDPGPU_SHARDED_THREADS: each runs sdpgpu_solve_sharded's sweep; the copies of a shared
    device are ordered by a per-period rendezvous of the threads)."""
    w = make()
    V, pol = _oracle_tables(oracle, w)
    engs = []
    try:
        for r in range(world):
            d = w.desc()
            d.rank, d.world_size, d.device = r, world, 0
            engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))
        for rep in range(2):  # the second call reuses the group
            sia.SdpEngine.solve_multi(engs, sync=True, threads=threads)
            cells = 0
            for r, e in enumerate(engs):
                _check_rank(e, V, pol, w.T, 2, f"rank {r}/{world} (call {rep})")
                cells += int(e.stats().cells_evaluated)
            assert cells == int(engs[0].stats().cells_all_ranks)
        sia.SdpEngine.solve_multi(engs, sync=True, gather_first=True, threads=threads)
        for r, e in enumerate(engs):
            _check_rank(e, V, pol, w.T, 1, f"rank {r}/{world} (V_1 gathered)")
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("make", [cases.f1_clsp_main, cases.f2_clamped, cases.f3_tenths, cases.f5_cash_leadtime],
                         ids=lambda v: v.__name__)
@pytest.mark.parametrize("overlap", [False, True])
def test_solve_sharded_one_rank_communicator(sia, oracle, make, overlap):
    w = make()
    V, pol = _oracle_tables(oracle, w)
    d = w.desc()
    d.device = 0
    with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
        eng.comm_prepare()  # (the non-collective half: tables, device, RCCL load -- the ranks agree on it before comm_init)
        eng.comm_prepare()  # idempotent
        eng.comm_init(sia.SdpEngine.comm_unique_id(), 0, 1)
        eng.solve_sharded(overlap=overlap, sync=True, gather_first=True)
        _check_rank(eng, V, pol, w.T, 1, "one-rank communicator")
        eng.comm_destroy()
        with pytest.raises(sia.SdpgpuError):
            eng.solve_sharded()


@pytest.mark.parametrize("make", ["staff_planning_small", "staff_testing_small", "staff_short_rows"])
def test_staff_family_slabs(sia, make):
    """The level-dependent pmf family cut into slabs (its footprint spans the whole table: blocking exchange)."""
    from oracle import staffref
    import staff_cases
    staffref.build()
    c = getattr(staff_cases, make)()
    V, pol, _ = c.oracle_problem(staffref).solve()
    engs = []
    try:
        for r in range(3):
            d = c.functor.to_desc(c.T)
            d.rank, d.world_size, d.device = r, 3, 0
            engs.append(sia.SdpEngine(d, None, [float(m) for m in c.functor.minStaffNum], level_pmf=c.table,
                                      level_row_len=c.row_len))
        sia.SdpEngine.solve_multi(engs, sync=True)
        for r, e in enumerate(engs):
            _check_rank(e, V, pol, c.T, 2, f"{make} rank {r}/3")
    finally:
        for e in engs:
            e.close()


def test_comm_argument_errors(sia):
    w = cases.f1_small()
    d = w.desc()
    d.device = 0
    with sia.SdpEngine(d, w.pmf) as eng:
        with pytest.raises(sia.SdpgpuError):
            eng.solve_sharded()                    # no communicator
        with pytest.raises(sia.SdpgpuError):
            eng.exchange(1)
        uid = sia.SdpEngine.comm_unique_id()
        with pytest.raises(sia.SdpgpuError):
            eng.comm_init(uid, 1, 2)               # the handle was created as rank 0 of 1
    d2 = w.desc()
    d2.rank, d2.world_size, d2.device = 1, 2, 0
    with sia.SdpEngine(d2, w.pmf) as e1, sia.SdpEngine(w.desc(), w.pmf) as e0:
        with pytest.raises(sia.SdpgpuError):
            sia.SdpEngine.solve_multi([e0, e1])    # e0 is rank 0 of ONE


def test_solve_multi_refuses_handles_of_different_problems(sia):
    """handles[r] must describe ONE problem: a second handle with another grid (or pmf size) would make the exchange copy
    S_pad / n elements of handle 0's row into a shorter row.  SDPGPU_ERR_ARG, nothing launched."""
    w = cases.f1_small()
    other = cases.f1_clsp_main()
    other.pmf = other.pmf[:w.T] if len(other.pmf) >= w.T else other.pmf + [other.pmf[-1]] * (w.T - len(other.pmf))
    d0, d1 = w.desc(), other.desc()
    d0.rank, d0.world_size, d0.device = 0, 2, 0
    d1.rank, d1.world_size, d1.device = 1, 2, 0
    assert d0.periods == d1.periods
    with sia.SdpEngine(d0, w.pmf) as e0, sia.SdpEngine(d1, other.pmf) as e1:
        for threads in (False, True):
            with pytest.raises(sia.SdpgpuError) as ei:
                sia.SdpEngine.solve_multi([e0, e1], threads=threads)
            assert ei.value.code == 1 and "another problem" in ei.value.message


def test_solve_multi_threads_reports_a_failing_rank(sia, monkeypatch):
    """One rank's sweep fails (a forced window plan that does not exist): the other ranks' threads are released from the
    rendezvous and the call returns the failing rank's error instead of hanging."""
    w = cases.f1_clsp_main()
    engs = []
    try:
        for r in range(3):
            d = w.desc()
            d.rank, d.world_size, d.device = r, 3, 0
            if r == 1:
                monkeypatch.setenv("SDPGPU_WIN_R", "7")  # (read at create: only rank 1's handle carries it)
            else:
                monkeypatch.delenv("SDPGPU_WIN_R", raising=False)
            engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))
        with pytest.raises(sia.SdpgpuError) as ei:
            sia.SdpEngine.solve_multi(engs, sync=True, threads=True)
        assert "rank 1" in ei.value.message and "no instantiation" in ei.value.message
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("threads", [False, True], ids=["one-thread", "thread-per-rank"])
@pytest.mark.parametrize("name,world", [("target_8_slabs", 8), ("cfg4_4_slabs", 4), ("cfg3t_8_slabs", 8)])
def test_baseline_shardings_rehearsed_on_one_device(sia, name, world, threads):
    """The shardings BASELINE.json names -- 8 slabs of the F1 target shape (500 actions x 200 demands: the (4, 8) block, chunk
    rows and key rows per slab), 4 slabs of configs[3]'s lead-time shape, 8 slabs of CashConstraint.main's shape (ragged action
    counts along the cash axis) -- at reduced width, all ranks on ONE device (slabs exchanged by copies inside
    sdpgpu_solve_multi), against the single-rank sweep of the same library, every table bit for bit.  (The single-rank sweep
    is what the oracle tests pin; what one device cannot show is RCCL's transport between devices.)"""
    from stochastic_inventory_amd import workloads
    w = {"target_8_slabs": lambda: workloads.target_grid(T=3, S=400000),
         "cfg4_4_slabs": lambda: workloads.cfg4_leadtime(T=3, NX=600, A=120, D=100),
         "cfg3t_8_slabs": lambda: workloads.cfg3_tenths(T=3, NX=40, maxCash=400.0, A=60, D=25)}[name]()
    d1 = w.desc()
    d1.device = 0
    engs = []
    with sia.SdpEngine(d1, w.pmf, w.overhead()) as ref:
        ref.solve(sync=True)
        try:
            for r in range(world):
                d = w.desc()
                d.rank, d.world_size, d.device = r, world, 0
                engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))
            sia.SdpEngine.solve_multi(engs, sync=True, gather_first=True, threads=threads)
            cells = 0
            for r, e in enumerate(engs):
                for period in range(1, w.T + 1):
                    _, lo, hi = e.slab(period)
                    assert np.array_equal(e.values(period), ref.values(period)), f"{name} rank {r}: V_{period}"
                    assert np.array_equal(e.policy(period), ref.policy(period)[lo:hi]), f"{name} rank {r}: policy {period}"
                cells += int(e.stats().cells_evaluated)
            assert cells == int(ref.stats().cells_evaluated)
        finally:
            for e in engs:
                e.close()


@pytest.mark.parametrize("name,world", [("configs3_pipeline_4_slabs", 4), ("configs4_1e8_states_8_slabs", 8)])
def test_full_size_baseline_shardings_on_one_device(sia, name, world):
    """BASELINE.json's two sharded configs at FULL width, all ranks on one device (slabs exchanged by copies inside
    sdpgpu_solve_multi): configs[3] -- the 250 x 200 x 200 pipeline state, 1e7 states, 200 actions, 100 demands, cut into 4
    slabs -- and configs[4] -- 1e8 states x 500 x 200 on ping-pong tables, cut into 8 slabs, three periods (3e13 cells) --
    against the single-rank sweep of the same library: the last two value tables and this rank's policy slabs, bit for
    bit.  What one device cannot show is RCCL's transport (tests/test_gpu_multirank.py has the two-device test)."""
    from stochastic_inventory_amd import workloads
    if name.startswith("configs3"):
        w, store_all = workloads.cfg4_pipeline(T=2), 1
    else:
        w, store_all = workloads.cfg5_scaled(S=100_000_000, T=3), 0

    def desc(rank, n):
        d = w.desc()
        d.rank, d.world_size, d.device, d.store_all_values = rank, n, 0, store_all
        return d

    periods = (1, 2)
    with sia.SdpEngine(desc(0, 1), w.pmf, w.overhead()) as ref:
        ref.solve(sync=True)
        want_v = {p: ref.values(p) for p in periods}
        want_pol = {p: ref.policy(p) for p in periods}
        cells = int(ref.stats().cells_evaluated)
    engs = []
    try:
        for r in range(world):
            engs.append(sia.SdpEngine(desc(r, world), w.pmf, w.overhead()))
        sia.SdpEngine.solve_multi(engs, sync=True, gather_first=True)
        total = 0
        for r, e in enumerate(engs):
            for p in periods:
                _, lo, hi = e.slab(p)
                assert np.array_equal(e.values(p), want_v[p]), f"{name} rank {r}: V_{p}"
                assert np.array_equal(e.policy(p), want_pol[p][lo:hi]), f"{name} rank {r}: policy of period {p}"
            total += int(e.stats().cells_evaluated)
        assert total == cells
    finally:
        for e in engs:
            e.close()

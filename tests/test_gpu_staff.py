"""GPU parity of the STAFF family (workforce.StaffRecursion on the HIP engine) against oracle/staffref.c: every
period's value table and policy bit for bit, the mirror class, slabs, and the full WorkforcePlanning.main size."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import staff_cases  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def staffref():
    from oracle import staffref as m
    m.build()
    return m


def _engine(sia, c, rank=0, world=1):
    d = c.functor.to_desc(c.T)
    d.rank, d.world_size = rank, world
    return sia.SdpEngine(d, None, [float(m) for m in c.functor.minStaffNum], level_pmf=c.table,
                         level_row_len=c.row_len)


@pytest.mark.parametrize("make", staff_cases.ALL, ids=lambda m: m.__name__)
def test_tables_match_oracle(sia, staffref, make):
    c = make()
    P = c.oracle_problem(staffref)
    V, pol, cells = P.solve()
    with _engine(sia, c) as eng:
        eng.solve(sync=True)
        for period in range(1, c.T + 1):
            assert eng.grid(period)[:2] == (float(P.x_lo[period - 1]), int(P.nx[period - 1]))
            assert np.array_equal(eng.values(period), V[period - 1]), period
            assert np.array_equal(eng.policy(period), pol[period - 1]), period
        st = eng.stats()
        assert st.cells_evaluated == cells == st.cells_all_ranks


def test_mirror_class_reads_like_the_reference_driver(sia, staffref):
    """WorkforcePlanning.java:104-112: construct, getExpectedValue(initialState), getAction, getOptTable."""
    c = staff_cases.staff_testing_small()
    f = c.functor
    P = c.oracle_problem(staffref)
    V, pol, _ = P.solve()
    # the reference's own pmf shape: double[T][xLength][i + 1][2]
    pmf = [[[[j, c.table[t, i, j]] for j in range(i + 1)] for i in range(c.table.shape[1])] for t in range(c.T)]
    recursion = sia.StaffRecursion(f.feasibleActions, f.stateTransition, f.immediateValue, pmf, c.T, functor=f)
    initialState = sia.StaffState(1, f.iniStaffNum)
    opt = recursion.getExpectedValue(initialState)
    assert opt == V[0][0] and recursion.getAction(initialState) == pol[0][0]
    table = recursion.getOptTable()
    width = int(P.x_lo[-1] + P.nx[-1])
    _, _, val, act, seen, _ = P.memo(width)
    want = [(t + 1, x, act[t, x]) for t in range(c.T) for x in np.nonzero(seen[t])[0]]
    assert [tuple(r) for r in table.tolist()] == [(float(a), float(b), float(q)) for a, b, q in want]
    with pytest.raises(KeyError):
        recursion.getExpectedValue(sia.StaffState(1, 5))  # period 1 holds the initial staff number only
    with pytest.raises(NotImplementedError):
        recursion.getExpectedValue(initialState, 3)
    recursion.close()


def test_reachable_interval_clamped(sia, staffref):
    c = staff_cases.staff_rates()
    P = c.oracle_problem(staffref)
    _, _, _, _, seen, _ = P.memo(int(P.x_lo[-1] + P.nx[-1]))
    with _engine(sia, c) as eng:
        for period in range(1, c.T + 1):
            assert np.array_equal(eng.reachable(period).astype(bool), seen[period - 1][: int(P.nx[period - 1])]), period


@pytest.mark.parametrize("make,world", [(staff_cases.staff_planning_small, 2), (staff_cases.staff_testing_small, 3),
                                        (staff_cases.staff_wide_actions, 4)], ids=lambda v: getattr(v, "__name__", str(v)))
def test_slabs(sia, staffref, make, world):
    """world_size N slabs from one process, the all-gather played by hand (as tests/test_gpu_parity.py does)."""
    import torch
    c = make()
    V, pol, _ = c.oracle_problem(staffref).solve()
    engs = [_engine(sia, c, r, world) for r in range(world)]
    bufs = []
    for e in engs:
        t = torch.zeros(e.values_bytes() // 8, dtype=torch.float64, device="cuda")
        e.attach_values(t.data_ptr(), t.numel() * 8)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        bufs.append(t)
    for period in range(c.T, 0, -1):
        rows = []
        for r, e in enumerate(engs):
            e.run_period(period)
            pad, _, _ = e.slab(period)
            base = (e.exchange_ptr(period) - bufs[r].data_ptr()) // 8
            rows.append(bufs[r][base: base + pad])
        torch.cuda.synchronize()
        full = torch.zeros_like(rows[0])
        for r, e in enumerate(engs):
            _, lo, hi = e.slab(period)
            full[lo:hi] = rows[r][lo:hi]
        for r in range(world):
            rows[r].copy_(full)
    torch.cuda.synchronize()
    for period in range(1, c.T + 1):
        for e in engs:
            assert np.array_equal(e.values(period), V[period - 1])
        assert np.array_equal(np.concatenate([e.policy(period) for e in engs]), pol[period - 1])
    assert sum(e.stats().cells_evaluated for e in engs) == engs[0].stats().cells_all_ranks
    for e in engs:
        e.close()


def test_workforce_planning_main_size(sia, staffref):
    """WorkforcePlanning.java:33-50 at full size: T = 3, staff 0..600, hires 0..500, Binomial(y, 0.5) turnover --
    5.4e8 cells, every table bit for bit."""
    from stochastic_inventory_amd.pmf import staff_level_pmf
    f = sia.StaffFunctor(fixCost=100, unitVariCost=10, salary=20, unitPenalty=80, minStaffNum=[40, 40, 40], maxHireNum=500,
                         minX=0, maxX=600, clampStaff=True, iniStaffNum=0)
    c = staff_cases.StaffCase("workforce_planning_main", f, staff_level_pmf([0.5, 0.5, 0.5], 601))
    V, pol, cells = c.oracle_problem(staffref).solve(nthreads=16)
    with _engine(sia, c) as eng:
        eng.solve(sync=True)
        for period in range(1, 4):
            assert np.array_equal(eng.values(period), V[period - 1]) and np.array_equal(eng.policy(period), pol[period - 1])
        assert eng.stats().cells_evaluated == cells


def test_misuse_is_reported(sia):
    c = staff_cases.staff_planning_small()
    d = c.functor.to_desc(c.T)
    with pytest.raises(ValueError):
        sia.SdpEngine(d, None, [8.0] * c.T)  # no level table
    with pytest.raises(sia.SdpgpuError):
        sia.SdpEngine(d, None, [8.0] * c.T, level_pmf=c.table, level_row_len=np.arange(31, dtype=np.int32) + 2)  # j > y
    with pytest.raises(sia.SdpgpuError):
        eng = sia.SdpEngine(d, None, [8.5] * c.T, level_pmf=c.table)  # minStaffNum is an int[]
        eng.solve(sync=True)
    with _engine(sia, c) as eng:
        eng.solve(sync=True)
        with pytest.raises(sia.SdpgpuError):
            eng.eval_states(1, [3.0], [0.0], [0.0])
        assert eng.footprint(1) is None


def _random_case(seed):
    rng = np.random.default_rng(seed)
    T = int(rng.integers(1, 5))
    clamp = bool(rng.integers(0, 2))
    rows = int(rng.integers(2, 70))
    max_hire = int(rng.integers(0, 40))
    rates = rng.uniform(0.05, 0.95, size=T)
    table = staff_cases.staff_level_pmf(list(rates), rows)
    row_len = None
    if rng.integers(0, 3) == 0:  # truncated rows (renormalised), shorter than y + 1
        cap = int(rng.integers(1, rows + 1))
        row_len = np.minimum(np.arange(rows) + 1, cap).astype(np.int32)
        t2 = np.zeros((T, rows, cap))
        for y in range(rows):
            n = row_len[y]
            t2[:, y, :n] = table[:, y, :n] / table[:, y, :n].sum(axis=1, keepdims=True)
        table = t2
    if clamp:
        min_x = int(rng.integers(0, 10))
        max_x = min_x + int(rng.integers(0, 150))
        ini = int(rng.integers(min_x, max_x + 1))
    else:  # a start well above the longest row: the next period's box then begins above zero
        min_x = max_x = 0
        ini = int(rng.integers(0, 3 * rows))
    f = staff_cases.StaffFunctor(fixCost=float(rng.integers(0, 200)), unitVariCost=float(rng.integers(0, 30)) / 4,
                                 salary=float(rng.integers(0, 40)) / 2, unitPenalty=float(rng.integers(0, 300)),
                                 minStaffNum=[int(v) for v in rng.integers(0, 60, size=T)], maxHireNum=max_hire,
                                 minX=min_x, maxX=max_x, clampStaff=clamp, iniStaffNum=ini)
    return staff_cases.StaffCase(f"staff_fuzz_{seed}", f, table, row_len)


def test_random_instances(sia, staffref):
    """Seeded random instances: clamped and not, truncated rows, tables shorter than the staff range, starts above the
    longest row, zero hires, single periods; whole tables and cell counts, slabs on every third one."""
    for seed in range(int(os.environ.get("SDP_FUZZ_N", "60"))):  # soak: SDP_FUZZ_N=3000
        c = _random_case(seed)
        V, pol, cells = c.oracle_problem(staffref).solve()
        world = 1 + (seed % 3 == 0) * (1 + seed % 4)
        got_cells = 0
        for rank in range(world):
            with _engine(sia, c, rank, world) as eng:
                if world == 1:
                    eng.solve(sync=True)
                    for period in range(1, c.T + 1):
                        assert np.array_equal(eng.values(period), V[period - 1]), (seed, period)
                        assert np.array_equal(eng.policy(period), pol[period - 1]), (seed, period)
                else:  # a slab of the last period only (no exchange needed there)
                    eng.run_period(c.T)
                    _, lo, hi = eng.slab(c.T)
                    assert np.array_equal(eng.values(c.T)[lo:hi], V[c.T - 1][lo:hi]), (seed, rank)
                    assert np.array_equal(eng.policy(c.T), pol[c.T - 1][lo:hi]), (seed, rank)
                got_cells += eng.stats().cells_evaluated
        if world == 1:
            assert got_cells == cells, seed


def test_two_states_per_lane_kernel_on_every_case(sia, staffref, monkeypatch):
    """staff_pair_kernel (two adjacent states per lane, 16-byte gathers) is chosen by itself only on large staff ranges;
    forced here on every named case and on the random ones: the clamp folds at the table's last row and at both ends
    of the staff range, odd slab ends, truncated rows."""
    monkeypatch.setenv("SDPGPU_STAFF_PAIR", "1")
    cases = [m() for m in staff_cases.ALL] + [_random_case(s) for s in range(100, 140)]
    for c in cases:
        V, pol, cells = c.oracle_problem(staffref).solve()
        with _engine(sia, c) as eng:
            eng.solve(sync=True)
            for period in range(1, c.T + 1):
                assert np.array_equal(eng.values(period), V[period - 1]), (c.name, period)
                assert np.array_equal(eng.policy(period), pol[period - 1]), (c.name, period)
    c = staff_cases.staff_wide_actions()
    V, pol, _ = c.oracle_problem(staffref).solve()
    for world in (2, 3):  # slabs with odd lengths: a lane's second state may belong to the next rank
        for rank in range(world):
            with _engine(sia, c, rank, world) as eng:
                eng.run_period(c.T)
                _, lo, hi = eng.slab(c.T)
                assert np.array_equal(eng.values(c.T)[lo:hi], V[c.T - 1][lo:hi]) and np.array_equal(eng.policy(c.T), pol[c.T - 1][lo:hi])


@pytest.mark.parametrize("win", ["4", "2"], ids=["S4", "S2"])
def test_window_kernel_on_every_case(sia, staffref, monkeypatch, win):
    """staff_window_kernel (S adjacent states per lane, one probability per LEVEL from an LDS-staged piece of the table row,
    immediate costs handed down the diagonal) is chosen by itself on staff ranges of 256 numbers and more (two or four states per lane by size); forced here on
    every named case and on random ones -- the fold at the table's last row, clamped and unclamped staff ranges, penalties,
    truncated rows, tiles and action blocks with ragged ends -- and on slabs with odd lengths."""
    monkeypatch.setenv("SDPGPU_STAFF_PAIR", "1")
    monkeypatch.setenv("SDPGPU_STAFF_WIN", win)
    cases = [m() for m in staff_cases.ALL] + [_random_case(s) for s in range(200, 260)]
    for c in cases:
        V, pol, cells = c.oracle_problem(staffref).solve()
        with _engine(sia, c) as eng:
            eng.solve(sync=True)
            assert eng.stats().cells_evaluated == cells, c.name
            for period in range(1, c.T + 1):
                assert np.array_equal(eng.values(period), V[period - 1]), (c.name, period)
                assert np.array_equal(eng.policy(period), pol[period - 1]), (c.name, period)
    c = staff_cases.staff_wide_actions()
    V, pol, _ = c.oracle_problem(staffref).solve()
    for world in (2, 3):
        for rank in range(world):
            with _engine(sia, c, rank, world) as eng:
                eng.run_period(c.T)
                _, lo, hi = eng.slab(c.T)
                assert np.array_equal(eng.values(c.T)[lo:hi], V[c.T - 1][lo:hi]) and np.array_equal(eng.policy(c.T), pol[c.T - 1][lo:hi])


def test_lanes_are_actions_kernel_on_every_case(sia, staffref, monkeypatch):
    """staff_action_kernel (the lanes of a wave are 64 consecutive actions of ONE state; chosen by itself for periods of at most 16
    states, period 1 of every run from one initial staff number) forced on every named case and on random ones: the wave's
    arg-min with the reference's first-best rule, action ranges that are not multiples of 64, rows of unequal lengths, the
    table's last row, both clamps -- and on slabs."""
    monkeypatch.setenv("SDPGPU_STAFF_LANES", "1")
    cases = [m() for m in staff_cases.ALL] + [_random_case(s) for s in range(300, 340)]
    for c in cases:
        V, pol, cells = c.oracle_problem(staffref).solve()
        with _engine(sia, c) as eng:
            eng.solve(sync=True)
            assert eng.stats().cells_evaluated == cells, c.name
            for period in range(1, c.T + 1):
                assert np.array_equal(eng.values(period), V[period - 1]), (c.name, period)
                assert np.array_equal(eng.policy(period), pol[period - 1]), (c.name, period)
    c = staff_cases.staff_wide_actions()
    V, pol, _ = c.oracle_problem(staffref).solve()
    for world in (2, 3):
        for rank in range(world):
            with _engine(sia, c, rank, world) as eng:
                eng.run_period(c.T)
                _, lo, hi = eng.slab(c.T)
                assert np.array_equal(eng.values(c.T)[lo:hi], V[c.T - 1][lo:hi]) and np.array_equal(eng.policy(c.T), pol[c.T - 1][lo:hi])


def test_kernel_forms_chosen_by_range_size(sia, staffref, monkeypatch):
    """Left alone the launcher picks the form by the period's staff range (one state: lanes = actions; a few hundred numbers: two
    states per lane from a window; ~1500 and more: four) -- a run whose periods cross all of them, against the oracle, and against
    the same run with the window form and the lanes-are-actions form switched off."""
    T = 4  # staff ranges of 1, 601, 1201 and 1801 numbers
    f = staff_cases.StaffFunctor(fixCost=50, unitVariCost=20, salary=5, unitPenalty=250, minStaffNum=[40, 90, 60, 30], maxHireNum=600,
                                 clampStaff=False, iniStaffNum=0)
    c = staff_cases.StaffCase("staff_forms", f, staff_cases.staff_level_pmf([0.3] * T, 601))
    V, pol, cells = c.oracle_problem(staffref).solve()
    assert [len(v) for v in V] == [1, 601, 1201, 1801]
    with _engine(sia, c) as eng:
        eng.solve(sync=True)
        got = [(eng.values(t).copy(), eng.policy(t).copy()) for t in range(1, c.T + 1)]
        assert eng.stats().cells_evaluated == cells
    for t in range(c.T):
        assert np.array_equal(got[t][0], V[t]) and np.array_equal(got[t][1], pol[t])
    # (the window kernel's blocks beyond the table's last row -- here every tile from staff number 600 up -- read one scalar
    # probability a step instead of a staged row piece: switched off, the same tables)
    monkeypatch.setenv("SDPGPU_STAFF_UNI", "0")
    with _engine(sia, c) as eng:
        eng.solve(sync=True)
        for t in range(1, c.T + 1):
            assert np.array_equal(eng.values(t), got[t - 1][0]) and np.array_equal(eng.policy(t), got[t - 1][1])
    monkeypatch.setenv("SDPGPU_STAFF_WIN", "0")
    monkeypatch.setenv("SDPGPU_STAFF_LANES", "0")
    with _engine(sia, c) as eng:
        eng.solve(sync=True)
        for t in range(1, c.T + 1):
            assert np.array_equal(eng.values(t), got[t - 1][0]) and np.array_equal(eng.policy(t), got[t - 1][1])


def test_ping_pong_tables(sia, staffref):
    """store_all_values = 0: two value rows reused period after period; V_1 (and the policy of every period) must not
    care."""
    c = staff_cases.staff_testing_small()
    V, pol, _ = c.oracle_problem(staffref).solve()
    d = c.functor.to_desc(c.T)
    d.store_all_values = 0
    with sia.SdpEngine(d, None, [float(m) for m in c.functor.minStaffNum], level_pmf=c.table) as eng:
        eng.solve(sync=True)
        assert np.array_equal(eng.values(1), V[0])
        for period in range(1, c.T + 1):
            assert np.array_equal(eng.policy(period), pol[period - 1])
        with pytest.raises(sia.SdpgpuError):
            eng.values(c.T)  # overwritten long ago

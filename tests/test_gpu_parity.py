"""Parity tests proper: the HIP path through the C ABI against the CPU oracle, same inputs.

Bar (BASELINE.json north_star): value tables within 1e-9 relative, policy indices bit-exact.
The kernels keep the reference's operation order, so the values are in fact bit-identical and
the tests assert exact equality; the 1e-9 tolerance is stated once, in _assert_tables.
"""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

REL_TOL = 1e-9  # north_star tolerance for value tables (we hold the stronger bit-for-bit bar)


def _assert_tables(gv, gp, ov, op, what):
    assert gp.dtype == np.int32 and np.array_equal(gp, op), f"{what}: policy indices differ"
    scale = np.maximum(np.abs(ov), 1e-300)
    assert np.all(np.abs(gv - ov) <= REL_TOL * scale), f"{what}: values beyond 1e-9 relative"
    assert np.array_equal(gv, ov), f"{what}: values are within tolerance but not bit-identical"


def _solve_both(sia, oracle, w, kernel=0, nthreads=8):
    desc = w.desc()
    desc.kernel = kernel
    eng = sia.SdpEngine(desc, w.pmf, w.overhead())
    eng.solve(sync=True)
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    V, pol, cells = P.solve(nthreads=nthreads)
    return eng, P, V, pol, cells


@pytest.mark.parametrize("make", cases.ALL, ids=lambda f: f.__name__)
@pytest.mark.parametrize("kernel", [0, 1], ids=["auto", "gather"])
def test_small_cases_bit_exact(sia, oracle, make, kernel):
    w = make()
    eng, P, V, pol, cells = _solve_both(sia, oracle, w, kernel)
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} t={period}")
    st = eng.stats()
    assert st.cells_evaluated == cells == st.cells_all_ranks
    assert st.periods_run == w.T
    eng.close()


def test_clsp_main_instance(sia, oracle):
    """The instance left in CLSP.java:196-211 (601 states, 61 actions, 4 periods)."""
    w = cases.f1_clsp_main()
    eng, P, V, pol, _ = _solve_both(sia, oracle, w)
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"clsp t={period}")
    eng.close()


def test_cfg1_full(sia, oracle):
    from stochastic_inventory_amd import workloads
    w = workloads.cfg1_sS()
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert cells == 200 * 101 * 25 * 12
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"cfg1 t={period}")
    eng.close()


@pytest.mark.parametrize("kernel", [1, 0], ids=["gather", "auto"])
def test_cfg2_shape_reduced_horizon(sia, oracle, kernel):
    """configs[1] at full width (1e4 x 200 x 100) for 3 periods: 6e8 cells, seconds on 8 host threads."""
    from stochastic_inventory_amd import workloads
    w = workloads.cfg2_clsp(T=3)
    eng, P, V, pol, _ = _solve_both(sia, oracle, w, kernel)
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"cfg2 t={period}")
    eng.close()


def test_cfg2_full_horizon_full_tables(sia, oracle):
    """configs[1] exactly as bench.py's secondary entry runs it: 1e4 x 200 x 100 at the FULL horizon of 52 periods
    (1.04e10 cells; the oracle sweeps it in seconds on the box's threads) -- every table of every period, bit for bit."""
    import os
    from stochastic_inventory_amd import workloads
    w = workloads.cfg2_clsp()
    assert w.T == 52
    eng, P, V, pol, cells = _solve_both(sia, oracle, w, 0, nthreads=min(os.cpu_count() or 1, 16))
    assert cells == 10000 * 200 * 100 * 52 == eng.stats().cells_evaluated
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"cfg2 t={period}")
    eng.close()


@pytest.mark.parametrize("env", [{"SDPGPU_CASH_PAIR_S": "2"}, {"SDPGPU_CASH_PAIR_S": "1"}, {"SDPGPU_CASH_PAIR": "0"},
                                 {"SDPGPU_CASH_PAIR": "0", "SDPGPU_CASH_UNI": "0"}, {"SDPGPU_CASH_BANDS": "0"},
                                 {"SDPGPU_CASH_BANDS": "3", "SDPGPU_CASH_PAIR_S": "2"}, {"SDPGPU_CASH_SHARE": "1"},
                                 {"SDPGPU_CASH_SHARE": "1", "SDPGPU_CASH_PAIR_S": "1"}, {"SDPGPU_CASH_TAB": "1"},
                                 {"SDPGPU_CASH_TAB": "1", "SDPGPU_CASH_PAIR_S": "1"}, {"SDPGPU_CASH_SLOTS": "1"},
                                 {"SDPGPU_CASH_SLOTS": "1", "SDPGPU_CASH_PAIR_S": "1"}],
                         ids=lambda e: ",".join(f"{k[12:]}={v}" for k, v in e.items()))
@pytest.mark.parametrize("make", [cases.f3_grid_prices, cases.f3_half_grid_prices, cases.f3_testing, cases.f3_xr],
                         ids=lambda f: f.__name__)
def test_cash_row_kernel_variants(sia, oracle, monkeypatch, make, env):
    """Every variant of the cash row kernels (one / two points per lane, one / two tiles per wave, with and without the
    uniform-key trips, banded and row-major block order, the (row, action) operand blocks formed per wave (default), shared
    by the four tiles of a workgroup (SHARE=1) or copied from cash_row_table_kernel's table (TAB=1)) gives the oracle's tables bit for bit -- on grids with on-grid
    prices (all trips uniform), half-grid prices (uniform and tie steps mixed), integer cash and the (x, R) state."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = make()
    if make is cases.f3_testing or make is cases.f3_xr:  # widen the cash axis so that the two-point kernels apply
        w.functor.maxCashState = 700.0
    eng, P, V, pol, _ = _solve_both(sia, oracle, w)
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} t={period}")
    eng.close()


def _od_cases():
    """F4 / F5 instances for the two-point row kernel (round 4): balances on both sides of zero (uniform-key trips where a
    tile pays no interest under the order, the quantiser elsewhere), deposit interest on positive balances (r0 != 0: no tile is
    ever interest-free above zero), an interest-free overdraft band, a fixed order cost (F4), half-grid prices (tie steps)."""
    import copy
    out = []
    w = cases.f5_cash_leadtime()
    out.append(("f5_spl_shape", w))
    w = cases.f5_cash_leadtime()
    w.functor.r0 = 0.02
    out.append(("f5_deposit_interest", w))
    w = cases.f5_cash_leadtime()
    w.functor.interestFreeAmount = 3.0
    w.functor.price = 4.5
    out.append(("f5_interest_free_band_half_grid_price", w))
    w = cases.f5_cash_leadtime()
    w.functor.price = 3.333
    out.append(("f5_off_grid_price", w))
    w = cases.f4_overdraft()
    w.functor.cashRoundIntDiv = False  # tenths grid, no long division: the two-point kernel's quantiser
    w.functor.r0 = 0.0
    out.append(("f4_tenths", w))
    w = cases.f4_overdraft()
    w.functor.cashRoundIntDiv = False
    w.functor.fixOrderCost = 2.5
    w.functor.cashRoundMult = w.functor.cashRoundDiv = 100.0
    w.functor.minCashState, w.functor.maxCashState = -20.0, 30.0
    out.append(("f4_hundredths_fixed_cost_deposit", w))
    return out


@pytest.mark.parametrize("env", [{}, {"SDPGPU_CASH_OD_PAIR": "0"}, {"SDPGPU_CASH_RW": "1"}, {"SDPGPU_CASH_DIAG_ORDER": "0"},
                                 {"SDPGPU_CASH_PAIR_S": "1"}, {"SDPGPU_CASH_PAIR_S": "2"}, {"SDPGPU_CASH_DIAG_SEG": "3"},
                                 {"SDPGPU_CASH_SLOTS": "1"}, {"SDPGPU_CASH_SLOTS": "1", "SDPGPU_CASH_RW": "1", "SDPGPU_CASH_PAIR_S": "2"},
                                 {"SDPGPU_CASH_BANDS": "0", "SDPGPU_CASH_DIAG_ORDER": "0"}],
                         ids=lambda e: ",".join(f"{k[12:]}={v}" for k, v in e.items()) or "default")
@pytest.mark.parametrize("name", [n for n, _ in _od_cases()])
def test_overdraft_pair_kernel_variants(sia, oracle, monkeypatch, name, env):
    """F4 / F5 on cash_row_pair_kernel (round 4) in every launch form -- one / two tiles per wave, one row per workgroup or four
    rows of a level (RW), diagonal unit order with odd segment sizes or the band numbering, one or two setup slots, and the
    one-point row kernel of rounds 1-3 (OD_PAIR=0) -- against the oracle, every table bit for bit."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = dict(_od_cases())[name]
    eng, P, V, pol, _ = _solve_both(sia, oracle, w)
    assert eng.stats().kernel_used == 2
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{name} t={period}")
    eng.close()


@pytest.mark.parametrize("pair_s", ["2", "1", None], ids=["two-tiles-per-wave", "one-tile", "planned"])
def test_cash_row_wide_pmf_above_64KiB_of_lds(sia, oracle, monkeypatch, pair_s):
    """A 345-point pmf on CashConstraint.main's tenths grid: 152 B of per-wave entries per demand point + the read-out
    scratch of 128- or 256-point tiles = 60-66 KiB of LDS.  The launcher sizes the LDS from the tile it actually picked and
    raises the kernel's limit above 64 KiB (round 2 budgeted for 64-point tiles and failed the launch at 66,168 B)."""
    from stochastic_inventory_amd import workloads
    from stochastic_inventory_amd.workloads import truncated_poisson_tile
    if pair_s:
        monkeypatch.setenv("SDPGPU_CASH_PAIR_S", pair_s)
    w = workloads.cfg3_tenths(T=2, NX=50, maxCash=1100.0, A=8, D=345)
    w.pmf = [truncated_poisson_tile(170.0, 345), truncated_poisson_tile(150.0, 345)]
    eng, P, V, pol, cells = _solve_both(sia, oracle, w, nthreads=16)
    assert eng.stats().cells_evaluated == cells and eng.stats().kernel_used == 2
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} pair_s={pair_s} t={period}")
    eng.close()


@pytest.mark.parametrize("D", [3, 4, 5, 31, 32, 33, 64])
def test_cash_row_pair_setup_slots_around_half_a_wave(sia, oracle, D):
    """The pair kernel forms the entries of TWO actions per setup pass when a pmf fits half a wave (lanes 0-31 / 32-63): pmfs of
    3 .. 5 points (trips of four and single tail steps), 31, 32 (the last width with two slots), 33 and 64 points (one slot), on
    CashConstraint.main's tenths grid with an odd number of actions (the last pass has no second action) -- every table against the oracle."""
    from stochastic_inventory_amd import workloads
    from stochastic_inventory_amd.workloads import truncated_poisson_tile
    w = workloads.cfg3_tenths(T=3, NX=40, maxCash=300.0, A=23, D=D)
    w.pmf = [truncated_poisson_tile(m, D) for m in (max(1.0, D / 3.0), max(1.0, D / 2.5), max(1.0, D / 4.0))]
    eng, P, V, pol, cells = _solve_both(sia, oracle, w, nthreads=8)
    assert eng.stats().cells_evaluated == cells and eng.stats().kernel_used == 2
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} D={D} t={period}")
    eng.close()


def _cfg3_small():
    from stochastic_inventory_amd import workloads
    return workloads.cfg3_cash(T=3, NX=24, NC=700, A=70, D=30)


def _dyadic_wide_min():
    """f3_dyadic_wide under OptDirection.MIN (the arg-min instantiation of the diagonal kernel)."""
    from stochastic_inventory_amd.functors import OptDirection
    w = cases.f3_dyadic_wide()
    w.direction = OptDirection.MIN
    w.name = "f3_dyadic_wide_min"
    return w


@pytest.mark.parametrize("env", [{}, {"SDPGPU_CASH_DIAG_S": "2"}, {"SDPGPU_CASH_DIAG_S": "1"}, {"SDPGPU_CASH_DIAG": "0"},
                                 {"SDPGPU_CASH_DIAG_BANDS": "1"}],
                         ids=lambda e: ",".join(f"{k[12:]}={v}" for k, v in e.items()) or "default")
@pytest.mark.parametrize("make", [_cfg3_small, cases.f3_dyadic_wide, _dyadic_wide_min, cases.f3_dyadic_big_fixed], ids=lambda f: f.__name__)
def test_cash_diag_kernel_variants(sia, oracle, monkeypatch, make, env):
    """The diagonal form of the uniform-shift kernel (cash_diag_kernel: a block of consecutive actions reads one staged row
    segment per demand step; one and two tiles per wave) and the per-cell gather form it replaces give the oracle's tables
    bit for bit: configs[2]'s family on 700 cash points, a half-unit grid with fractional slides and a fixed-cost break, and
    a grid whose fixed cost forces the direct-gather steps."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = make()
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert eng.stats().cells_evaluated == cells
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} t={period}")
    eng.close()


def test_cfg3_shape_reduced(sia, oracle):
    """configs[2] family (2-D inventory x cash, ragged action counts) at 40 x 600 states."""
    from stochastic_inventory_amd import workloads
    w = workloads.cfg3_cash(T=3, NX=40, NC=600, A=60, D=30)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert eng.stats().cells_evaluated == cells
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"cfg3 t={period}")
    eng.close()


def test_cfg4_shape_reduced(sia, oracle):
    from stochastic_inventory_amd import workloads
    w = workloads.cfg4_leadtime(T=3, NX=120, A=40, D=30)
    eng, P, V, pol, _ = _solve_both(sia, oracle, w)
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"cfg4 t={period}")
    eng.close()


@pytest.mark.parametrize("win_s", [None, "4", "2", "1"], ids=["planned", "S4", "S2", "S1"])
@pytest.mark.parametrize("pipeline", [False, True], ids=["leadtime1", "pipeline"])
def test_f2_row_window_states_per_lane(sia, oracle, monkeypatch, pipeline, win_s):
    """The F2 row-window kernel with one, two and four adjacent states per lane (wave-private row staging, 16-byte LDS reads
    of two demand steps for S >= 2) on rows wide enough for 256-state tiles, ragged at the end: every table against the oracle."""
    from stochastic_inventory_amd import workloads
    if win_s:
        monkeypatch.setenv("SDPGPU_WIN_S", win_s)
    w = workloads.cfg4_pipeline(T=3, NX=300, A=10, D=22) if pipeline else workloads.cfg4_leadtime(T=3, NX=300, A=37, D=30)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert eng.stats().cells_evaluated == cells and eng.stats().kernel_used == 2
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} S={win_s} t={period}")
    eng.close()


@pytest.mark.parametrize("env", [{"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8"}, {"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8", "SDPGPU_WIN_NCH": "3"},
                                 {"SDPGPU_WIN_R": "8", "SDPGPU_WIN_S": "4"}, {"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "4", "SDPGPU_WIN_NCH": "1"},
                                 {"SDPGPU_WIN_R": "5", "SDPGPU_WIN_S": "1"}, {"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "2", "SDPGPU_WIN_NCH": "4"}],
                         ids=lambda e: ",".join(f"{k[11:]}={v}" for k, v in e.items()))
@pytest.mark.parametrize("shape", ["min", "max", "gapped", "unclamped"])
def test_f1_window_register_blocks(sia, oracle, monkeypatch, shape, env):
    """The F1 window kernel under forced plans -- eight states per lane (the plan of the large grids), eight actions by four
    states, one task per tile, five actions per lane, several chunks per tile (key rows + deferred read-out): every table
    against the oracle, on a grid of 1601 states (ragged against the 64- to 512-state tiles) with 38 actions (ragged against
    every register block) and 23 demand points, under MIN and MAX, with a gapped support and with the unclamped transition."""
    from stochastic_inventory_amd.functors import BackorderFunctor
    from stochastic_inventory_amd.states import OptDirection
    from stochastic_inventory_amd.workloads import Workload
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    f = BackorderFunctor(fixedOrderingCost=30, variOrderingCost=1.5, holdingCost=1, penaltyCost=7, minInventory=-700,
                         maxInventory=900, maxOrderQuantity=37, iniInventory=5, clampInventory=shape != "unclamped")
    pmf = cases.discrete_pmf(3, [1, 4, 5, 9, 16, 22], [0.1, 0.2, 0.3, 0.2, 0.15, 0.05]) if shape == "gapped" else cases._pmf([9, 12, 7], 23)
    w = Workload(f"f1_ragged_{shape}", f, OptDirection.MAX if shape == "max" else OptDirection.MIN, pmf)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert eng.stats().cells_evaluated == cells
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} {env} t={period}")
    eng.close()


@pytest.mark.parametrize("env,store_all", [({"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8", "SDPGPU_WIN_NCH": "1"}, 1),
                                           ({"SDPGPU_WIN_R": "8", "SDPGPU_WIN_S": "4", "SDPGPU_WIN_NCH": "1"}, 1),
                                           ({"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8", "SDPGPU_WIN_NCH": "2"}, 1),
                                           ({}, 0)],
                         ids=["4x8-one-task-78KiB", "8x4-one-task", "4x8-two-chunks", "ping-pong-planned"])
def test_f1_window_plans_above_64KiB_of_lds(sia, oracle, monkeypatch, env, store_all):
    """Workgroups of more than 64 KiB of LDS (gfx950 has 160 KiB per compute unit): the 500-action x 200-demand shape of the
    target grid with ONE task per tile on the (4, 8) block (78.4 KiB; no chunk rows, no key atomics, no finalize pass), its
    neighbours, and the plan the library itself takes with ping-pong tables (which cannot keep chunk rows) -- on 1601 states
    (ragged against the 512-state tile), every table against the oracle."""
    from stochastic_inventory_amd import workloads
    for k in ("SDPGPU_WIN_R", "SDPGPU_WIN_S", "SDPGPU_WIN_NCH"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = workloads.cfg5_scaled(S=1601, T=3)
    desc = w.desc()
    desc.store_all_values = store_all
    eng = sia.SdpEngine(desc, w.pmf, w.overhead())
    pl = eng.plan(1)
    assert pl.kernel == 2
    if env.get("SDPGPU_WIN_S") == "8" and env.get("SDPGPU_WIN_NCH") == "1":
        assert pl.lds_bytes == 80288 and pl.workgroups_per_cu == 2
    if not store_all:
        assert pl.chunks == 1  # (a grid this small takes a finer block than (4, 8); the large ones: test_planner_geometry.py)
    eng.solve(sync=True)
    V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=8)
    assert eng.stats().cells_evaluated == cells
    for period in ((1, 2) if not store_all else range(1, w.T + 1)):  # (ping-pong: the last two tables are the ones held)
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} {env} t={period}")
    eng.close()


def test_infeasible_forced_plan_is_refused_before_launch(sia, monkeypatch):
    """3000 actions x 400 demand steps forced into one chunk on the (4, 8) block: 251 KiB of windows.  sdpgpu_run_period
    returns SDPGPU_ERR_ARG with the planner's reason (was: a launch error from the runtime)."""
    from stochastic_inventory_amd import workloads
    for k, v in {"SDPGPU_WIN_R": "4", "SDPGPU_WIN_S": "8", "SDPGPU_WIN_NCH": "1"}.items():
        monkeypatch.setenv(k, v)
    w = workloads.cfg5_scaled(S=2000, T=2, A=3000, D=400)
    eng = sia.SdpEngine(w.desc(), w.pmf, w.overhead())
    with pytest.raises(sia.SdpgpuError) as ei:
        eng.solve(sync=True)
    assert ei.value.code == 1 and "of LDS per workgroup" in ei.value.message
    eng.close()


@pytest.mark.parametrize("env", [{"SDPGPU_WIN_R": "8", "SDPGPU_WIN_S": "2"}, {"SDPGPU_WIN_NCH": "1"},
                                 {"SDPGPU_WIN_NCH": "2", "SDPGPU_WIN_S": "4"}, {"SDPGPU_WIN_R": "5", "SDPGPU_WIN_S": "4"}],
                         ids=lambda e: ",".join(f"{k[11:]}={v}" for k, v in e.items()))
def test_f2_row_window_block_plans(sia, oracle, monkeypatch, env):
    """The F2 row-window kernel's other plans: eight actions per lane with 120 demand points (the LDS budget then lets only
    three of a workgroup's four waves take action blocks), one or two chunks per tile (every wave walks several blocks and
    re-stages its rows between them), five actions per lane."""
    from stochastic_inventory_amd import workloads
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = workloads.cfg4_leadtime(T=2, NX=140, A=43, D=120)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert eng.stats().cells_evaluated == cells and eng.stats().kernel_used == 2
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} {env} t={period}")
    eng.close()


@pytest.mark.parametrize("win_s", ["4", "2"], ids=["S4", "S2"])
def test_f2_row_window_above_64KiB_of_lds(sia, oracle, monkeypatch, win_s):
    """400 demand points: a wave's four row segments are 21 KiB, a workgroup's 69-86 KiB -- above the 64 KiB a launch could ask
    for before the launcher raised the kernel's dynamic-LDS limit (three of the four waves take action blocks at S = 4)."""
    from stochastic_inventory_amd import workloads
    monkeypatch.setenv("SDPGPU_WIN_S", win_s)
    w = workloads.cfg4_leadtime(T=2, NX=300, A=22, D=400)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert eng.stats().cells_evaluated == cells and eng.stats().kernel_used == 2
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} S={win_s} t={period}")
    eng.close()


@pytest.mark.parametrize("kernel", [0, 1], ids=["auto", "gather"])
def test_cfg4_pipeline_shape_reduced(sia, oracle, kernel):
    """configs[3] as a two-stage pipeline (x, q1, q2) at 90 x 24 x 24 states (the full shape is 250 x 200 x 200)."""
    from stochastic_inventory_amd import workloads
    w = workloads.cfg4_pipeline(T=3, NX=90, A=24, D=30)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w, kernel)
    assert eng.stats().cells_evaluated == cells == 3 * 90 * 24 * 24 * 24 * 30
    assert eng.stats().kernel_used == (2 if kernel == 0 else 1)
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"cfg4p t={period}")
    eng.close()


def test_pipeline_eval_reachable_simulate(sia, oracle):
    """Lead time 2: off-grid queries, the reachable set from (x, q1, q2) = (2, 3, 1) and the policy rollout."""
    w = cases.f2_pipeline()
    eng, P, V, pol, _ = _solve_both(sia, oracle, w)
    rng = np.random.default_rng(11)
    for period in (1, 2, w.T):
        x, _, q1 = P.state_arrays(period)
        q2 = P.preq2_array(period)
        pick = rng.integers(0, len(x), size=53)
        v_next = V[period] if period < w.T else None
        ov, oa = P.eval_states(period, v_next, x[pick], None, q1[pick], q2[pick])
        gv, ga = eng.eval_states(period, x[pick], None, q1[pick], q2[pick])
        _assert_tables(gv, ga, ov, oa, f"pipeline eval t={period}")
        assert np.array_equal(gv, V[period - 1][pick])
    reach = P.reachable()
    for period in range(1, w.T + 1):
        assert np.array_equal(eng.reachable(period), reach[period - 1]), f"pipeline reach t={period}"
    assert reach[1].sum() > 1 and reach[0].sum() == 1
    dem = rng.integers(0, 10, size=(200, w.T)).astype(np.float64)
    disc = np.ones(w.T)
    f = w.functor
    osum, ovalid = P.simulate(V, pol, dem, disc, f.iniInventory, 0.0, f.iniPreQ)
    gsum, gvalid = eng.simulate(dem, disc, f.iniInventory, 0.0, f.iniPreQ)
    assert np.array_equal(gsum, osum) and np.array_equal(gvalid, ovalid) and ovalid.all()
    eng.close()


@pytest.mark.parametrize("make", [cases.f1_small, cases.f3_tenths, cases.f5_cash_leadtime, cases.f2_unclamped,
                                  cases.f6_survival_gamma],
                         ids=lambda f: f.__name__)
def test_eval_states_off_grid(sia, oracle, make):
    """getExpectedValue(state) for states that are not grid points (off-grid initial cash etc.)."""
    w = make()
    eng, P, V, pol, _ = _solve_both(sia, oracle, w)
    rng = np.random.default_rng(7)
    for period in (1, w.T):
        x, cash, preq = P.state_arrays(period)
        pick = rng.integers(0, len(x), size=37)
        xs, cs, qs = x[pick].copy(), cash[pick].copy(), preq[pick].copy()
        if w.desc().family in (3, 4, 5, 6):
            cs = cs + 0.013  # off the cash grid
        v_next = V[period] if period < w.T else None
        ov, oa = P.eval_states(period, v_next, xs, cs, qs)
        gv, ga = eng.eval_states(period, xs, cs, qs)
        _assert_tables(gv, ga, ov, oa, f"{w.name} eval t={period}")
    eng.close()


@pytest.mark.parametrize("make", [cases.f1_small, cases.f2_unclamped, cases.f3_testing, cases.f5_cash_leadtime,
                                  cases.f6_survival],
                         ids=lambda f: f.__name__)
def test_reachable_set(sia, oracle, make):
    w = make()
    eng, P, V, pol, _ = _solve_both(sia, oracle, w)
    reach = P.reachable()
    for period in range(1, w.T + 1):
        assert np.array_equal(eng.reachable(period), reach[period - 1]), f"{w.name} t={period}"
    eng.close()


def test_wide_ranges_fall_back_to_the_gather_kernel(sia, oracle):
    """2401 actions x 1200 demand points exceed the window kernel's LDS span."""
    w = cases.f1_wide()
    eng, P, V, pol, _ = _solve_both(sia, oracle, w)
    assert eng.stats().kernel_used == 1
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"wide t={period}")
    eng.close()


@pytest.mark.parametrize("make", [cases.f2_clamped, cases.f1_clsp_main, cases.f3_testing], ids=lambda f: f.__name__)
def test_ping_pong_tables(sia, oracle, make):
    w = make()
    d = w.desc()
    d.store_all_values = 0
    eng = sia.SdpEngine(d, w.pmf, w.overhead())
    eng.solve()
    V, pol, _ = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve()
    _assert_tables(eng.values(1), eng.policy(1), V[0], pol[0], "ping-pong t=1")
    assert np.array_equal(eng.values(2), V[1])
    with pytest.raises(sia.SdpgpuError):
        eng.values(3)  # overwritten by V_1
    for period in range(1, w.T + 1):
        assert np.array_equal(eng.policy(period), pol[period - 1])
    eng.close()


@pytest.mark.parametrize("make,world", [(cases.f3_tenths, 3), (cases.f2_clamped, 3), (cases.f2_unclamped, 2),
                                        (cases.f1_small, 2), (cases.f1_clsp_main, 4), (cases.f1_unclamped, 2),
                                        (cases.f3_testing, 3), (cases.f3_dyadic, 2), (cases.f5_cash_leadtime, 2),
                                        (cases.f2_pipeline, 3), (cases.f6_survival, 3)],
                         ids=lambda v: getattr(v, "__name__", str(v)))
def test_sharded_periods_single_process(sia, oracle, make, world):
    """world_size N slabs driven from one process: each rank computes its slab into its own copy of
    V_t, the test plays the all-gather by hand.  Covers slab bounds, padding and the policy slabs
    (for F2 the slabs cut through preQ rows; for F1 through window tiles)."""
    _run_sharded(sia, oracle, make(), world)


@pytest.mark.parametrize("family", [1, 2, 3])
def test_sharded_random_instances(sia, oracle, family):
    """Seeded random instances (tests/test_gpu_fuzz.py) cut into 2..5 slabs: ragged slabs, empty slabs, slabs
    smaller than a tile, key rows and value rows, split and whole periods."""
    import test_gpu_fuzz as tf
    for seed in range(12):
        w = tf.make_instance(family, 100 + seed)
        _run_sharded(sia, oracle, w, 2 + seed % 4)


@pytest.mark.parametrize("make,world,k", [(cases.f1_small, 2, 2), (cases.f1_clsp_main, 3, 2), (cases.f1_clsp_main, 4, 4),
                                          (cases.f1_gapped, 2, 3), (cases.f1_max, 2, 3)],
                         ids=lambda v: getattr(v, "__name__", str(v)))
def test_widened_slabs_k_periods_per_exchange(sia, oracle, make, world, k):
    """sdpgpu_run_period_range / sdpgpu_footprint / sdpgpu_set_halo: K periods between exchanges.  The test plays
    the all-gathers by hand and as LATE as the schedule allows (a block's rows only when the next block starts, the
    rest at the very end), so a period that read a row too early would read zeros."""
    _run_blocked(sia, oracle, make(), world, k)


def test_widened_slabs_random_instances(sia, oracle):
    import test_gpu_fuzz as tf
    done = 0
    for seed in range(40):
        w = tf.make_instance(1, 200 + seed)
        _run_blocked(sia, oracle, w, 2 + seed % 3, 1 + seed % 4)  # (no footprint, e.g. unclamped: returns at once)
        done += 1 if w.functor.clampInventory else 0
    assert done >= 20


def _run_blocked(sia, oracle, w, world, k):
    import torch
    from stochastic_inventory_amd.sharded import ShardedSolver, SlabBackend
    engs = []
    for r in range(world):
        d = w.desc()
        d.rank, d.world_size = r, world
        engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))

    class Geo(SlabBackend):  # footprints only: plan_blocks is the production code
        T = w.T

        def footprint(self, period):
            return engs[0].footprint(period)

    plan = ShardedSolver(Geo()).plan_blocks(k)
    if plan is None:  # no bounded footprint (a support too sparse for the window kernel): nothing to test
        for e in engs:
            e.close()
        return
    blocks, halo = plan
    V, pol, _ = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve()
    bufs, kbufs = [], []
    for e in engs:
        e.set_halo(halo)
        t = torch.zeros(e.values_bytes() // 8, dtype=torch.float64, device="cuda")
        e.attach_values(t.data_ptr(), t.numel() * 8)
        kb = torch.zeros(max(e.keys_bytes() // 8, 1), dtype=torch.int64, device="cuda")
        if e.keys_bytes():
            e.attach_keys(kb.data_ptr(), kb.numel() * 8)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        bufs.append(t)
        kbufs.append(kb)

    def row_of(r, period):
        ptr = engs[r].exchange_ptr(period)
        pad, _, _ = engs[r].slab(period)
        for arena in (bufs[r], kbufs[r]):
            if arena.data_ptr() <= ptr < arena.data_ptr() + arena.numel() * 8:
                base = (ptr - arena.data_ptr()) // 8
                return arena[base: base + pad].view(torch.int64)
        raise AssertionError("exchange row outside the attached arenas")

    def gather(period):
        torch.cuda.synchronize()
        rows = [row_of(r, period) for r in range(world)]
        full = torch.zeros_like(rows[0])
        for r, e in enumerate(engs):
            _, lo, hi = e.slab(period)
            full[lo:hi] = rows[r][lo:hi]
        for r in range(world):
            rows[r].copy_(full)

    late = []
    boundary = None
    for t_hi, t_lo, ext in blocks:
        if boundary is not None:
            gather(boundary)
        for period in range(t_hi, t_lo - 1, -1):
            for e in engs:
                _, lo, hi = e.slab(period)
                S = e.num_states(period)
                a, b = max(0, lo - ext[period][0]), min(S, hi + ext[period][1])
                if hi > lo:
                    e.run_period_range(period, a, b)
                else:
                    e.run_period_range(period, min(lo, S), min(lo, S))
            if period != t_lo:
                late.append(period)
        boundary = t_lo
    for period in late + [boundary]:
        gather(period)
    for e in engs:
        e.finalize()
    torch.cuda.synchronize()
    for period in range(1, w.T + 1):
        for e in engs:
            assert np.array_equal(e.values(period), V[period - 1]), (w.name, period)
        got = np.concatenate([e.policy(period) for e in engs])
        assert np.array_equal(got, pol[period - 1]), (w.name, period)
    for e in engs:
        e.close()


def _run_sharded(sia, oracle, w, world):
    import torch
    engs = []
    for r in range(world):
        d = w.desc()
        d.rank, d.world_size = r, world
        engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))
    V, pol, _ = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve()
    bufs, kbufs = [], []
    for e in engs:
        t = torch.zeros(e.values_bytes() // 8, dtype=torch.float64, device="cuda")
        e.attach_values(t.data_ptr(), t.numel() * 8)
        k = torch.zeros(max(e.keys_bytes() // 8, 1), dtype=torch.int64, device="cuda")
        if e.keys_bytes():
            e.attach_keys(k.data_ptr(), k.numel() * 8)
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        bufs.append(t)
        kbufs.append(k)

    def row_of(r, period):  # the 8-byte-element row rank r exchanges after run_period(period)
        ptr = engs[r].exchange_ptr(period)
        pad, _, _ = engs[r].slab(period)
        for arena in (bufs[r], kbufs[r]):
            if arena.data_ptr() <= ptr < arena.data_ptr() + arena.numel() * 8:
                base = (ptr - arena.data_ptr()) // 8
                return arena[base: base + pad].view(torch.int64)
        raise AssertionError("exchange row outside the attached arenas")

    for period in range(w.T, 0, -1):
        for r, e in enumerate(engs):
            if period < w.T and (r + period) % 2 == 0:  # mix whole-period and interior+boundary launches
                e.run_period_part(period, sia._abi.PART_INTERIOR)
                e.run_period_part(period, sia._abi.PART_BOUNDARY)
            else:
                e.run_period(period)
        torch.cuda.synchronize()
        rows = [row_of(r, period) for r in range(world)]
        full = torch.zeros_like(rows[0])
        for r, e in enumerate(engs):
            _, lo, hi = e.slab(period)
            full[lo:hi] = rows[r][lo:hi]
        for r in range(world):
            rows[r].copy_(full)  # the all-gather, by hand
    for e in engs:
        e.finalize()
    torch.cuda.synchronize()
    for period in range(1, w.T + 1):
        for e in engs:
            assert np.array_equal(e.values(period), V[period - 1])  # every rank holds the whole table
        got = np.concatenate([e.policy(period) for e in engs])
        assert np.array_equal(got, pol[period - 1])
    for e in engs:
        e.close()


def test_mirror_api_drop_in(sia, oracle):
    """The reference-shaped classes: construct with lambdas + functor, query like CLSPTesting.java:111-118."""
    w = cases.f1_small()
    f = w.functor
    T = w.T
    rec = sia.Recursion(sia.OptDirection.MIN, w.pmf,
                        lambda s: f.feasibleActions(s, T),
                        lambda s, a, r: f.stateTransition(s, a, r, T),
                        lambda s, a, r: f.immediateValue(s, a, r, T), functor=f)
    assert rec.validateFunctor(64) == 64
    ini = sia.State(1, f.iniInventory)
    m = oracle.Problem(w.desc(), w.pmf).memo()
    assert rec.getExpectedValue(ini) == m["value"]
    assert rec.getAction(ini) == m["action"]
    table = rec.getOptTable()
    assert table.shape == (m["n"], 3)
    order = np.lexsort((m["x"], m["period"]))
    assert np.array_equal(table[:, 0], m["period"][order].astype(float))
    assert np.array_equal(table[:, 1], m["x"][order])
    assert np.array_equal(table[:, 2], m["actions"][order])
    assert table[0].tolist()[:2] == [1.0, f.iniInventory]  # FitsS.java:102-106 relies on row 0
    acts = rec.getCacheActions()
    assert acts[ini] == m["action"] and len(acts) == m["n"]


def test_mirror_cash_and_leadtime(sia, oracle):
    w = cases.f3_tenths()
    rec = sia.CashRecursion(sia.OptDirection.MAX, w.pmf, functor=w.functor, discountFactor=1.0)
    rec.setTreeMapCacheAction()
    ini = sia.CashState(1, w.functor.iniInventory, w.functor.iniCash)
    m = oracle.Problem(w.desc(), w.pmf, w.overhead()).memo()
    assert rec.getExpectedValue(ini) == m["value"] and rec.getAction(ini) == m["action"]
    off = sia.CashState(1, 0.0, 4.93)  # not a multiple of 0.1: answered by eval_states
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    V, _, _ = P.solve()
    ov, oa = P.eval_states(1, V[1], [0.0], [4.93], [0.0])
    assert rec.getExpectedValue(off) == ov[0] and rec.getAction(off) == oa[0] * w.functor.stepSize
    t = rec.getOptTable()
    assert t.shape == (m["n"], 4)

    w2 = cases.f2_unclamped()
    lt = sia.LeadtimeRecursion(w2.pmf, functor=w2.functor)
    m2 = oracle.Problem(w2.desc(), w2.pmf).memo()
    ini2 = sia.LeadtimeState(1, 0.0, 0.0)
    assert lt.getExpectedValue(ini2) == m2["value"] and lt.getAction(ini2) == m2["action"]
    t2 = lt.getOptTable()
    assert t2.shape == (m2["n"], 4)
    order = np.lexsort((m2["preq"], m2["x"], m2["period"]))
    assert np.array_equal(t2[:, 1], m2["x"][order]) and np.array_equal(t2[:, 3], m2["actions"][order])

    w3 = cases.f5_cash_leadtime()
    cl = sia.CashLeadtimeRecursion(w3.pmf, functor=w3.functor)
    m3 = oracle.Problem(w3.desc(), w3.pmf, w3.overhead()).memo()
    ini3 = sia.CashLeadtimeState(1, 0.0, 0.0, 0.0)
    assert cl.getExpectedValue(ini3) == m3["value"] and cl.getAction(ini3) == m3["action"]
    assert cl.getOptTable().shape == (m3["n"], 5)


def test_mirror_risk_recursion(sia, oracle):
    """RiskRecursion.getSurvProb (cashSurvival.java's driver shape): survival probability, first-period order,
    opt table over the visited (never bankrupt) states; ties between actions are the rule here (many actions
    reach probability 1), so the lowest-index rule is exercised on nearly every state."""
    w = cases.f6_survival()
    f = w.functor
    T = w.T
    rec = sia.RiskRecursion(w.pmf, lambda s: f.feasibleActions(s, T), lambda s, a, r: f.stateTransition(s, a, r, T),
                            lambda s, a, r: f.immediateValue(s, a, r, T), functor=f)
    rec.setTreeMapCacheAction()
    assert rec.validateFunctor(64) == 64
    ini = sia.RiskState(1, f.iniInventory, f.iniCash, False)
    m = oracle.Problem(w.desc(), w.pmf, w.overhead()).memo()
    assert 0.0 < m["value"] < 1.0
    assert rec.getSurvProb(ini) == m["value"] and rec.getAction(ini) == m["action"]
    t = rec.getOptTable()
    assert t.shape == (m["n"], 5) and not t[:, 3].any() and (t[:, 2] >= 0).all()
    order = np.lexsort((m["cash"], m["x"], m["period"]))
    assert np.array_equal(t[:, 1], m["x"][order]) and np.array_equal(t[:, 2], m["cash"][order])
    assert np.array_equal(t[:, 4], m["actions"][order])
    with pytest.raises(AttributeError):
        rec.getExpectedValue(ini)
    # RiskSimulation.simulateLostSale: device rollout == oracle rollout == the reference's loop on the host lambdas
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    V, pol, _ = P.solve()
    rng = np.random.default_rng(5)
    dem = rng.integers(0, 11, size=(300, T)).astype(np.float64)
    osum, ovalid = P.simulate(V, pol, dem, np.ones(T), f.iniInventory, f.iniCash)
    gsum, gvalid = rec.engine.simulate(dem, np.ones(T), f.iniInventory, f.iniCash)
    assert np.array_equal(gsum, osum) and np.array_equal(rec.engine.last_sim_flags, P.last_sim_flags)
    assert 0 < gsum.sum() < len(gsum) and ((P.last_sim_flags >> 1) & 1).any()
    sim = sia.RiskSimulation([sia.PoissonDist(m) for m in (4, 6, 3, 5)], 300, rec)
    res = sim.simulateLostSaleOnDemands(ini, dem)
    assert res[0] == 1 - osum.sum() / 300 and res == sim.simulateOnHost(ini, dem)
    est = sim.simulateLostSale(ini)  # LHS paths from the Poisson quantiles: close to the computed probability
    assert abs(est[0] - m["value"]) < 0.12


def test_cfg2_full_horizon_properties(sia, oracle):
    """configs[1] at BASELINE size (1e4 x 200 x 100, 52 periods = 1.04e10 cells): too big for the
    oracle to sweep in test time, so check (i) the two kernels against each other bit for bit,
    (ii) 2,000 sampled states per checked period against the oracle fed the GPU's own V_{t+1},
    (iii) the backorder structure: order-up-to levels never exceed capacity, V_t >= 0."""
    from stochastic_inventory_amd import workloads
    w = workloads.cfg2_clsp()
    da, dg = w.desc(), w.desc()
    dg.kernel = sia.KERNEL_GATHER
    ea, eg = sia.SdpEngine(da, w.pmf), sia.SdpEngine(dg, w.pmf)
    ea.solve()
    eg.solve()
    P = oracle.Problem(w.desc(), w.pmf)
    rng = np.random.default_rng(11)
    assert ea.stats().cells_evaluated == 10000 * 200 * 100 * 52
    for period in (52, 51, 40, 17, 2, 1):
        va, pa = ea.values(period), ea.policy(period)
        assert np.array_equal(va, eg.values(period)) and np.array_equal(pa, eg.policy(period))
        x, _, _ = P.state_arrays(period)
        pick = np.unique(np.concatenate([rng.integers(0, len(x), size=2000), [0, 1, len(x) - 2, len(x) - 1]]))
        v_next = ea.values(period + 1) if period < w.T else None
        ov, oa = P.eval_states(period, v_next, x[pick])
        _assert_tables(va[pick], pa[pick], ov, oa, f"cfg2 full t={period}")
        assert (va >= 0).all() and pa.min() >= 0 and pa.max() <= 199
    ea.close()
    eg.close()


def test_kernel_selection(sia):
    """The specialised kernels are the ones that run where they apply (kernel_used: 1 gather, 2 specialised)."""
    expect = {cases.f1_small: 2, cases.f1_gapped: 2, cases.f1_sparse_support: 1, cases.f1_unclamped: 2, cases.f1_edge_single: 2, cases.f2_clamped: 2,
              cases.f2_unclamped: 2, cases.f2_pipeline: 2,
              cases.f3_tenths: 2,      # tenths, end-cash penalty: cash row kernel
              cases.f3_row: 2,         # tenths, no penalty: cash row kernel
              cases.f3_testing: 2, cases.f3_dyadic: 2, cases.f3_min_gamma: 2, cases.f4_overdraft: 2,
              cases.f5_cash_leadtime: 2, cases.f6_survival: 2, cases.f6_survival_gamma: 2}
    for make, kind in expect.items():
        w = make()
        with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
            eng.solve()
            assert eng.stats().kernel_used == kind, w.name


def _graph_cases():
    from stochastic_inventory_amd import workloads
    return [("cfg2_chunked_keys", lambda: workloads.cfg2_clsp(T=5, S=3000), 1), ("f1_one_task_per_tile", lambda: workloads.cfg5_scaled(S=70000, T=3, A=40, D=30), 1),
            ("f1_ping_pong", lambda: workloads.cfg5_scaled(S=70000, T=4, A=40, D=30), 0),
            ("f2_leadtime", lambda: workloads.cfg4_leadtime(T=3, NX=120, A=40, D=30), 1), ("f2_pipeline", lambda: workloads.cfg4_pipeline(T=3, NX=90, A=24, D=30), 1),
            ("f3_dyadic_diag", _cfg3_small, 1), ("f3_tenths_pair", lambda: workloads.cfg3_tenths(T=3, NX=12, maxCash=60.0, A=9, D=12), 1),
            ("f4_overdraft", cases.f4_overdraft, 1), ("f5_level_order", cases.f5_cash_leadtime, 1), ("f6_survival", cases.f6_survival, 1)]


@pytest.mark.parametrize("name,make,store_all", _graph_cases(), ids=[c[0] for c in _graph_cases()])
def test_solve_replays_its_sweep_as_one_hip_graph(sia, oracle, monkeypatch, name, make, store_all):
    """sdpgpu_solve: call 1 eager, call 2 captured into a HIP graph while it is enqueued, calls 3.. one hipGraphLaunch each.
    Every call leaves the oracle's tables, bit for bit; whatever changes what a sweep launches (a period stepped by hand, a new
    overhead) drops the graph and the next sweeps are eager / capturing again."""
    monkeypatch.setenv("SDPGPU_GRAPH", "1")  # (opt-in: replay measured slower than the eager sweep, see sdpgpu_solve)
    w = make()
    V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=8)
    d = w.desc()
    d.store_all_values = store_all
    periods = range(1, w.T + 1) if store_all else (1, 2)

    def check(eng, what):
        for period in periods:
            _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{name} {what} t={period}")

    with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
        for call, replays in ((1, 0), (2, 1), (3, 2), (4, 3)):
            eng.solve(sync=True)
            assert eng.stats().graph_replays == replays, f"{name}: call {call}"
            check(eng, f"call {call}")
        eng.run_period(w.T)                       # a period stepped by hand: the captured sweep is dropped ...
        eng.solve(sync=True)                      # ... the next sweep is eager again
        assert eng.stats().graph_replays == 3
        check(eng, "after a hand-stepped period")
        eng.solve(sync=True)                      # captured again
        eng.solve(sync=True)
        assert eng.stats().graph_replays == 5
        check(eng, "re-captured")
        eng.set_profiling(True)                   # per-period events: eager, the graph is kept
        eng.solve(sync=True)
        assert eng.stats().graph_replays == 5 and eng.period_ms(1) > 0
        eng.set_profiling(False)
        eng.solve(sync=True)
        assert eng.stats().graph_replays == 6
        check(eng, "after a profiled sweep")
        assert eng.stats().cells_evaluated == cells


def test_graph_is_off_by_default(sia, oracle, monkeypatch):
    from stochastic_inventory_amd import workloads
    monkeypatch.delenv("SDPGPU_GRAPH", raising=False)
    w = workloads.cfg2_clsp(T=4, S=3000)
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        for _ in range(4):
            eng.solve(sync=True)
        assert eng.stats().graph_replays == 0


@pytest.mark.parametrize("family", ["f1", "f2"])
def test_pmf_too_wide_for_any_window_falls_back_to_the_generic_kernel(sia, oracle, family):
    """A pmf of 2300 (F1) / 3900 (F2) points: the window of even the smallest register block would need more than the 160 KiB
    of LDS of a compute unit.  With the automatic kernel choice such a period runs on the generic kernel (round 2: a launch
    error); asked for explicitly (kernel = WINDOW) it is SDPGPU_ERR_ARG with the planner's reason."""
    from stochastic_inventory_amd import workloads
    w = workloads.cfg5_scaled(S=600, T=2, A=6, D=2300) if family == "f1" else workloads.cfg4_leadtime(T=2, NX=80, A=4, D=3900)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert eng.stats().kernel_used == 1 and eng.stats().cells_evaluated == cells
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} t={period}")
    if family == "f1":
        assert eng.plan(1).kernel == 1
    eng.close()
    d = w.desc()
    d.kernel = 2
    with sia.SdpEngine(d, w.pmf, w.overhead()) as forced:
        with pytest.raises(sia.SdpgpuError) as ei:
            forced.solve(sync=True)
        if family == "f1":
            assert ei.value.code == 1 and "of LDS per workgroup" in ei.value.message
        else:  # (actions + demand steps beyond 3500: not a window-kernel shape at all)
            assert ei.value.code == 4


@pytest.mark.parametrize("shape", [(6, 300, 5, 900), (64, 16500, 5, 370)], ids=["D900-generic-fallback", "D370-82KiB-tiles-of-512"])
def test_uniform_shift_kernel_wide_pmfs(sia, oracle, shape):
    """cash_shift_kernel keeps 144 B of LDS per demand point: 370 points on its 512-point tiles are 82 KiB per workgroup (the
    launch raises the kernel's limit: gfx950 has 160 KiB per compute unit), 900 points are over the budget of two workgroups
    per compute unit and the period goes to the next more general kernel (round 2 launched either with more than 64 KiB and
    failed)."""
    from stochastic_inventory_amd import workloads
    NX, NC, A, D = shape
    w = workloads.cfg3_cash(T=2, NX=NX, NC=NC, A=A, D=D)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w, nthreads=16)
    assert eng.stats().cells_evaluated == cells
    assert eng.stats().kernel_used == (1 if D == 900 else 2)
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} t={period}")
    eng.close()


def test_generic_kernel_takes_the_longest_pmf(sia, oracle):
    """3950 demand points (the ABI's limit is 4000): 63 KiB of {d, p} pairs + the scratch = 66 KiB of LDS in the generic kernel."""
    from stochastic_inventory_amd import workloads
    w = workloads.cfg5_scaled(S=300, T=2, A=3, D=3950)
    eng, P, V, pol, cells = _solve_both(sia, oracle, w, kernel=1)
    assert eng.stats().kernel_used == 1 and eng.stats().cells_evaluated == cells
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} t={period}")
    eng.close()


@pytest.mark.parametrize("env", [{}, {"SDPGPU_CASH_DIAG": "0"}, {"SDPGPU_CASH_SHIFT": "0"}], ids=["auto", "no-diag", "cash-row"])
def test_dyadic_cash_grid_with_gapped_demand_support(sia, oracle, monkeypatch, env):
    """Demand supports {0, 2, 3, 7}, {1, 2, 5, 6}, ... on a dyadic cash row of 361 points: the diagonal form of the
    uniform-shift kernel (action k + i paired with demand j + i, ONE staged row per step) is only valid on consecutive
    demand values.  Round 2 took it here and got every table before period T wrong; the F3 / F4 bridge instance of round 3
    found it.  Such periods now run on the uniform-shift kernel."""
    import test_oracle_kat
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w, _ = test_oracle_kat.bridge_f3_f4_workloads()
    eng, P, V, pol, cells = _solve_both(sia, oracle, w)
    assert eng.stats().kernel_used == 2 and eng.stats().cells_evaluated == cells
    for period in range(1, w.T + 1):
        _assert_tables(eng.values(period), eng.policy(period), V[period - 1], pol[period - 1], f"{w.name} {env} t={period}")
    eng.close()


@pytest.mark.parametrize("make", [cases.f1_small, cases.f3_grid_prices, cases.f5_cash_leadtime, cases.f2_clamped], ids=lambda f: f.__name__)
def test_period_cells_add_up_to_the_sweep(sia, oracle, make):
    """sdpgpu_period_cells (ABI 6): the cells of every period, as the oracle counts them, and their sum = stats.cells_evaluated
    (F5's last period offers one order only: what bench.py prices a kernel's counters against)."""
    w = make()
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        assert eng.period_cells(1) == -1  # nothing has run
        eng.solve()
        per = [eng.period_cells(p) for p in range(1, w.T + 1)]
        assert sum(per) == eng.stats().cells_evaluated
        v = None
        for period in range(w.T, 0, -1):
            v, _, cells = P.period(period, v)
            assert per[period - 1] == cells, period

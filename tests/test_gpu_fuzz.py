"""Randomised parity: seeded random instances of every built-in family (parameters, grid bounds, rounding mode,
PMF support), each solved by the automatically selected kernel, by the generic kernel and by the CPU oracle.
All three must agree bit for bit (values) and exactly (policy indices); the north-star tolerance (1e-9 relative)
is implied.  Sizes are kept small so that the whole file takes seconds."""
import os

import numpy as np
import pytest

from stochastic_inventory_amd.functors import (BackorderFunctor, CashFunctor, CashLeadtimeFunctor, LeadtimeFunctor,
                                                OverdraftFunctor, SurvivalFunctor)
from stochastic_inventory_amd.states import OptDirection
from stochastic_inventory_amd.workloads import Workload

pytestmark = pytest.mark.gpu


def _pmf(rng, T, unit_stride=True, d_max=9):
    tiles = []
    for _ in range(T):
        n = int(rng.integers(1, d_max + 1))
        if unit_stride:
            d = np.arange(n, dtype=np.float64) + float(rng.integers(0, 3))
        else:
            d = np.sort(rng.choice(np.arange(0, 3 * d_max), size=n, replace=False)).astype(np.float64)
        p = rng.random(n) + 0.05
        p /= p.sum()
        tiles.append(np.stack([d, p], axis=1))
    return tiles


def _money(rng, lo, hi, kind):
    """A cost parameter: integer, dyadic fraction or an arbitrary decimal."""
    v = rng.uniform(lo, hi)
    if kind == "int":
        return float(round(v))
    if kind == "dyadic":
        return round(v * 8) / 8
    return round(v, 2)


def _rounding(rng):
    mode = rng.integers(0, 4)
    if mode == 0:
        return dict(cashRoundMult=1.0, cashRoundDiv=1.0, cashRoundIntDiv=True)      # Math.round(c * 1) / 1
    if mode == 1:
        return dict(cashRoundMult=10.0, cashRoundDiv=10.0, cashRoundIntDiv=False)   # Math.round(c * 10) / 10.0
    if mode == 2:
        return dict(cashRoundMult=10.0, cashRoundDiv=10.0, cashRoundIntDiv=True)    # Math.round(c * 10) / 10
    return dict(cashRoundMult=2.0, cashRoundDiv=2.0, cashRoundIntDiv=False)         # halves


def make_instance(family, seed):
    rng = np.random.default_rng(1000 * family + seed)
    T = int(rng.integers(1, 5))
    kind = ["int", "dyadic", "decimal"][int(rng.integers(0, 3))]
    if family == 1:
        clamp = bool(rng.integers(0, 4))
        f = BackorderFunctor(fixedOrderingCost=_money(rng, 0, 30, kind), variOrderingCost=_money(rng, 0, 3, kind),
                             holdingCost=_money(rng, 0, 3, kind), penaltyCost=_money(rng, 0, 12, kind),
                             minInventory=-float(rng.integers(0, 80)), maxInventory=float(rng.integers(1, 400)),
                             maxOrderQuantity=float(rng.integers(0, 70)), iniInventory=float(rng.integers(-3, 4)),
                             clampInventory=clamp)
        direction = OptDirection.MIN if rng.integers(0, 4) else OptDirection.MAX
        return Workload(f"fuzz_f1_{seed}", f, direction,
                        _pmf(rng, T, unit_stride=bool(rng.integers(0, 3)), d_max=int(rng.integers(3, 30))))
    if family == 2:
        lead2 = bool(rng.integers(0, 2))
        clamp = True if lead2 else bool(rng.integers(0, 2))
        f = LeadtimeFunctor(fixedOrderingCost=_money(rng, 0, 10, kind), variOrderingCost=_money(rng, 0, 3, kind),
                            holdingCost=_money(rng, 0, 3, kind), penaltyCost=_money(rng, 0, 12, kind),
                            maxOrderQuantity=float(rng.integers(0, 9 if lead2 else 14)), clampInventory=clamp,
                            minInventory=-float(rng.integers(0, 15)), maxInventory=float(rng.integers(1, 70)),
                            iniInventory=float(rng.integers(0, 3)), iniPreQ=0.0, leadTime=2 if lead2 else 1)
        return Workload(f"fuzz_f2_{seed}", f, OptDirection.MIN, _pmf(rng, T, unit_stride=bool(rng.integers(0, 3))))
    common = dict(price=_money(rng, 2, 12, kind), variCost=max(0.25, _money(rng, 0.5, 3, kind)),
                  salvageValue=_money(rng, 0, 1, kind), maxOrderQuantity=float(rng.integers(0, 14)),
                  minInventoryState=0.0, maxInventoryState=float(rng.integers(1, 16)),
                  minCashState=-float(rng.integers(0, 40)), maxCashState=float(rng.integers(20, 90)),
                  iniInventory=0.0, iniCash=float(rng.integers(0, 15)))
    overheads = [_money(rng, 0, 8, kind) for _ in range(T)]
    if family == 3:
        f = CashFunctor(fixOrderCost=_money(rng, 0, 5, kind), holdingCost=_money(rng, 0, 1, kind),
                        depositeRate=float(rng.choice([0, 0, 0.01])), overheadRate=float(rng.choice([0, 0, 0.05])),
                        penaltyCost=float(rng.choice([0, 0, 0.3])), discountFactor=float(rng.choice([1.0, 0.95])),
                        cashFormula=int(rng.integers(0, 2)), overheadCosts=overheads, **_rounding(rng), **common)
        direction = OptDirection.MAX if rng.integers(0, 4) else OptDirection.MIN
        return Workload(f"fuzz_f3_{seed}", f, direction, _pmf(rng, T, unit_stride=bool(rng.integers(0, 2))))
    if family == 4:
        f = OverdraftFunctor(fixOrderCost=_money(rng, 0, 5, kind), r0=float(rng.choice([0, 0.01])), r2=0.1,
                             r3=float(rng.choice([1.0, 2.0])), limit=float(rng.integers(10, 40)),
                             interestFreeAmount=float(rng.integers(0, 10)), discountFactor=float(rng.choice([1.0, 0.9])),
                             overheadCosts=overheads, **_rounding(rng), **common)
        return Workload(f"fuzz_f4_{seed}", f, OptDirection.MAX, _pmf(rng, T))
    if family == 5:
        common["maxOrderQuantity"] = float(rng.integers(0, 7))
        common["maxCashState"] = float(rng.integers(10, 30))
        common["minCashState"] = -float(rng.integers(0, 15))
        rnd = _rounding(rng)
        f = CashLeadtimeFunctor(r0=float(rng.choice([0, 0.01])), r2=0.1, r3=2.0, limit=float(rng.integers(5, 20)),
                                interestFreeAmount=float(rng.integers(0, 5)), iniPreQ=0.0,
                                overheadCosts=overheads[:min(T, 3)], **rnd, **common)
        return Workload(f"fuzz_f5_{seed}", f, OptDirection.MAX, _pmf(rng, min(T, 3), d_max=6))
    f = SurvivalFunctor(fixOrderCost=_money(rng, 0, 3, kind), holdingCost=_money(rng, 0, 1, kind),
                        depositeRate=float(rng.choice([0, 0.02])), discountFactor=float(rng.choice([1.0, 0.97])),
                        overheadCosts=[o + 4 for o in overheads], **common)
    return Workload(f"fuzz_f6_{seed}", f, OptDirection.MAX, _pmf(rng, T))


@pytest.mark.parametrize("family", [1, 2, 3, 4, 5, 6])
def test_random_instances_bit_exact(sia, oracle, family):
    n = 16 if family == 5 else 40
    if os.environ.get("SDP_FUZZ_N"):  # soak run: SDP_FUZZ_N=500 python -m pytest tests/test_gpu_fuzz.py -m gpu
        n = int(os.environ["SDP_FUZZ_N"]) // (3 if family == 5 else 1)
    kernels_seen = set()
    for seed in range(n):
        w = make_instance(family, seed)
        P = oracle.Problem(w.desc(), w.pmf, w.overhead())
        V, pol, cells = P.solve(nthreads=4)
        for kernel in (0, 1):
            d = w.desc()
            d.kernel = kernel
            with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
                eng.solve()
                if kernel == 0:
                    kernels_seen.add(eng.stats().kernel_used)
                assert eng.stats().cells_evaluated == cells, w.name
                for period in range(1, w.T + 1):
                    gv, gp = eng.values(period), eng.policy(period)
                    assert np.array_equal(gp, pol[period - 1]), f"{w.name} kernel {kernel} t={period}: policy"
                    assert np.array_equal(gv, V[period - 1]), f"{w.name} kernel {kernel} t={period}: values"
    if not any(os.environ.get(k) == "0" for k in ("SDPGPU_CASH_ROW", "SDPGPU_CASH_SHIFT")):
        assert 2 in kernels_seen  # the specialised kernels took part


def make_wide_cash_instance(seed):
    """F3 on WIDE cash rows (256 points and more: the rows the two-points-per-lane kernels and the diagonal form of the
    uniform-shift kernel take), dyadic or on-grid parameters so that those kernels are eligible: random quantum, prices,
    fixed cost (zero, small, and beyond the staged segment's reach), holding cost, horizon, demand support, direction."""
    rng = np.random.default_rng(77000 + seed)
    T = int(rng.integers(2, 4))
    dyadic = bool(rng.integers(0, 3))           # two in three: every parameter dyadic (uniform-shift / diagonal kernels)
    q = float(rng.choice([1, 1, 2, 4])) if dyadic else 10.0
    unit = 1.0 / 8 if dyadic else 0.1
    money = lambda lo, hi: float(round(rng.uniform(lo, hi) / unit) * unit)
    price = money(2, 9)
    vari = max(unit, money(0.25, min(price, 3)))
    fix = float(rng.choice([0.0, money(0, 4), money(20, 60)]))
    nc = int(rng.integers(256, 620))
    min_cash = -float(rng.integers(0, 20))
    f = CashFunctor(price=price, fixOrderCost=fix, variCost=vari, holdingCost=float(rng.choice([0.0, money(0, 1)])),
                    depositeRate=0.0, overheadRate=0.0, penaltyCost=0.0, salvageValue=money(0, 1),
                    discountFactor=float(rng.choice([1.0, 0.9375])), cashFormula=int(rng.integers(0, 2)),
                    overheadCosts=[money(0, 3) for _ in range(T)], cashRoundMult=q, cashRoundDiv=q,
                    cashRoundIntDiv=bool(q == 1.0 and rng.integers(0, 2)), maxOrderQuantity=float(rng.integers(3, 40)),
                    minInventoryState=0.0, maxInventoryState=float(rng.integers(3, 25)), minCashState=min_cash,
                    maxCashState=min_cash + (nc - 1) / q, iniInventory=0.0, iniCash=float(rng.integers(0, 15)))
    direction = OptDirection.MAX if rng.integers(0, 4) else OptDirection.MIN
    # (one instance in three has a demand support with gaps: the diagonal kernel pairs action k + i with demand j + i and must
    # leave such periods to the uniform-shift kernel -- round 2 did not, and no test had a gapped support on a wide dyadic row)
    return Workload(f"fuzz_wide_cash_{seed}", f, direction,
                    _pmf(rng, T, unit_stride=bool(rng.integers(0, 3)), d_max=int(rng.integers(3, 40))))


def test_random_wide_cash_rows_bit_exact(sia, oracle, monkeypatch):
    monkeypatch.setenv("SDPGPU_CASH_DIAG_CHECK", "1")  # (the guard word behind the diagonal kernel's spread bound)
    n = int(os.environ.get("SDP_FUZZ_N", "24"))
    for seed in range(n):
        w = make_wide_cash_instance(seed)
        V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=8)
        with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
            eng.solve()
            assert eng.stats().cells_evaluated == cells and eng.stats().kernel_used == 2, w.name
            for period in range(1, w.T + 1):
                assert np.array_equal(eng.policy(period), pol[period - 1]), f"{w.name} t={period}: policy"
                assert np.array_equal(eng.values(period), V[period - 1]), f"{w.name} t={period}: values"


# ---------------------------------------------------------------------------------------------------------------
# Round 4: F5 (cash + lead time) at large balances on the two-point kernel -- the two-addition rounding of the quantiser and the
# uniform-key trips of tiles that pay no interest, at magnitudes where one ulp of the balance is 2e-10
# ---------------------------------------------------------------------------------------------------------------
def make_large_magnitude_f5_instance(seed):
    """SingleProductLeadtime's shape with cash in hundredths on rows of 1500-4000 points whose balances sit around +-1e6:
    positive rows pay no interest (uniform-key trips when r0 = 0), negative rows pay the piecewise interest beyond the limit
    (the quantiser on every cell), rows across zero mix the two inside one launch.  bound * mult stays below the 5e8 the
    launcher admits (the bound multiplies the balance by 1 + r3)."""
    rng = np.random.default_rng(7300 + seed)
    T = int(rng.integers(2, 4))
    mult = 100.0
    nc = int(rng.integers(1500, 4000))
    where = seed % 3  # 0: positive balances, 1: negative, 2: across zero
    mag = float(rng.uniform(0.5, 0.9)) * 5.0e8 / mult / 3.05
    base = {0: mag, 1: -mag - nc / mult, 2: -0.5 * nc / mult}[where]
    base = float(np.floor(base * mult) / mult)
    on_grid = bool(rng.integers(0, 3))
    money = (lambda lo, hi: float(round(rng.uniform(lo, hi) * mult) / mult)) if on_grid else \
            (lambda lo, hi: float(round(rng.uniform(lo, hi), 3) + 0.0005))
    f = CashLeadtimeFunctor(price=money(0.3, 2.0), variCost=max(0.01, money(0.05, 0.6)), salvageValue=money(0, 0.2),
                            maxOrderQuantity=float(rng.integers(2, 6)), minInventoryState=0.0,
                            maxInventoryState=float(rng.integers(2, 7)), minCashState=base, maxCashState=base + (nc - 1) / mult,
                            iniInventory=0.0, iniCash=base + 3.0, cashRoundMult=mult, cashRoundDiv=mult, cashRoundIntDiv=False,
                            r0=float(rng.choice([0, 0, 0.01])), r2=0.1, r3=2.0, limit=float(rng.integers(5, 20)),
                            interestFreeAmount=float(rng.integers(0, 4)), iniPreQ=0.0,
                            overheadCosts=[money(0, 1.0) for _ in range(T)])
    return Workload(f"fuzz_big_f5_{seed}", f, OptDirection.MAX, _pmf(rng, T, d_max=8))


def test_large_magnitude_f5_bit_exact(sia, oracle, monkeypatch):
    """Automatic kernel (two-point kernel, four rows of a level per workgroup, diagonal order), the one-point row kernel and the
    generic kernel against the oracle, every table bit for bit."""
    n = int(os.environ.get("SDP_FUZZ_N", "9"))
    for seed in range(n):
        w = make_large_magnitude_f5_instance(seed)
        V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=8)
        for variant in ("auto", "one-point", "generic"):
            monkeypatch.delenv("SDPGPU_CASH_OD_PAIR", raising=False)
            if variant == "one-point":
                monkeypatch.setenv("SDPGPU_CASH_OD_PAIR", "0")
            d = w.desc()
            d.kernel = 1 if variant == "generic" else 0
            with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
                eng.solve()
                assert eng.stats().kernel_used == (1 if variant == "generic" else 2), f"{w.name} {variant}"
                assert eng.stats().cells_evaluated == cells, w.name
                for period in range(1, w.T + 1):
                    assert np.array_equal(eng.policy(period), pol[period - 1]), f"{w.name} {variant} t={period}: policy"
                    assert np.array_equal(eng.values(period), V[period - 1]), f"{w.name} {variant} t={period}: values"


# ---------------------------------------------------------------------------------------------------------------
# Large magnitudes: the proof-carrying shortcuts of the cash kernels near their admission limit
# ---------------------------------------------------------------------------------------------------------------
def make_large_magnitude_cash_instance(seed, past_limit=False):
    """F3 / F4 on wide cash rows whose BALANCES are large: cash in tenths or hundredths around 2.5e7 .. 4.9e7 (tenths) or
    2.5e6 .. 4.9e6 (hundredths), i.e. bound * mult within a factor two of the 5e8 up to which cash_row_eligible admits the
    integer-domain clamp, the uniform-key trips (|mult * inc - delta| < 2^-20 with the balance cancelled) and the LEAN
    elisions -- the magnitudes at which the chain-error argument behind them (the reference's rounded floating-point chain
    cash + inc, times mult, stays within +-0.5 of key + delta) has the least room: one ulp of the balance is 7.5e-9 there,
    against 1.4e-14 at the toy magnitudes of make_instance.  Decimal prices on and off the cash grid.
    past_limit: the same shapes with bound * mult beyond 5e8 -- the launcher must fall back to the generic kernel."""
    rng = np.random.default_rng(991000 + seed)
    T = int(rng.integers(2, 4))
    hundredths = bool(rng.integers(0, 2))
    mult = 100.0 if hundredths else 10.0
    unit = 1.0 / mult
    on_grid = bool(rng.integers(0, 4))            # three in four: prices and costs on the cash grid (uniform-key trips)
    money = (lambda lo, hi: float(round(rng.uniform(lo, hi) / unit) * unit)) if on_grid else \
            (lambda lo, hi: float(round(rng.uniform(lo, hi), 3) + 0.0005))
    scale = 5.0e8 / mult                          # the balance at which bound * mult reaches the limit
    base = float(rng.uniform(1.06, 1.6) if past_limit else rng.uniform(0.5, 0.93)) * scale
    # rows wide enough, and prices small enough, for most successors to land INSIDE the row (not on its clamped ends): a unit
    # sold moves the balance by 5 .. 60 keys, the row has 1500 .. 5000 of them
    nc = int(rng.integers(1500, 5000))
    family4 = bool(rng.integers(0, 4) == 0)
    if family4:
        base /= 3.02                              # (the launcher's bound multiplies the balance by 1 + the largest rate, r3 = 2)
    base = float(np.floor(base * mult) / mult)
    pscale = 60.0 / mult
    common = dict(price=money(0.1 * pscale, pscale), variCost=max(unit, money(0.03 * pscale, 0.3 * pscale)),
                  salvageValue=money(0, 0.1 * pscale),
                  maxOrderQuantity=float(rng.integers(3, 16)), minInventoryState=0.0, maxInventoryState=float(rng.integers(2, 9)),
                  minCashState=base, maxCashState=base + (nc - 1) / mult, iniInventory=0.0, iniCash=base + 5.0,
                  cashRoundMult=mult, cashRoundDiv=mult, cashRoundIntDiv=False)
    overheads = [money(0, 0.5 * pscale) for _ in range(T)]
    if family4:
        f = OverdraftFunctor(fixOrderCost=money(0, 0.5 * pscale), r0=float(rng.choice([0, 0.01])), r2=0.1, r3=2.0,
                             limit=float(rng.integers(10, 40)), interestFreeAmount=float(rng.integers(0, 10)),
                             discountFactor=float(rng.choice([1.0, 0.9])), overheadCosts=overheads, **common)
        return Workload(f"fuzz_big_f4_{seed}", f, OptDirection.MAX, _pmf(rng, T, d_max=12))
    penalty = float(rng.choice([0, 0, 0, 0.3]))
    if penalty:  # (the launcher's bound multiplies by 1 + the end-cash penalty rate)
        base = float(np.floor(base / 1.31 * mult) / mult)
        common.update(minCashState=base, maxCashState=base + (nc - 1) / mult, iniCash=base + 5.0)
    f = CashFunctor(fixOrderCost=float(rng.choice([0.0, money(0, 0.5 * pscale)])),
                    holdingCost=float(rng.choice([0.0, money(0, 0.1 * pscale)])),
                    depositeRate=float(rng.choice([0, 0, 0, 1e-9])), overheadRate=0.0, penaltyCost=penalty,
                    discountFactor=float(rng.choice([1.0, 0.95])), cashFormula=int(rng.integers(0, 2)),
                    overheadCosts=[float(rng.choice([0.0, o])) for o in overheads], **common)
    direction = OptDirection.MAX if rng.integers(0, 4) else OptDirection.MIN
    return Workload(f"fuzz_big_f3_{seed}", f, direction, _pmf(rng, T, unit_stride=bool(rng.integers(0, 3)), d_max=14))


@pytest.mark.parametrize("past_limit", [False, True], ids=["within-2x-of-the-limit", "past-the-limit"])
def test_large_magnitude_cash_shortcuts_bit_exact(sia, oracle, monkeypatch, past_limit):
    """Three kernels against the oracle at balances of 2.5e7 .. 8e7 (tenths) / 2.5e6 .. 8e6 (hundredths): the automatically
    selected one (the cash row kernels with every shortcut, while bound * mult < 5e8 -- the generic kernel beyond), the
    cash row kernel with pairs and uniform-key trips switched off (the per-point quantiser path), and the generic kernel."""
    n = int(os.environ.get("SDP_FUZZ_N", "14"))
    seen = set()
    for seed in range(n):
        w = make_large_magnitude_cash_instance(seed, past_limit)
        V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=8)
        for variant in ("auto", "no-shortcuts", "generic"):
            for k in ("SDPGPU_CASH_PAIR", "SDPGPU_CASH_UNI"):
                monkeypatch.delenv(k, raising=False)
            if variant == "no-shortcuts":
                monkeypatch.setenv("SDPGPU_CASH_PAIR", "0")
                monkeypatch.setenv("SDPGPU_CASH_UNI", "0")
            d = w.desc()
            d.kernel = 1 if variant == "generic" else 0
            with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
                eng.solve()
                used = eng.stats().kernel_used
                if variant != "generic":
                    assert used == (1 if past_limit else 2), f"{w.name} {variant}: kernel {used}"
                    seen.add(used)
                assert eng.stats().cells_evaluated == cells, w.name
                for period in range(1, w.T + 1):
                    assert np.array_equal(eng.policy(period), pol[period - 1]), f"{w.name} {variant} t={period}: policy"
                    assert np.array_equal(eng.values(period), V[period - 1]), f"{w.name} {variant} t={period}: values"
    assert seen == ({1} if past_limit else {2})


def make_stepped_instance(family, seed, step):
    """make_instance(family, seed) on a COARSER inventory grid: stepSize = 2 or 4 (the ABI takes any power-of-two integer; every
    in-scope driver, and until round 3 every test, uses 1).  Inventory bounds, initial inventory and demands are multiplied by
    the step, so every level stays a multiple of it; costs and the cash axis are as they were.  The order bound is multiplied
    too for F1 / F2, whose action list has (int)(maxQ / step) + 1 entries; the cash families' lists have (int)maxQ + 1 entries
    WHATEVER the step (`limit((int) maxQ + 1)`, CashConstraint.java:99 ...), so there the bound stays and the largest order is
    maxQ * step."""
    w = make_instance(family, seed)
    f = w.functor
    f.stepSize = float(step)
    names = ["minInventory", "maxInventory", "minInventoryState", "maxInventoryState", "iniInventory", "iniPreQ", "iniPreQ2"]
    if family in (1, 2):
        names.append("maxOrderQuantity")
    for name in names:
        if hasattr(f, name) and getattr(f, name) is not None:
            setattr(f, name, float(getattr(f, name)) * step)
    w.pmf = [np.stack([t[:, 0] * step, t[:, 1]], axis=1) for t in w.pmf]
    w.name = f"{w.name}_step{step}"
    return w


@pytest.mark.parametrize("step", [2, 4])
@pytest.mark.parametrize("family", [1, 2, 3, 4, 6])
def test_random_instances_on_coarser_inventory_grids(sia, oracle, family, step):
    kernels_seen = set()
    for seed in range(24):
        w = make_stepped_instance(family, 300 + seed, step)
        V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=4)
        for kernel in (0, 1):
            d = w.desc()
            d.kernel = kernel
            with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
                eng.solve()
                kernels_seen.add(eng.stats().kernel_used)
                assert eng.stats().cells_evaluated == cells, w.name
                for period in range(1, w.T + 1):
                    assert np.array_equal(eng.policy(period), pol[period - 1]), f"{w.name} kernel {kernel} t={period}: policy"
                    assert np.array_equal(eng.values(period), V[period - 1]), f"{w.name} kernel {kernel} t={period}: values"
    assert len(kernels_seen) >= 2, kernels_seen  # (the family's fast kernel took some of the instances, the gather kernel the rest)


def test_pipeline_cash_family_refuses_a_coarser_grid(sia):
    """F5's orders are k * step for k <= (int)maxQ (SingleProductLeadtime.java:76) and become the next state's preQ: with step != 1
    they leave the pipeline axis of (int)(maxQ / step) + 1 planes.  No in-scope driver does that; the engine says so at create."""
    w = make_stepped_instance(5, 300, 2)
    with pytest.raises(sia.SdpgpuError) as e:
        sia.SdpEngine(w.desc(), w.pmf, w.overhead())
    assert e.value.code == 4 and "step must be 1" in e.value.message  # SDPGPU_ERR_UNSUPPORTED


# ---------------------------------------------------------------------------------------------------------------
# Degenerate and boundary SHAPES: the axis lengths and pmf widths at which tiles, waves and LDS segments are ragged or empty
# ---------------------------------------------------------------------------------------------------------------
def _set(f, **kw):
    for name, v in kw.items():
        if hasattr(f, name):
            setattr(f, name, v)


def _one_inventory_level(w, rng):
    f = w.functor
    _set(f, clampInventory=True)
    for lo, hi, ini in (("minInventory", "maxInventory", "iniInventory"), ("minInventoryState", "maxInventoryState", "iniInventory")):
        if hasattr(f, lo):
            v = float(rng.integers(0, 3))
            _set(f, **{lo: v, hi: v, ini: v})


def _one_cash_point(w, rng):
    f = w.functor
    if hasattr(f, "minCashState"):
        c = float(rng.integers(0, 20))
        _set(f, minCashState=c, maxCashState=c, iniCash=c)
    else:  # (no cash axis: a two-level inventory grid instead)
        _set(f, clampInventory=True, minInventory=0.0, maxInventory=1.0, iniInventory=0.0)


def _no_orders(w, rng):
    _set(w.functor, maxOrderQuantity=0.0)


def _one_demand(w, rng):
    w.pmf = [np.array([[float(rng.integers(0, 6)), 1.0]]) for _ in w.pmf]


def _negative_demands(w, rng):
    w.pmf = [np.stack([t[:, 0] - float(rng.integers(1, 8)), t[:, 1]], axis=1) for t in w.pmf]


def _demands_beyond_grid(w, rng):
    w.pmf = [np.stack([t[:, 0] * float(rng.integers(20, 60)), t[:, 1]], axis=1) for t in w.pmf]


def _zero_probabilities(w, rng):
    out = []
    for t in w.pmf:
        p = t[:, 1].copy()
        p[rng.random(len(p)) < 0.4] = 0.0
        p[0] = 0.0
        p[-1] = 0.0
        out.append(np.stack([t[:, 0], p], axis=1))
    w.pmf = out


def _wide_pmf(n):
    def mutate(w, rng):
        p = rng.random(n) + 0.01
        w.pmf = [np.stack([np.arange(n, dtype=np.float64) + float(rng.integers(0, 2)), p / p.sum()], axis=1) for _ in w.pmf]
        if hasattr(w.functor, "maxCashState"):  # (keep the cash families' cell counts modest)
            _set(w.functor, maxCashState=min(w.functor.maxCashState, 30.0))
    return mutate


def _one_period(w, rng):
    w.pmf = w.pmf[:1]
    f = w.functor
    if getattr(f, "overheadCosts", None) is not None:
        f.overheadCosts = list(f.overheadCosts)[:1]


SHAPES = {"one_inventory_level": _one_inventory_level, "one_cash_point": _one_cash_point, "no_orders": _no_orders, "one_demand": _one_demand,
          "negative_demands": _negative_demands, "demands_beyond_grid": _demands_beyond_grid, "zero_probabilities": _zero_probabilities,
          "one_period": _one_period, **{f"pmf_{n}_points": _wide_pmf(n) for n in (63, 64, 65, 127, 128, 129, 255, 256, 257)}}


def make_shaped_instance(family, seed, shape):
    w = make_instance(family, seed)
    SHAPES[shape](w, np.random.default_rng(5150 + 100 * family + seed))
    w.name = f"{w.name}_{shape}"
    return w


@pytest.mark.parametrize("shape", list(SHAPES))
def test_degenerate_and_boundary_shapes_bit_exact(sia, oracle, shape):
    """Every family on grids with ONE inventory level, ONE cash point, no orders, one demand point, negative demands, demands
    far beyond the grid (every successor clamps), zero probabilities at both ends of the support, one period, and pmf widths
    around the wave (64), the window kernels' LDS segments (128) and the workgroup (256): automatically selected kernel and the
    generic kernel against the oracle."""
    wide = shape.startswith("pmf_")
    for family in (1, 2, 3, 4, 5, 6):
        for seed in range(3 if wide else 6):
            w = make_shaped_instance(family, 40 + seed, shape)
            V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=8)
            for kernel in (0, 1):
                d = w.desc()
                d.kernel = kernel
                with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
                    eng.solve()
                    assert eng.stats().cells_evaluated == cells, w.name
                    for period in range(1, w.T + 1):
                        assert np.array_equal(eng.policy(period), pol[period - 1]), f"{w.name} kernel {kernel} t={period}: policy"
                        assert np.array_equal(eng.values(period), V[period - 1]), f"{w.name} kernel {kernel} t={period}: values"


# ---------------------------------------------------------------------------------------------------------------
# The read-out side on the same random instances: reachable set, device rollout, off-grid evaluation
# ---------------------------------------------------------------------------------------------------------------
def _readout_instances(family):
    for seed in range(8):
        yield make_instance(family, 500 + seed)
    if family != 5:
        for seed in range(4):
            yield make_stepped_instance(family, 520 + seed, 2)
    for shape in ("one_inventory_level", "no_orders", "negative_demands", "demands_beyond_grid", "zero_probabilities", "pmf_65_points"):
        yield make_shaped_instance(family, 540, shape)


@pytest.mark.parametrize("family", [1, 2, 3, 4, 5, 6])
def test_random_instances_read_out_bit_exact(sia, oracle, family):
    """getOptTable's reachable-set filter (Recursion.java:177-186), the simulators' table-lookup rollout (Simulation.java:53-107) and
    the evaluation of states that are not grid points (what getExpectedValue answers for an arbitrary State) on the random,
    coarser-grid and degenerate instances of this file: sdpgpu_reachable / sdpgpu_simulate / sdpgpu_eval_states against the
    oracle's literal loops, bit for bit."""
    import math
    import zlib
    for w in _readout_instances(family):
        f = w.functor
        rng = np.random.default_rng(zlib.crc32(w.name.encode()))
        P = oracle.Problem(w.desc(), w.pmf, w.overhead())
        V, pol, _ = P.solve(nthreads=4)
        with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
            eng.solve()
            reach = P.reachable()
            for period in range(1, w.T + 1):
                assert np.array_equal(eng.reachable(period), reach[period - 1]), f"{w.name} t={period}: reachable"
            # rollout: demands drawn from each period's support (zero-probability points included: the table still answers)
            n = 64
            dem = np.stack([rng.choice(np.asarray(w.pmf[t])[:, 0], size=n) for t in range(w.T)], axis=1)
            gamma = getattr(f, "discountFactor", 1.0) if w.desc().family in (3, 4) else 1.0
            disc = np.array([math.pow(gamma, t) for t in range(w.T)])
            ini = (getattr(f, "iniInventory", 0.0), getattr(f, "iniCash", 0.0) or 0.0, getattr(f, "iniPreQ", 0.0) or 0.0)
            gs, gv = eng.simulate(dem, disc, *ini)
            os_, ov = P.simulate(V, pol, dem, disc, *ini)
            assert np.array_equal(gv, ov), f"{w.name}: rollout validity"
            assert np.array_equal(gs[gv.astype(bool)], os_[ov.astype(bool)]), f"{w.name}: rollout sums"
            # states between grid points (and on them) of a middle period
            period = max(1, w.T - 1)
            x, cash, preq = P.state_arrays(period)
            pick = rng.choice(len(x), size=min(40, len(x)), replace=False)
            step = float(f.stepSize) if hasattr(f, "stepSize") and f.stepSize else 1.0
            qx = x[pick]
            qc = cash[pick] + (rng.integers(0, 3, size=len(pick)) * 0.013 if family in (3, 4, 5, 6) else 0.0)
            qp = preq[pick]
            v_next = V[period] if period < w.T else None
            kw = dict(cash=qc if family in (3, 4, 5, 6) else None, preq=qp if family in (2, 5) else None)
            if family == 2 and w.desc().lead_time == 2:
                kw["preq2"] = P.preq2_array(period)[pick]
            ov_, oa_ = P.eval_states(period, v_next, qx, **kw)
            gv_, ga_ = eng.eval_states(period, qx, **kw)
            assert np.array_equal(ga_, oa_) and np.array_equal(gv_, ov_), f"{w.name}: eval_states t={period}"


# ---------------------------------------------------------------------------------------------------------------
# sdp.cash.CashRecursionXR (family CASH, cash_formula 2): state (x, R), order-up-to actions bounded by R / variCost
# ---------------------------------------------------------------------------------------------------------------
def make_xr_instance(seed):
    from stochastic_inventory_amd.functors import CashXRFunctor
    rng = np.random.default_rng(424200 + seed)
    T = int(rng.integers(1, 5))
    kind = ["int", "dyadic", "decimal"][int(rng.integers(0, 3))]
    vari = max(0.25, _money(rng, 0.5, 3, kind))
    max_inv = float(rng.integers(2, 16))
    f = CashXRFunctor(price=_money(rng, 2, 9, kind), fixOrderCost=float(rng.choice([0.0, _money(rng, 0, 4, kind)])), variCost=vari,
                      holdingCost=float(rng.choice([0.0, _money(rng, 0, 1, kind)])), depositeRate=float(rng.choice([0, 0, 0.02])),
                      overheadCost=float(rng.choice([0.0, _money(rng, 0, 2, kind)])), overheadRate=float(rng.choice([0, 0, 0.05])),
                      salvageValue=_money(rng, 0, 1, kind), discountFactor=float(rng.choice([1.0, 0.96])),
                      maxOrderQuantity=float(rng.integers(5, 200)), minInventoryState=0, maxInventoryState=max_inv,
                      minCashState=-float(rng.integers(0, 12)), maxCashState=float(rng.integers(15, 70)),
                      iniInventory=float(rng.integers(0, 3)), iniCash=float(rng.integers(5, 30)), **_rounding(rng))
    return Workload(f"fuzz_xr_{seed}", f, OptDirection.MAX if rng.integers(0, 4) else OptDirection.MIN,
                    _pmf(rng, T, unit_stride=bool(rng.integers(0, 3)), d_max=int(rng.integers(2, 12))))


def test_random_xr_instances_bit_exact(sia, oracle):
    seen = set()
    for seed in range(int(os.environ.get("SDP_FUZZ_N", "40"))):
        w = make_xr_instance(seed)
        V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=4)
        for kernel in (0, 1):
            d = w.desc()
            d.kernel = kernel
            with sia.SdpEngine(d, w.pmf, w.overhead()) as eng:
                eng.solve()
                seen.add(eng.stats().kernel_used)
                assert eng.stats().cells_evaluated == cells, w.name
                for period in range(1, w.T + 1):
                    assert np.array_equal(eng.policy(period), pol[period - 1]), f"{w.name} kernel {kernel} t={period}: policy"
                    assert np.array_equal(eng.values(period), V[period - 1]), f"{w.name} kernel {kernel} t={period}: values"
    assert seen == {1, 2}


# ---------------------------------------------------------------------------------------------------------------
# The same random instances cut into slabs: N rank-handles on one device through sdpgpu_solve_multi
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family", [1, 2, 3, 4, 5, 6])
def test_random_instances_sharded_bit_exact(sia, oracle, family):
    """Slabs of 2, 3, 5 and 7 ranks (ragged against every grid: more ranks than states happens too) of the random, coarser-grid and
    degenerate instances, the slabs exchanged inside the library (value rows or key rows, whichever the period's plan makes
    travel): every rank's tables against the oracle's."""
    instances = [make_instance(family, 700 + s) for s in range(6)]
    if family != 5:
        instances += [make_stepped_instance(family, 720 + s, 2) for s in range(2)]
    instances += [make_shaped_instance(family, 730, shape) for shape in ("one_inventory_level", "one_demand", "pmf_129_points")]
    for i, w in enumerate(instances):
        world = (2, 3, 5, 7)[i % 4]
        V, pol, cells = oracle.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=4)
        engs = []
        try:
            for r in range(world):
                d = w.desc()
                d.rank, d.world_size, d.device = r, world, 0
                engs.append(sia.SdpEngine(d, w.pmf, w.overhead()))
            sia.SdpEngine.solve_multi(engs, sync=True, gather_first=True, threads=bool(i % 2))
            total = 0
            for r, e in enumerate(engs):
                total += int(e.stats().cells_evaluated)
                for period in range(1, w.T + 1):
                    _, lo, hi = e.slab(period)
                    assert np.array_equal(e.values(period), V[period - 1]), f"{w.name} rank {r}/{world}: V_{period}"
                    assert np.array_equal(e.policy(period), pol[period - 1][lo:hi]), f"{w.name} rank {r}/{world}: policy of period {period}"
            assert total == cells, w.name
        finally:
            for e in engs:
                e.close()

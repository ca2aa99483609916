"""The compiled-language host mirror (include/sdpgpu_mirror.hpp): driver programs in the shape of the reference's mains, written against the mirror,
to C++ (tests/cpp/mirror_drivers.cpp) run on the GPU and print what the CPU oracle computes."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpp") / "mirror_drivers"
    libdir = os.path.join(ROOT, "stochastic-inventory_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", str(out),
                    os.path.join(ROOT, "tests", "cpp", "mirror_drivers.cpp"), "-L", libdir, "-lsdpgpu",
                    f"-Wl,-rpath,{libdir}"], check=True)
    return str(out)


def _write_pmf(path, pmf):
    with open(path, "w") as f:
        f.write(f"{len(pmf)}\n")
        for tile in pmf:
            f.write(f"{len(tile)}\n")
            for d, p in tile:
                f.write(f"{float(d)!r} {float(p)!r}\n")


def _run(exe, which, pmf, tmp_path, *extra):
    path = tmp_path / f"{which}.pmf"
    _write_pmf(path, pmf)
    out = subprocess.run([exe, which, str(path), *extra], check=True, capture_output=True, text=True).stdout
    return out.strip().splitlines()


def _last_number(line):
    return float(line.rsplit(" ", 1)[1])


def test_clsp_testing_main(exe, tmp_path, sia, oracle):
    """CLSPTesting.main's first parameter set (K = 200, v = 1, pi = 10, h = 1; :46-56) on Poisson demands."""
    tiles = sia.GetPmf([sia.PoissonDist(m) for m in (10, 10, 10, 10, 10, 10, 10, 10)], 0.9999, 1).getpmf()
    lines = _run(exe, "clsp", tiles, tmp_path)
    f = sia.BackorderFunctor(fixedOrderingCost=200, variOrderingCost=1, holdingCost=1, penaltyCost=10,
                             minInventory=-500, maxInventory=500, maxOrderQuantity=500, iniInventory=0)
    m = oracle.Problem(f.to_desc(8), tiles).memo()
    assert _last_number(lines[0]) == m["value"]
    assert _last_number(lines[1]) == m["action"]
    parts = lines[2].split()
    assert int(parts[2]) == m["n"] and [float(v) for v in parts[4:7]] == [1.0, 0.0, m["action"]]
    # the simulated path: replay it with the oracle's tables
    lookup = {(int(p), float(x)): float(a) for p, x, a in zip(m["period"], m["x"], m["actions"])}
    x, total = 0.0, 0.0
    for t, tile in enumerate(tiles):
        q = lookup[(t + 1, x)]
        d = float(sia.java_round(float(tile[len(tile) // 2][0]) + 0.4))
        level = x + q - d
        total += (200 if q > 0 else 0) + 1 * q + 1 * max(level, 0.0) + 10 * max(-level, 0.0)
        x = min(500.0, max(-500.0, level))
    assert _last_number(lines[3]) == total


def test_leadtime_main(exe, tmp_path, sia, oracle):
    """Leadtime.main as it stands: meanDemand {10, 10, 10}, no inventory clamp, maxOrderQuantity 100."""
    tiles = sia.GetPmf([sia.PoissonDist(10.0)] * 3, 0.9999, 1).getpmf()
    lines = _run(exe, "leadtime", tiles, tmp_path)
    f = sia.LeadtimeFunctor(fixedOrderingCost=0, variOrderingCost=1, holdingCost=2, penaltyCost=10, maxOrderQuantity=100,
                            clampInventory=False, iniInventory=0, iniPreQ=0)
    m = oracle.Problem(f.to_desc(3), tiles).memo()
    assert _last_number(lines[0]) == m["value"]
    assert _last_number(lines[1]) == m["action"]
    assert int(lines[2].split()[-1]) == m["n"]


def test_cash_constraint_main(exe, tmp_path, sia, oracle):
    """CashConstraint.main's lambdas (cash in tenths, formula 0) on a smaller state box."""
    tiles = sia.GetPmf([sia.PoissonDist(10.0)] * 4, 0.9999, 1).getpmf()
    lines = _run(exe, "cash", tiles, tmp_path)
    f = sia.CashFunctor(price=10, fixOrderCost=0, variCost=1, holdingCost=0, salvageValue=0.5, maxOrderQuantity=100,
                        minInventoryState=0, maxInventoryState=120, minCashState=0, maxCashState=600, iniInventory=0,
                        iniCash=100)
    P = oracle.Problem(f.to_desc(4, sia.OptDirection.MAX), tiles)
    V, pol, _ = P.solve(nthreads=8)
    i0 = (0 * 6001) + 1000  # x = 0, cash = 100.0 -> key 1000
    assert _last_number(lines[0]) == V[0][i0]
    assert _last_number(lines[1]) == pol[0][i0]
    q, d = float(pol[0][i0]), float(tiles[0][len(tiles[0]) // 2][0])
    s1 = f.stateTransition(sia.CashState(1, 0.0, 100.0), q, d, 4)
    idx = int(s1.getIniInventory()) * 6001 + int(round(s1.getIniCash() * 10))
    assert _last_number(lines[2]) == V[1][idx]


def test_cash_survival_main(exe, tmp_path, sia, oracle):
    """cashSurvival.main's lambdas and parameters (mean demands {14, 23, 33, 46, 50}, overhead 100, price 4) on a
    smaller state box and with initial cash 150 instead of 80 (80 goes bankrupt in period 1 with probability
    0.995): survival probability, first order, number of visited states."""
    tiles = sia.GetPmf([sia.PoissonDist(m) for m in (14, 23, 33, 46, 50)], 0.99, 1).getpmf()
    lines = _run(exe, "survival", tiles, tmp_path)
    f = sia.SurvivalFunctor(price=4, fixOrderCost=0, variCost=1, holdingCost=0, depositeRate=0, overheadCost=100,
                            salvageValue=0.5, maxOrderQuantity=200, minInventoryState=0, maxInventoryState=200,
                            minCashState=-100, maxCashState=1500, iniInventory=0, iniCash=150)
    m = oracle.Problem(f.to_desc(5, sia.OptDirection.MAX), tiles).memo(cap=1 << 23)
    assert 0.0 < m["value"] < 1.0
    assert _last_number(lines[0]) == m["value"]
    assert _last_number(lines[1]) == m["action"]
    assert int(lines[2].split()[-1]) == m["n"]


def test_overdraft_limit_main_with_user_lambdas(exe, tmp_path, sia, oracle):
    """CashOverdraftLimit.main's lambdas are not a built-in family: the C++ driver passes them as HIP device text
    (functor.user.source) beside its own C++ lambdas; the oracle runs the same text compiled for the host."""
    import cases
    import custom_sources as cs
    from test_gpu_custom_functor import _overdraft_limit_case
    shape, params, pmf = _overdraft_limit_case(sia)
    src = tmp_path / "limit.hip"
    src.write_text(cs.OVERDRAFT_LIMIT)
    lines = _run(exe, "limit", pmf, tmp_path, str(src))
    P = oracle.Problem(shape.to_desc(4, sia.OptDirection.MAX), pmf)
    with oracle.custom_functor(cs.OVERDRAFT_LIMIT, params):
        V, pol, _ = P.solve(nthreads=4)
        m = P.memo()
    assert _last_number(lines[0]) == 10 + m["value"]
    assert _last_number(lines[1]) == m["action"]
    by = {(int(p), x, c): v for p, x, c, v in zip(m["period"], m["x"], m["cash"], m["values"])}
    assert _last_number(lines[2]) in [v for (p, x, c), v in by.items() if p == 2]


def test_workforce_planning_main(exe, tmp_path, sia):
    """WorkforcePlanning.main's lambdas on a smaller staff range (0..60, hires 0..40, three periods, turnover 0.5):
    optimal cost, first hire, visited states and one successor's value against oracle/staffref.c."""
    from oracle import staffref
    from stochastic_inventory_amd.pmf import staff_level_pmf
    T, rows = 3, 61
    table = staff_level_pmf([0.5] * T, rows)
    path = tmp_path / "workforce.tbl"
    with open(path, "w") as f:
        f.write(f"{T} {rows}\n")
        for t in range(T):
            for i in range(rows):
                f.write(" ".join(repr(float(p)) for p in table[t, i, : i + 1]) + "\n")
    lines = subprocess.run([exe, "workforce", str(path)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    P = staffref.Problem(T=T, min_x=0, max_x=rows - 1, clamp=True, ini_x=0, max_hire=40, fix_cost=100, unit_vari_cost=10,
                         salary=20, unit_penalty=80, min_staff=[12] * T, prob=table)
    V, pol, _ = P.solve()
    root, act, val, acts, seen, _ = P.memo(rows)
    assert _last_number(lines[0]) == V[0][0] == root
    assert int(lines[1].split()[-1]) == pol[0][0] == act
    assert int(lines[2].split()[-1]) == int(seen.sum())
    assert _last_number(lines[3]) == V[1][pol[0][0] - 3]


def test_multi_item_cash_xr_main(exe, tmp_path, sia, oracle):
    """MultiItemCashXR.main's parameters with Qbound 20 and demand supports cut at the 0.9 quantile, through the C++
    mirror class: final cash, order-up-to levels and the number of visited states against the oracle's recursion."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multicash_cases
    kw = multicash_cases.xr_main_instance(q_bound=20, q=0.9)
    path = tmp_path / "multixr.pmf"
    with open(path, "w") as f:
        f.write(f"{kw['T']}\n")
        for tile in kw["pmf"]:
            f.write(f"{len(tile)}\n")
            for d1, d2, p in tile:
                f.write(f"{float(d1)!r} {float(d2)!r} {float(p)!r}\n")
    lines = subprocess.run([exe, "multixr", str(path)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    fv, y1, y2, states, _ = oracle.multixr_memo(0.0, **kw)
    assert _last_number(lines[0]) == fv
    assert lines[1].endswith(f"y1 = {y1}, y2 = {y2}")
    assert int(lines[2].split()[-1]) == sum(states)

"""GPU: CashRecursionMulti over MultiItemCash's lambdas on the reachable-set engine (sdpgpu_multicash_solve) against
the oracle's literal recursion: random instances, and MultiItemCash.main at the size it is written for."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import multicash_cases  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(24))
def test_random_instances_match_the_oracle(sia, oracle, seed):
    kw = multicash_cases.random_instance(seed)
    r = sia.multicash_solve(**kw)
    fv, q1, q2, states, cells = oracle.multicash_memo(**kw)
    assert r.finalValue == fv and (r.firstAction, r.secondAction) == (q1, q2)
    assert r.statesPerPeriod == states and r.cells == cells


def test_multi_item_cash_main_smaller_action_box(sia, oracle):
    """MultiItemCash.main's parameters with Qbound 16 instead of 100 (the oracle's single-threaded recursion needs two
    minutes for the full box)."""
    kw = multicash_cases.main_instance()
    kw["q_bound"] = 16
    r = sia.multicash_solve(**kw)
    fv, q1, q2, states, cells = oracle.multicash_memo(**kw)
    assert r.finalValue == fv and (r.firstAction, r.secondAction) == (q1, q2)
    assert r.statesPerPeriod == states and r.cells == cells


def test_multi_item_cash_main(sia):
    """MultiItemCash.main (its solve is commented out in the reference; parameters as they stand): T = 2, Qbound 100,
    iniCash 100 -- one period-1 state, 24673 successors, 9.1e9 cells -- against the oracle's result committed in
    tests/golden/multicash_main.json (tests/golden/make_golden.py)."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "multicash_main.json")))
    kw = multicash_cases.main_instance()
    for t in range(kw["T"]):  # the instance itself (scipy's normal cdf) has not drifted
        assert kw["pmf"][t].tolist() == g["pmf"][t]
    r = sia.multicash_solve(**kw)
    print(f"MultiItemCash.main: final optimal cash {r.finalValue!r}, Q1 = {r.firstAction}, Q2 = {r.secondAction}, "
          f"states {r.statesPerPeriod}, {r.cells:.3g} cells in {r.gpu_ms:.1f} ms")
    assert r.finalValue == g["final_value"] and (r.firstAction, r.secondAction) == (g["q1"], g["q2"])
    assert r.statesPerPeriod == g["states_per_period"] and r.cells == g["cells"]


def test_misuse(sia):
    kw = multicash_cases.random_instance(0)
    kw["min_cash"] = -5
    with pytest.raises(sia.SdpgpuError):
        sia.multicash_solve(**kw)


@pytest.mark.parametrize("seed", range(24))
def test_xr_random_instances_match_the_oracle(sia, oracle, seed):
    dep, kw = multicash_cases.xr_random_instance(seed)
    r = sia.multixr_solve(dep, **kw)
    fv, y1, y2, states, cells = oracle.multixr_memo(dep, **kw)
    assert r.finalValue == fv and (r.firstAction, r.secondAction) == (y1, y2)
    assert r.statesPerPeriod == states and r.cells == cells


def test_multi_item_cash_xr_main_smaller(sia, oracle):
    """MultiItemCashXR.main's parameters on a smaller box (Qbound 30, demand supports cut at the 0.9 quantile: 8.9e8
    cells; the full instance is 1.8e11 cells, half an hour for the oracle's single-threaded recursion)."""
    kw = multicash_cases.xr_main_instance(q_bound=30, q=0.9)
    r = sia.multixr_solve(0.0, **kw)
    fv, y1, y2, states, cells = oracle.multixr_memo(0.0, **kw)
    assert r.finalValue == fv and (r.firstAction, r.secondAction) == (y1, y2)
    assert r.statesPerPeriod == states and r.cells == cells


def test_multi_item_cash_xr_main(sia):
    """MultiItemCashXR.main as it stands (its solve is live in the reference): T = 2, Qbound 50, 352 demand pairs,
    71767 period-2 states, 6.3e10 cells -- against the oracle's result committed in tests/golden/multixr_main.json
    (six minutes on eight threads, tests/golden/make_golden.py)."""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "multixr_main.json")))
    kw = multicash_cases.xr_main_instance()
    for t in range(kw["T"]):
        assert kw["pmf"][t].tolist() == g["pmf"][t]  # the instance itself (scipy's Poisson pmf) has not drifted
    r = sia.multixr_solve(0.0, **kw)
    print(f"MultiItemCashXR.main: final optimal cash {r.finalValue!r}, y1 = {r.firstAction}, y2 = {r.secondAction}, "
          f"states {r.statesPerPeriod}, {r.cells:.3g} cells in {r.gpu_ms:.1f} ms")
    assert r.finalValue == g["final_value"] and (r.firstAction, r.secondAction) == (g["y1"], g["y2"])
    assert r.statesPerPeriod == g["states_per_period"] and r.cells == g["cells"]


@pytest.mark.parametrize("kind,seed", [("multicash", s) for s in range(8)] + [("multixr", s) for s in range(8)])
def test_whole_memo_matches_the_oracle(sia, oracle, kind, seed):
    """Every visited state -- its tuple, V(state) and the chosen action pair, in the reference's key order (what
    getCacheActions / getOptTable iterate) -- not only the root."""
    if kind == "multicash":
        kw = multicash_cases.random_instance(seed)
        r = sia.multicash_solve(table=True, **kw)
        (_, _, _, states, _), want = oracle.memo_table("multicash", **kw)
    else:
        dep, kw = multicash_cases.xr_random_instance(seed)
        r = sia.multixr_solve(dep, table=True, **kw)
        (_, _, _, states, _), want = oracle.memo_table("multixr", dep, **kw)
    assert r.table.shape == want.shape == (sum(states), 9)
    assert (r.table == want).all()


@pytest.mark.parametrize("index_words", ["32-bit", "64-bit"])
@pytest.mark.parametrize("kind", ["multicash", "multixr"])
def test_lattice_path_whole_memo(sia, oracle, kind, index_words, monkeypatch):
    """The bitmap / rank path of the reachable-set engine (chosen by itself when states x actions x demand pairs
    outgrow 32-bit candidate indices), forced on small instances: every visited state, value and action pair -- with the
    successor's lattice index formed in 32-bit words (the default where the box allows it) and in 64-bit words."""
    monkeypatch.setenv("SDPGPU_MULTI_LATTICE", "1")
    if index_words == "64-bit":
        monkeypatch.setenv("SDPGPU_MULTI_I32", "0")
    done = 0
    for seed in range(24):
        if kind == "multicash":
            kw = multicash_cases.random_instance(seed)
            r = sia.multicash_solve(table=True, **kw)
            (fv, q1, q2, states, cells), want = oracle.memo_table("multicash", **kw)
        else:
            dep, kw = multicash_cases.xr_random_instance(seed)
            if any(c != int(c) for c in kw["vari_cost"]):
                kw["vari_cost"] = [float(int(c) + 1) for c in kw["vari_cost"]]  # R = cash + c . x must stay an integer
            r = sia.multixr_solve(dep, table=True, **kw)
            (fv, q1, q2, states, cells), want = oracle.memo_table("multixr", dep, **kw)
        assert r.finalValue == fv and (r.firstAction, r.secondAction) == (q1, q2), seed
        assert r.statesPerPeriod == states and r.cells == cells, seed
        assert r.table.shape == want.shape and (r.table == want).all(), seed
        done += 1
    assert done == 24


def test_xr_marking_through_post_order_triples(sia, oracle, monkeypatch):
    """MultiItemCashXR's forward pass with exact data (no deposit rate, integer unit costs and demands, prices on a 1/8 grid):
    the successors are marked from the distinct post-order triples (y1, y2, R - c . y) instead of from every (state, order, demand
    pair) -- chosen by itself on periods of 2e10 candidates and more, forced here on small instances: the same visited states,
    values and actions as the oracle's memoised recursion, and as the run with the form switched off."""
    monkeypatch.setenv("SDPGPU_MULTI_LATTICE", "1")
    done = 0
    for seed in range(40):
        _, kw = multicash_cases.xr_random_instance(seed)
        kw["vari_cost"] = [float(int(c) + 1) for c in kw["vari_cost"]]
        monkeypatch.setenv("SDPGPU_MULTI_TRIPLES", "1")
        r = sia.multixr_solve(0.0, table=True, **kw)
        (fv, q1, q2, states, cells), want = oracle.memo_table("multixr", 0.0, **kw)
        assert r.finalValue == fv and (r.firstAction, r.secondAction) == (q1, q2), seed
        assert r.statesPerPeriod == states and r.cells == cells, seed
        assert r.table.shape == want.shape and (r.table == want).all(), seed
        monkeypatch.setenv("SDPGPU_MULTI_TRIPLES", "0")
        r0 = sia.multixr_solve(0.0, table=True, **kw)
        assert (r0.table == r.table).all() and r0.statesPerPeriod == r.statesPerPeriod
        done += 1
    assert done == 40


def test_mirror_classes_read_like_the_reference_mains(sia, oracle):
    """MultiItemCash.main (:120-133) / MultiItemCashXR.main (:150-164) / MultiProductLeadtime.main (:225-239) through
    the mirror classes: construct, getExpectedValue(iniState), getAction(iniState), getOptTable(variCost)."""
    # --- CashRecursionMulti
    kw = multicash_cases.random_instance(5)
    functor = {k: v for k, v in kw.items() if k not in ("pmf", "ini_i1", "ini_i2", "ini_cash", "T", "discount")}
    recursion = sia.CashRecursionMulti(kw["discount"], kw["pmf"], None, None, None, kw["T"], functor=functor)
    iniState = sia.CashStateMulti(1, kw["ini_i1"], kw["ini_i2"], kw["ini_cash"])
    finalValue = kw["ini_cash"] + recursion.getExpectedValue(iniState)
    (fv, q1, q2, states, _), memo = oracle.memo_table("multicash", **kw)
    assert finalValue == fv
    assert (recursion.getAction(iniState).getFirstAction(), recursion.getAction(iniState).getSecondAction()) == (q1, q2)
    table = recursion.getOptTable(kw["vari_cost"])
    assert table.shape == (sum(states), 11)
    assert (table[:, [0, 1, 2, 3, 7, 8]] == memo[:, [0, 1, 2, 5, 7, 8]]).all()
    row = memo[len(memo) // 2]  # any visited state can be asked for afterwards
    s = sia.CashStateMulti(int(row[0]), row[1], row[2], row[5])
    assert recursion.getExpectedValue(s) == row[6]
    with pytest.raises(KeyError):
        recursion.getAction(sia.CashStateMulti(1, 77, 77, 77))
    # --- CashRecursionMultiXR
    dep, kw = multicash_cases.xr_random_instance(6)
    functor = {k: v for k, v in kw.items() if k not in ("pmf", "ini_i1", "ini_i2", "ini_cash", "T", "discount")}
    functor["depositeRate"] = dep
    recursion = sia.CashRecursionMultiXR(kw["discount"], kw["pmf"], None, None, None, kw["T"], functor=functor)
    iniState = sia.CashStateMultiXR(1, kw["ini_i1"], kw["ini_i2"], kw["ini_cash"])
    (fv, y1, y2, states, _), memo = oracle.memo_table("multixr", dep, **kw)
    assert kw["ini_cash"] + recursion.getExpectedValue(iniState) == fv
    assert recursion.getAction(iniState) == [float(y1), float(y2)]
    assert recursion.getOptTable(kw["vari_cost"]).shape == (sum(states), 11)
    # --- CashRecursionMultiLead (KAT-1)
    import json
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_reference.json")))["kat1"]
    functor = {n: k[n] for n in ("q_bound", "price", "vari_cost", "sal_value", "r0", "r1", "r2", "limit", "interest_free",
                                 "min_inventory", "max_inventory", "min_cash", "max_cash", "overhead", "values", "probs")}
    recursion = sia.CashRecursionMultiLead(k["discount"], None, None, None, None, k["T"], functor=functor)
    iniState = sia.CashStateMultiLead(1, k["ini_i1"], k["ini_i2"], 0, 0, k["ini_cash"])
    assert k["ini_cash"] + recursion.getExpectedValue(iniState) == k["expected_final_cash"]  # -17.800000000000008
    a = recursion.getAction(iniState)
    assert (a.getFirstAction(), a.getSecondAction()) == (k["expected_q1"], k["expected_q2"])
    assert len(recursion.getCacheActions()) == 2501


@pytest.mark.skipif(not os.environ.get("SDP_FUZZ_N"), reason="soak run: SDP_FUZZ_N=400 python -m pytest tests/test_gpu_multicash.py -m gpu -k soak")
def test_soak_whole_memo_both_paths(sia, oracle, monkeypatch):
    n = int(os.environ["SDP_FUZZ_N"])
    for seed in range(1000, 1000 + n):
        for path in ("0", "1"):
            monkeypatch.setenv("SDPGPU_MULTI_LATTICE", path)
            kw = multicash_cases.random_instance(seed)
            r = sia.multicash_solve(table=True, **kw)
            (fv, q1, q2, states, cells), want = oracle.memo_table("multicash", **kw)
            assert r.finalValue == fv and r.statesPerPeriod == states and r.cells == cells and (r.table == want).all(), (seed, path)
            dep, kw = multicash_cases.xr_random_instance(seed)
            if path == "1":
                kw["vari_cost"] = [float(int(c) + 1) for c in kw["vari_cost"]]
            r = sia.multixr_solve(dep, table=True, **kw)
            (fv, y1, y2, states, cells), want = oracle.memo_table("multixr", dep, **kw)
            assert r.finalValue == fv and r.statesPerPeriod == states and r.cells == cells and (r.table == want).all(), (seed, path)


def test_action_box_beyond_the_lds_is_refused_with_a_reason(sia):
    """Q(s, a) of every action pair of a state sits in LDS (8 B x Qbound^2): Qbound 200 would need 320 KB, over the 160 KiB of a
    compute unit -- SDPGPU_ERR_UNSUPPORTED with the arithmetic, not a launch error."""
    kw = multicash_cases.main_instance()
    kw["q_bound"] = 200
    with pytest.raises(sia.SdpgpuError) as ei:
        sia.multicash_solve(**kw)
    assert ei.value.code == 4 and "of LDS per state" in ei.value.message


@pytest.mark.parametrize("seed", [3, 7, 11])
@pytest.mark.parametrize("q_bound,T", [(1, None), (2, 1), (1, 1)], ids=["one-action", "one-period", "one-action-one-period"])
def test_degenerate_action_boxes_and_horizons(sia, oracle, seed, q_bound, T):
    """Qbound 1 (the only action pair is (0, 0)) and a horizon of one period, both two-product cash recursions."""
    kw = multicash_cases.random_instance(seed)
    dep, kx = multicash_cases.xr_random_instance(seed)
    for k in (kw, kx):
        k["q_bound"] = q_bound
        if T:
            k["T"] = T
            k["pmf"] = k["pmf"][:T]
    r = sia.multicash_solve(**kw)
    fv, q1, q2, states, cells = oracle.multicash_memo(**kw)
    assert r.finalValue == fv and (r.firstAction, r.secondAction) == (q1, q2)
    assert r.statesPerPeriod == states and r.cells == cells
    r = sia.multixr_solve(dep, **kx)
    fv, y1, y2, states, cells = oracle.multixr_memo(dep, **kx)
    assert r.finalValue == fv and (r.firstAction, r.secondAction) == (y1, y2)
    assert r.statesPerPeriod == states and r.cells == cells

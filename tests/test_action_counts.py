"""sdpgpu_set_action_counts: a `Function<State, double[]> getFeasibleAction` (Recursion.java:49,129) whose list is a prefix
of the action grid with a length of its own (here: a storage capacity, Q <= cap - x, and arbitrary tables incl. empty
lists).  CPU: the oracle's twin of the call -- dense sweep == literal memoised recursion, empty-list semantics
(Recursion.java:132-134), cell counts.  GPU: every table bit-identical to the oracle, cells counted alike, the specialised
kernels step aside, misuse is reported."""
import numpy as np
import pytest

import cases


def _capacity_counts(P, w, cap):
    """F1 with a storage capacity: orders 0, 1, ... while x + Q <= cap (at least the empty order)."""
    out = []
    full = int(w.functor.maxOrderQuantity / w.functor.stepSize) + 1
    for period in range(1, w.T + 1):
        x, _, _ = P.state_arrays(period)
        out.append(np.clip(np.floor(cap - x).astype(np.int64) + 1, 1, full).astype(np.int32))
    return out


def _random_counts(P, w, seed, allow_empty=True):
    rng = np.random.default_rng(seed)
    full = int(w.functor.maxOrderQuantity / w.functor.stepSize) + 1
    out = []
    for period in range(1, w.T + 1):
        c = rng.integers(0 if allow_empty else 1, full + 1, size=P.S[period - 1]).astype(np.int32)
        out.append(c if period != 2 else None)  # one period keeps the family's rule
    return out


@pytest.mark.parametrize("make", [cases.f1_small, cases.f3_tenths, cases.f2_clamped], ids=lambda f: f.__name__)
def test_oracle_counts_dense_equals_memo_and_empty_lists(oracle, make):
    w = make()
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    counts = _random_counts(P, w, 3)
    with P.action_counts(counts):
        V, pol, cells = P.solve()
        m = P.memo()
    # cells: the caller's lengths where given, the family's rule in the period that has none
    plain = oracle.Problem(w.desc(), w.pmf, w.overhead())
    want_cells = 0
    for t, c in enumerate(counts):
        if c is not None:
            want_cells += int(c.sum()) * len(w.pmf[t])
        else:
            want_cells += plain.period(t + 1, None if t + 1 == w.T else V[t + 1])[2]
    assert cells == want_cells
    big = 1.7976931348623157e308
    for t, c in enumerate(counts):
        if c is None:
            continue
        empty = c == 0
        want = big if w.direction.name == "MIN" else -big
        assert np.all(V[t][empty] == want) and np.all(pol[t][empty] == 0)   # Recursion.java:132-134
        assert np.all(pol[t] < np.maximum(c, 1))
    # literal memoised recursion from the initial state agrees with the dense tables on what it visited
    for i in range(m["n"]):
        period = int(m["period"][i])
        x, cash, preq = P.state_arrays(period)
        hit = np.nonzero((x == m["x"][i]) & (cash == m["cash"][i]) & (preq == m["preq"][i]))[0]
        if len(hit):
            assert V[period - 1][hit[0]] == m["values"][i]


@pytest.mark.gpu
@pytest.mark.parametrize("make,kind", [(cases.f1_small, "capacity"), (cases.f1_clsp_main, "capacity"), (cases.f1_small, "random"),
                                       (cases.f3_tenths, "random"), (cases.f3_dyadic, "random"), (cases.f2_clamped, "random"),
                                       (cases.f4_overdraft, "random"), (cases.f5_cash_leadtime, "random")],
                         ids=lambda v: getattr(v, "__name__", str(v)))
def test_gpu_matches_oracle_with_caller_counts(sia, oracle, make, kind):
    w = make()
    P = oracle.Problem(w.desc(), w.pmf, w.overhead())
    counts = _capacity_counts(P, w, 7.0) if kind == "capacity" else _random_counts(P, w, 11)
    with P.action_counts(counts):
        V, pol, cells = P.solve()
    with sia.SdpEngine(w.desc(), w.pmf, w.overhead()) as eng:
        for t, c in enumerate(counts):
            if c is not None:
                eng.set_action_counts(t, c)
        eng.solve()
        st = eng.stats()
        assert st.cells_evaluated == cells
        for period in range(1, w.T + 1):
            assert np.array_equal(eng.values(period), V[period - 1]), f"{w.name} V_{period}"
            assert np.array_equal(eng.policy(period), pol[period - 1]), f"{w.name} policy {period}"
    # sharded: the counts are indexed by the flat state index, so every rank is given the whole table
    engs = []
    try:
        for r in range(3):
            d = w.desc()
            d.rank, d.world_size, d.device = r, 3, 0
            e = sia.SdpEngine(d, w.pmf, w.overhead())
            for t, c in enumerate(counts):
                if c is not None:
                    e.set_action_counts(t, c)
            engs.append(e)
        sia.SdpEngine.solve_multi(engs)
        for e in engs:
            for period in range(2, w.T + 1):
                assert np.array_equal(e.values(period), V[period - 1])
    finally:
        for e in engs:
            e.close()


def test_misuse_is_reported(sia):
    w = cases.f1_small()
    with sia.SdpEngine(w.desc(), w.pmf) as eng:
        n = eng.num_states(1)
        full = int(w.functor.maxOrderQuantity) + 1
        with pytest.raises(sia.SdpgpuError):
            eng.set_action_counts(0, np.full(n, full + 1))      # more actions than the action grid has
        with pytest.raises(sia.SdpgpuError):
            eng.set_action_counts(0, np.full(n, -1))
        with pytest.raises(sia.SdpgpuError):
            eng.set_action_counts(w.T, np.ones(n))              # no such period
        eng.set_action_counts(0, np.ones(n + 5))                # wrong length: found when the grid is laid out ...
        with pytest.raises(sia.SdpgpuError):
            eng.solve()                                          # ... (here: no GPU, or the length check)

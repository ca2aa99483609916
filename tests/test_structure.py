"""Structural invariants of the single-item backorder recursion (protection (iv)): the check the
reference itself runs on its results -- K-convexity of G with a slack of 0.1, sdp.inventory.CheckKConvexity.check
(CheckKConvexity.java:39-68, called e.g. from WorkforcePlanning.java:208-209) -- and the (s, S) form of the optimal policy that
K-convexity implies (Scarf), on the oracle's tables of configs[0] (K = 500, v = 0, h = 2, pi = 10, Poisson(10) demand,
CLSP.java:207-210).  The reference records no outputs for this class, so these are the checks that do not depend on
a restatement being right: a wrong loop (accumulation, arg-min, transition, clamp) breaks them."""
import numpy as np
import pytest

from oracle import sdpref
from stochastic_inventory_amd import workloads


@pytest.fixture(scope="module")
def cfg1():
    w = workloads.cfg1_sS(T=12)
    V, pol, _ = sdpref.Problem(w.desc(), w.pmf, w.overhead()).solve(nthreads=4)
    return w, V, pol


def _G(w, V_next, period, ys):
    """G_t(y) = E[h (y - D)^+ + pi (D - y)^+ + V_{t+1}(clamp(y - D))]: the cost of standing at level y after ordering."""
    d = w.desc()
    dem, pr = w.pmf[period - 1][:, 0], w.pmf[period - 1][:, 1]
    out = []
    for y in ys:
        lev = y - dem
        nxt = np.clip(lev, d.min_inventory, d.max_inventory)
        out.append(float(np.sum(pr * (d.holding_cost * np.maximum(lev, 0) + d.penalty_cost * np.maximum(-lev, 0)))
                         + np.sum(pr * V_next[(nxt - d.min_inventory).astype(int)])))
    return np.array(out)


@pytest.mark.parametrize("period", [1, 6, 11])
def test_G_is_K_convex_as_the_reference_checks_it(cfg1, period):
    """CheckKConvexity.check: for all c < b < a, G(a) + K > G(b) + (a - b) (G(b) - G(c)) / (b - c) - 0.1."""
    w, V, _ = cfg1
    K = w.desc().fixed_order_cost
    ys = np.arange(-40, 95)  # (levels whose demand window stays inside the clamped grid)
    G = _G(w, V[period], period, ys)
    a, b, c = np.meshgrid(np.arange(len(ys)), np.arange(len(ys)), np.arange(len(ys)), indexing="ij")
    m = (c < b) & (b < a)
    lhs = G[a[m]] + K
    rhs = G[b[m]] + (a[m] - b[m]) * (G[b[m]] - G[c[m]]) / (b[m] - c[m]) - 0.1
    assert np.all(lhs > rhs), f"{int(np.sum(lhs <= rhs))} violated triples"


@pytest.mark.parametrize("period", [1, 6, 11])
def test_policy_is_s_S(cfg1, period):
    """K-convexity makes an (s, S) policy optimal: order up to ONE level S from every state at or below s, nothing above.
    (States from which S is out of reach of maxOrderQuantity order the maximum; the tie rule picks the lowest action.)"""
    w, V, pol = cfg1
    d = w.desc()
    x = np.arange(int(d.min_inventory), int(d.max_inventory) + 1)
    q = pol[period - 1]
    ordering = q > 0
    s = x[ordering].max()
    assert np.all(ordering[x <= s]) and not np.any(ordering[x > s])
    free = ordering & (q < d.max_order_quantity)  # not capped by the action list
    S = np.unique(x[free] + q[free])
    assert len(S) == 1 and S[0] > s
    assert np.all(q[ordering & ~free] == d.max_order_quantity)
    # V_t(x) = K + v (S - x) + G_t(S) below s, G_t(x) above: the value table agrees with G built from V_{t+1}
    G = _G(w, V[period], period, x.astype(float))
    Vt = V[period - 1]
    above = (x > s) & (x >= -40) & (x < 95)
    assert np.allclose(Vt[above], G[above], rtol=1e-12, atol=1e-9)
    below = free & (x >= -40)
    assert np.allclose(Vt[below], d.fixed_order_cost + d.unit_order_cost * (S[0] - x[below]) + G[x == S[0]][0], rtol=1e-12, atol=1e-9)

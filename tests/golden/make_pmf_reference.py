#!/usr/bin/env python3
"""Generate tests/golden/pmf_reference.json: the demand tiles of GetPmf.getpmf (GetPmf.java:82-134) and of CLSP.main's inline
construction (CLSP.java:219-247) in 50-digit arithmetic (mpmath), independent of scipy and of libsdpgpu.so's own cdf / quantile
code -- the pin for sdpgpu_getpmf that the reference cannot give (SSJ is absent and no PMF is recorded there).

    python tests/golden/make_pmf_reference.py        (run in the build container; mpmath 1.3.0)

Per case the file holds: the distribution parameters, the raw function values the formulas consume at 34 significant digits
(quantiles, pmf / cdf at every support point) and the resulting tile (support exactly, probabilities as decimal strings of the
exact quotient).  The STRUCTURE restated here is the reference's: (int) truncation of both quantiles; lower bound forced to 0 for
integer distributions; prob(j) indexed by POSITION j, not by the demand value (GetPmf.java:124); mass = cdf(ub) - cdf(lb - 1)
(discrete) or cdf(ub + step/2) - cdf(lb - step/2) (continuous); the CLSP variant keeps the un-truncated quantiles.
Every INPUT is taken as the fp64 number a caller passes (mp.mpf(float(x))): the level 0.9999 is the double nearest to it, and
1 - q is then exact, as it is in the reference's `1 - truncationQuantile`.  The script refuses a case whose quantile lies within 1e-9 of an integer (the (int) cast would then depend on the last bits
of whoever computes it) or whose Poisson cdf comes within 1e-12 of the quantile level.
"""
import json
import os

import mpmath as mp

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


def D(x):
    """The fp64 value of an input, exactly."""
    return mp.mpf(float(x))


def S(x, digits=34):
    return mp.nstr(mp.mpf(x), digits, strip_zeros=False)


class Poisson:
    kind, discrete = "poisson", True

    def __init__(self, lam):
        self.lam = D(lam)
        self.params = [float(lam), 0.0]

    def prob(self, k):
        k = int(k)
        return mp.exp(-self.lam) * self.lam ** k / mp.factorial(k) if k >= 0 else mp.mpf(0)

    def cdf(self, x):
        k = int(mp.floor(x))
        return mp.gammainc(k + 1, self.lam, mp.inf, regularized=True) if k >= 0 else mp.mpf(0)

    def inverseF(self, u):  # smallest x with F(x) >= u (SSJ's DiscreteDistributionInt.inverseF)
        u = mp.mpf(u)
        k = 0
        while self.cdf(k) < u:
            k += 1
        for kk in (k - 1, k):
            if kk >= 0 and abs(self.cdf(kk) - u) < mp.mpf("1e-12"):
                raise SystemExit(f"Poisson({self.lam}): cdf({kk}) is within 1e-12 of the level {u}: pick another case")
        return mp.mpf(k)


class Normal:
    kind, discrete = "normal", False

    def __init__(self, mu, sigma):
        self.mu, self.sigma = D(mu), D(sigma)
        self.params = [float(mu), float(sigma)]

    def cdf(self, x):
        return mp.ncdf((mp.mpf(x) - self.mu) / self.sigma)

    def inverseF(self, u):
        return self.mu + self.sigma * mp.sqrt(2) * mp.erfinv(2 * mp.mpf(u) - 1)


class Gamma:
    kind, discrete = "gamma", False  # shape alpha, RATE lambda (SSJ's GammaDist(alpha, lambda))

    def __init__(self, alpha, lam):
        self.alpha, self.lam = D(alpha), D(lam)
        self.params = [float(alpha), float(lam)]

    def cdf(self, x):
        x = mp.mpf(x)
        return mp.gammainc(self.alpha, 0, self.lam * x, regularized=True) if x > 0 else mp.mpf(0)

    def inverseF(self, u):
        u = mp.mpf(u)
        lo, hi = mp.mpf(0), self.alpha / self.lam + 1
        while self.cdf(hi) < u:
            hi *= 2
        for _ in range(400):
            mid = (lo + hi) / 2
            if self.cdf(mid) < u:
                lo = mid
            else:
                hi = mid
        return (lo + hi) / 2


def java_int(x):
    """(int) of a quantile, refusing values whose truncation is not robust."""
    r = mp.nint(x)
    if abs(x - r) < mp.mpf("1e-9"):
        raise SystemExit(f"quantile {x} is within 1e-9 of an integer: pick another case")
    return int(mp.floor(x)) if x >= 0 else -int(mp.floor(-x))


def getpmf_case(dists, q, step):
    """GetPmf.getpmf, GetPmf.java:82-134."""
    q, step = D(q), D(step)
    tiles = []
    for d in dists:
        ql, qu = d.inverseF(1 - q), d.inverseF(q)
        lb = 0 if dists[0].discrete else java_int(ql)
        ub = java_int(qu) if not d.discrete else int(qu)
        n = int(mp.floor((ub - lb + 1) / step))  # (int)((supportUB - supportLB + 1) / stepSize), :114
        support = [lb + j * step for j in range(n)]
        raw = {"quantile_lo": S(ql), "quantile_hi": S(qu)}
        if dists[0].discrete:
            mass = d.cdf(ub) - d.cdf(lb - 1)
            probs = [d.prob(j) / mass for j in range(n)]  # prob(j): position, :124
            raw["prob"] = [S(d.prob(j)) for j in range(n)]
            raw["cdf_ub"], raw["cdf_lb_minus_1"] = S(d.cdf(ub)), S(d.cdf(lb - 1))
        else:
            mass = d.cdf(ub + step / 2) - d.cdf(lb - step / 2)
            probs = [(d.cdf(x + step / 2) - d.cdf(x - step / 2)) / mass for x in support]
            raw["cdf_at_cell_edges"] = [S(d.cdf(lb - step / 2 + j * step)) for j in range(n + 1)]
        tiles.append({"support": [float(x) for x in support], "prob": [S(p) for p in probs], "raw": raw})
    return tiles


def clsp_case(dists, q, step):
    """CLSP.main's inline pmf, CLSP.java:219-247 (cdf-difference branch: PoissonDist is not a DiscreteDistribution)."""
    q, step = D(q), D(step)
    tiles = []
    for d in dists:
        lb, ub = d.inverseF(1 - q), d.inverseF(q)
        nreal = (ub - lb + 1) / step
        if abs(nreal - mp.nint(nreal)) < mp.mpf("1e-9") and not d.discrete:
            raise SystemExit("CLSP demandLength within 1e-9 of an integer: pick another case")
        n = int(mp.floor(nreal))
        support = [lb + j * step for j in range(n)]
        mass = d.cdf(ub + step / 2) - d.cdf(lb - step / 2)
        probs = [(d.cdf(x + step / 2) - d.cdf(x - step / 2)) / mass for x in support]
        tiles.append({"support": [S(x) for x in support], "prob": [S(p) for p in probs],
                      "raw": {"quantile_lo": S(lb), "quantile_hi": S(ub), "mass": S(mass)}})
    return tiles


CASES = [
    ("poisson_clsp_main", [Poisson(m) for m in (9, 23, 53, 29)], "0.9999", 1),
    ("poisson_cash", [Poisson(10)] * 2, "0.9999", 1),
    ("poisson_small_big", [Poisson("0.7"), Poisson("180.5"), Poisson(3)], "0.999", 1),
    ("normal_quarter_cv", [Normal(m, mp.mpf(m) / 4) for m in (20, 40, 60, 40)], "0.9999", 1),
    ("normal_step2", [Normal(50, 10), Normal("33.3", "7.7")], "0.99", 2),
    ("gamma_xr_main", [Gamma(8, 2)] * 2, "0.99", 1),
    ("gamma_mixed", [Gamma("2.5", "0.4"), Gamma(30, "1.5"), Gamma("0.8", "0.1")], "0.995", 1),
]


def main():
    out = {"generator": "tests/golden/make_pmf_reference.py", "mpmath": mp.__version__, "digits": mp.mp.dps, "cases": []}
    for name, dists, q, step in CASES:
        rec = {"name": name, "q": float(q), "step": float(step),
               "dists": [{"kind": d.kind, "a": d.params[0], "b": d.params[1]} for d in dists],
               "getpmf": getpmf_case(dists, q, step)}
        rec["clsp"] = clsp_case(dists, q, step)
        out["cases"].append(rec)
        print(name, [len(t["prob"]) for t in rec["getpmf"]], [len(t["prob"]) for t in rec["clsp"]], flush=True)
    with open(os.path.join(HERE, "pmf_reference.json"), "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")


if __name__ == "__main__":
    main()

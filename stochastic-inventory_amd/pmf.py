"""Demand-PMF construction: the step immediately BEFORE the hot path.

Restates `sdp.inventory.GetPmf.getpmf()` (src/sdp/inventory/GetPmf.java:82-134) and the inline
construction of `capacitated.CLSP.main` (src/capacitated/CLSP.java:219-247), including their
quirks, over small distribution classes with the three SSJ methods the reference calls
(`inverseF`, `cdf`, `prob`).  SSJ 3.3.0 itself (pom.xml:25-29) is not vendored in the reference
and not available here; the distribution functions below come from scipy.special / scipy.stats,
so **parity is unpinned at this boundary** (no reference test pins PMF values).  That does not
touch value parity: the recursion takes the PMF as an input array and the oracle and the GPU
consume the same one.

Host-side fp64 preprocessing (kilobytes); there is nothing here for a GPU to do.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np
from scipy import stats


def _d2i(x: float) -> int:
    """Java (int) cast."""
    if x != x:
        return 0
    return int(max(-2147483648.0, min(2147483647.0, x)))


class Distribution:
    """The slice of umontreal.ssj.probdist.Distribution the reference uses."""

    is_discrete_int = False

    def cdf(self, x: float) -> float:
        raise NotImplementedError

    def inverseF(self, u: float) -> float:
        raise NotImplementedError

    def getMean(self) -> float:
        raise NotImplementedError


class PoissonDist(Distribution):
    """umontreal.ssj.probdist.PoissonDist(lambda): a DiscreteDistributionInt."""

    is_discrete_int = True

    def __init__(self, lam: float):
        self.lam = float(lam)

    def prob(self, x: int) -> float:
        return float(stats.poisson.pmf(int(x), self.lam)) if x >= 0 else 0.0

    def cdf(self, x: float) -> float:
        return float(stats.poisson.cdf(math.floor(x), self.lam)) if x >= 0 else 0.0

    def inverseF(self, u: float) -> float:
        return float(stats.poisson.ppf(u, self.lam))  # smallest x with F(x) >= u, as SSJ

    def getMean(self) -> float:
        return self.lam


class NormalDist(Distribution):
    def __init__(self, mu: float, sigma: float):
        self.mu, self.sigma = float(mu), float(sigma)

    def cdf(self, x: float) -> float:
        return float(stats.norm.cdf(x, self.mu, self.sigma))

    def inverseF(self, u: float) -> float:
        return float(stats.norm.ppf(u, self.mu, self.sigma))

    def getMean(self) -> float:
        return self.mu

    def getStandardDeviation(self) -> float:
        return self.sigma


class GammaDist(Distribution):
    """umontreal.ssj.probdist.GammaDist(alpha, lambda): shape alpha, RATE lambda (mean alpha / lambda) -- the demand of
    cash.singleItem.CashConstraintXR.main (CashConstraintXR.java:67: `new GammaDist(meanDemand[i], 2)`)."""

    def __init__(self, alpha: float, lam: float):
        self.alpha, self.lam = float(alpha), float(lam)

    def cdf(self, x: float) -> float:
        return float(stats.gamma.cdf(x, self.alpha, scale=1.0 / self.lam)) if x > 0 else 0.0

    def inverseF(self, u: float) -> float:
        return float(stats.gamma.ppf(u, self.alpha, scale=1.0 / self.lam))

    def getMean(self) -> float:
        return self.alpha / self.lam


class UniformIntDist(Distribution):
    """umontreal.ssj.probdist.UniformIntDist(i, j)."""

    is_discrete_int = True

    def __init__(self, i: int, j: int):
        self.i, self.j = int(i), int(j)

    def getI(self):
        return self.i

    def getJ(self):
        return self.j

    def getXinf(self):
        return self.i

    def getXsup(self):
        return self.j

    def prob(self, x: int) -> float:
        return 1.0 / (self.j - self.i + 1.0) if self.i <= x <= self.j else 0.0

    def cdf(self, x: float) -> float:
        if x < self.i:
            return 0.0
        if x >= self.j:
            return 1.0
        return (math.floor(x) - self.i + 1.0) / (self.j - self.i + 1.0)

    def inverseF(self, u: float) -> float:
        return float(self.i + min(self.j - self.i, int(u * (self.j - self.i + 1.0))))

    def getMean(self) -> float:
        return 0.5 * (self.i + self.j)


class BinomialDist(Distribution):
    """umontreal.ssj.probdist.BinomialDist(n, p): `prob(j)` is all the workforce drivers call
    (WorkforcePlanning.java:61-66).  scipy's pmf stands in for SSJ's (parity unpinned at this boundary)."""

    is_discrete_int = True

    def __init__(self, n: int, p: float):
        self.n, self.p = int(n), float(p)

    def prob(self, j: int) -> float:
        return float(stats.binom.pmf(j, self.n, self.p))

    def cdf(self, x: float) -> float:
        return float(stats.binom.cdf(math.floor(x), self.n, self.p))

    def inverseF(self, u: float) -> float:
        return float(stats.binom.ppf(u, self.n, self.p))


def staff_level_pmf(turnoverRate: Sequence[float], xLength: int) -> np.ndarray:
    """The table the workforce drivers build (WorkforcePlanning.java:52-69, WorkforceTesting.java:66-82):
    out[t, i, j] = Binomial(i, turnoverRate[t]).prob(j) for j <= i, row 0 = {0: 1}; shape (T, xLength, xLength)."""
    out = np.zeros((len(turnoverRate), xLength, xLength))
    j = np.arange(xLength)
    for t, rate in enumerate(turnoverRate):
        out[t, 0, 0] = 1.0
        for i in range(1, xLength):
            out[t, i, : i + 1] = stats.binom.pmf(j[: i + 1], i, rate)
    return out


class DiscreteDistribution(Distribution):
    """umontreal.ssj.probdist.DiscreteDistribution(values, prob, n): values sorted ascending."""

    def __init__(self, values: Sequence[float], probs: Sequence[float], n: int = None):
        n = len(values) if n is None else n
        self.values = [float(v) for v in values[:n]]
        self.probs = [float(p) for p in probs[:n]]

    def getN(self) -> int:
        return len(self.values)

    def getValue(self, i: int) -> float:
        return self.values[i]

    def prob(self, i: int) -> float:
        return self.probs[i]

    def cdf(self, x: float) -> float:
        return sum(p for v, p in zip(self.values, self.probs) if v <= x)

    def inverseF(self, u: float) -> float:
        acc = 0.0
        for v, p in zip(self.values, self.probs):
            acc += p
            if acc >= u:
                return v
        return self.values[-1]

    def getMean(self) -> float:
        return sum(v * p for v, p in zip(self.values, self.probs))

    def tile(self) -> np.ndarray:
        """pmf[t] for a DiscreteDistribution demand (the commented-out variant at CashConstraint.java:79-89)."""
        return np.array([[v, p] for v, p in zip(self.values, self.probs)], dtype=np.float64)


class GetPmf:
    """sdp.inventory.GetPmf (GetPmf.java:23-30 constructor, :82-134 getpmf)."""

    def __init__(self, distributions: Sequence[Distribution], truncationQuantile: float, stepSize: float):
        self.distributions = list(distributions)
        self.truncationQuantile = float(truncationQuantile)
        self.stepSize = float(stepSize)

    def getpmf(self) -> List[np.ndarray]:
        dists, q, step = self.distributions, self.truncationQuantile, self.stepSize
        T = len(dists)
        supportLB, supportUB = [0.0] * T, [0.0] * T
        for i in range(T):
            supportLB[i] = float(_d2i(dists[i].inverseF(1 - q)))  # (int) truncation, GetPmf.java:87
            if dists[0].is_discrete_int:                           # lower bound forced to 0, :88-89
                supportLB[i] = 0.0
            supportUB[i] = float(_d2i(dists[i].inverseF(q)))       # :90
        pmf: List[np.ndarray] = []
        if isinstance(dists[0], UniformIntDist):                   # :97-111 (uses distributions[0] every period)
            d0 = dists[0]
            for _ in range(T):
                pmf.append(np.array([[float(j), d0.prob(j)] for j in range(d0.getXinf(), d0.getXsup() + 1)]))
            return pmf
        for i in range(T):
            demandLength = _d2i((supportUB[i] - supportLB[i] + 1) / step)  # :114
            tile = np.zeros((demandLength, 2), dtype=np.float64)
            for j in range(demandLength):
                tile[j, 0] = supportLB[i] + j * step               # :119
                if dists[0].is_discrete_int:                       # :120-124
                    probilitySum = dists[i].cdf(supportUB[i]) - dists[i].cdf(supportLB[i] - 1)
                    tile[j, 1] = dists[i].prob(j) / probilitySum   # prob(j): indexed by j, not by the demand value
                else:                                              # :125-129
                    probilitySum = dists[i].cdf(supportUB[i] + 0.5 * step) - dists[i].cdf(supportLB[i] - 0.5 * step)
                    tile[j, 1] = (dists[i].cdf(tile[j, 0] + 0.5 * step) - dists[i].cdf(tile[j, 0] - 0.5 * step)) / probilitySum
            pmf.append(tile)
        return pmf


def clsp_pmf(distributions: Sequence[Distribution], truncationQuantile: float, stepSize: float) -> List[np.ndarray]:
    """The inline PMF of capacitated.CLSP.main (CLSP.java:219-247): both quantiles un-truncated,
    discrete distributions normalised by 2q - 1 (CLSP.java:238-239), not by the covered mass."""
    q, step = float(truncationQuantile), float(stepSize)
    pmf = []
    for dist in distributions:
        lb = dist.inverseF(1 - q)
        ub = dist.inverseF(q)
        demandLength = _d2i((ub - lb + 1) / step)
        tile = np.zeros((demandLength, 2), dtype=np.float64)
        for j in range(demandLength):
            tile[j, 0] = lb + j * step
            demand = _d2i(tile[j, 0])
            # `distributions[0] instanceof DiscreteDistribution` (CLSP.java:236): in SSJ, PoissonDist derives
            # from DiscreteDistributionInt, which is NOT a DiscreteDistribution, so Poisson demands take
            # the cdf-difference branch below; only finite-support DiscreteDistribution takes this one
            # (where prob() is indexed by position, the reference passes the demand value).
            if isinstance(distributions[0], DiscreteDistribution):
                tile[j, 1] = dist.prob(demand) / (2 * q - 1)
            else:
                probabilitySum = dist.cdf(ub + 0.5 * step) - dist.cdf(lb - 0.5 * step)
                tile[j, 1] = (dist.cdf(tile[j, 0] + 0.5 * step) - dist.cdf(tile[j, 0] - 0.5 * step)) / probabilitySum
        pmf.append(tile)
    return pmf


def getpmf_native(distributions: Sequence[Distribution], truncationQuantile: float, stepSize: float,
                  clsp_variant: bool = False) -> List[np.ndarray]:
    """The same tiles from libsdpgpu.so's own GetPmf (sdpgpu_getpmf, csrc/sdpgpu_pmf.hip: no scipy, no SSJ) -- what a C
    or Java caller of the ABI gets.  Distributions: PoissonDist, NormalDist, UniformIntDist, GammaDist."""
    import ctypes as C

    from . import _abi
    lib = _abi.load()
    T = len(distributions)
    specs = (_abi.SdpgpuDistSpec * T)()
    for i, d in enumerate(distributions):
        if isinstance(d, PoissonDist):
            specs[i].kind, specs[i].a, specs[i].b = _abi.DIST_POISSON, d.lam, 0.0
        elif isinstance(d, NormalDist):
            specs[i].kind, specs[i].a, specs[i].b = _abi.DIST_NORMAL, d.mu, d.sigma
        elif isinstance(d, UniformIntDist):
            specs[i].kind, specs[i].a, specs[i].b = _abi.DIST_UNIFORM_INT, float(d.getI()), float(d.getJ())
        elif isinstance(d, GammaDist):
            specs[i].kind, specs[i].a, specs[i].b = _abi.DIST_GAMMA, d.alpha, d.lam
        else:
            raise TypeError(f"sdpgpu_getpmf has no {type(d).__name__}")
    out = []
    variant = _abi.PMF_CLSP if clsp_variant else _abi.PMF_GETPMF
    for t in range(T):
        n = C.c_int32(0)
        rc = lib.sdpgpu_getpmf(specs, T, float(truncationQuantile), float(stepSize), variant, t, None, None, 0, C.byref(n))
        if rc:
            raise _abi.SdpgpuError(rc, lib.sdpgpu_last_error(None).decode())
        dem = np.zeros(n.value)
        pr = np.zeros(n.value)
        rc = lib.sdpgpu_getpmf(specs, T, float(truncationQuantile), float(stepSize), variant, t,
                               dem.ctypes.data_as(C.POINTER(C.c_double)), pr.ctypes.data_as(C.POINTER(C.c_double)),
                               n.value, C.byref(n))
        if rc:
            raise _abi.SdpgpuError(rc, lib.sdpgpu_last_error(None).decode())
        out.append(np.stack([dem, pr], axis=1))
    return out

"""Forward simulation of the computed policy: the step immediately AFTER the hot path
 and the reference's only validation idiom.

Mirrors `sdp.inventory.Simulation` (src/sdp/inventory/Simulation.java:33-107) and the SDP part of
`sdp.cash.CashSimulation` (src/sdp/cash/CashSimulation.java:85-118): sample demand paths, round
them to integers (`Math.round`, Simulation.java:64), and per path walk
`getAction(state) -> immediateValue -> stateTransition` through the horizon.  Here the walk is a
batched table-lookup rollout on the device (`sdpgpu_simulate`, one path per lane) over the policy
tables the sweep left in HBM; only the sampling and the final mean stay on the host.

Differences, both deliberate:
* the reference's latin-hypercube sampler draws its jitter from `Math.random()`
  (Sampling.java:94) and is therefore not reproducible; this one takes a seed.
* `Arrays.stream(simuValues).sum()` is a compensated sum in Java; the mean here uses math.fsum.
  Per-path sums are accumulated on the device in period order exactly as the reference's
  `sum += ...`, so they are bit-identical to the CPU oracle's.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np

from .functors import java_round


class Sampling:
    """sdp.sampling.Sampling: latin hypercube / plain random sampling through inverseF."""

    def __init__(self, seed: int = 12345):
        self.rng = np.random.default_rng(seed)

    def generateLHSamples(self, distributions: Sequence, sampleNum: int) -> np.ndarray:
        """Sampling.java:86-103: one stratum [j/n, (j+1)/n) per sample and period, then a shuffle of the rows."""
        T = len(distributions)
        samples = np.empty((sampleNum, T), dtype=np.float64)
        for i in range(T):
            u = (np.arange(sampleNum) + self.rng.random(sampleNum)) / float(sampleNum)
            samples[:, i] = [distributions[i].inverseF(float(p)) for p in u]
        # Sampling.shuffle (Sampling.java:326-335): every period column is shuffled on its own, each
        # position swapped with a uniformly drawn one
        for i in range(T):
            marks = self.rng.integers(0, sampleNum, size=sampleNum)
            col = samples[:, i]
            for j in range(sampleNum):
                m = marks[j]
                col[j], col[m] = col[m], col[j]
        return samples

    def generateRanSamples(self, distributions: Sequence, sampleNum: int) -> np.ndarray:
        """Sampling.java:49-60."""
        T = len(distributions)
        samples = np.empty((sampleNum, T), dtype=np.float64)
        for i in range(T):
            samples[:, i] = [distributions[i].inverseF(float(p)) for p in self.rng.random(sampleNum)]
        return samples


def round_demands(samples: np.ndarray) -> np.ndarray:
    """`Math.round(samples[i][t])` element-wise (Simulation.java:64)."""
    return np.array([[float(java_round(float(v))) for v in row] for row in samples], dtype=np.float64)


class Simulation:
    """sdp.inventory.Simulation(distributions, sampleNum, recursion) -- also serves the cash classes
    (CashSimulation adds `Math.pow(discountFactor, t)` weights and `+ iniCash`, :108,114)."""

    def __init__(self, distributions: Sequence, sampleNum: int, recursion, discountFactor: float = 1.0,
                 seed: int = 12345):
        self.distributions = list(distributions)
        self.sampleNum = int(sampleNum)
        self.recursion = recursion
        self.discountFactor = float(discountFactor)
        self.stateTransition = recursion.getStateTransitionFunction()
        self.immediateValue = recursion.getImmediateValueFunction()
        self.sampling = Sampling(seed)
        self.last_values = None

    def setSampleNum(self, n: int):
        self.sampleNum = int(n)

    def _rollout(self, iniState, demands: np.ndarray) -> np.ndarray:
        rec = self.recursion
        rec.getExpectedValue(iniState)  # solves on first use, as Simulation.java:62 does
        T = rec.T
        disc = np.array([math.pow(self.discountFactor, t) for t in range(T)], dtype=np.float64)
        x, cash, preq = rec.functor.tuple_of(iniState)
        sums, valid = rec.engine.simulate(demands, disc, x, cash, preq)
        if not valid.all():
            raise RuntimeError(f"{int((~valid).sum())} sample paths left the state grid (demand outside the PMF support "
                               "of an unclamped family); the reference would re-enter the recursion there")
        return sums

    def simulateSDPGivenSamplNum(self, iniState) -> float:
        """Simulation.java:53-74: mean of the simulated totals over `sampleNum` LHS paths."""
        samples = self.sampling.generateLHSamples(self.distributions, self.sampleNum)
        sums = self._rollout(iniState, round_demands(samples))
        self.last_values = sums
        mean = math.fsum(sums.tolist()) / len(sums)
        if hasattr(iniState, "getIniCash"):  # CashSimulation.java:114
            mean += iniState.getIniCash()
        return mean

    def simulateSDPwithErrorConfidence(self, iniState, error: float, confidence: float, batch: int = 1000,
                                       maxRuns: int = 1000000):
        """Simulation.java:76-107: keep sampling until the normal confidence radius is below error * mean
        (at least 1000 runs).  Paths are rolled in batches on the device."""
        from scipy import stats as _st
        z = float(_st.norm.ppf(0.5 + confidence / 2.0))
        vals = np.empty(0)
        center, radius = 0.0, math.inf
        while len(vals) < 1000 or (radius >= center * error and len(vals) < maxRuns):
            samples = self.sampling.generateRanSamples(self.distributions, batch)
            vals = np.concatenate([vals, self._rollout(iniState, round_demands(samples))])
            center = float(vals.mean())
            radius = z * float(vals.std(ddof=1)) / math.sqrt(len(vals))
        self.last_values = vals
        return [center, radius]

    def simulateOnHost(self, iniState, demands: np.ndarray) -> np.ndarray:
        """The reference's loop verbatim (Simulation.java:59-69), one path at a time through the host
        lambdas and getAction -- for cross-checking the device rollout on a few paths."""
        out = np.empty(len(demands))
        for i, row in enumerate(demands):
            total, state = 0.0, iniState
            for t, d in enumerate(row):
                self.recursion.getExpectedValue(state)
                optQ = self.recursion.getAction(state)
                total += math.pow(self.discountFactor, t) * self.immediateValue(state, optQ, float(d))
                state = self.stateTransition(state, optQ, float(d))
            out[i] = total
        return out


class RiskSimulation:
    """sdp.cash.RiskSimulation(distributions, sampleNum, recursion): `simulateLostSale` (RiskSimulation.java:206-241),
    the validation run of the survival-probability recursion -- roll the policy along LHS demand paths, count the
    paths that ever hold negative cash and the paths that ever lose a demand.  The walk runs on the device
    (`sdpgpu_simulate`, family SURVIVAL); sampling and the two ratios stay on the host."""

    def __init__(self, distributions: Sequence, sampleNum: int, recursion, seed: int = 12345):
        self.distributions = list(distributions)
        self.sampleNum = int(sampleNum)
        self.recursion = recursion
        self.sampling = Sampling(seed)
        self.last_flags = None

    def simulateLostSale(self, iniState, immediateValue=None):
        """Returns [simulated survival probability, lost-sale rate] (RiskSimulation.java:237-240)."""
        samples = self.sampling.generateLHSamples(self.distributions, self.sampleNum)
        return self.simulateLostSaleOnDemands(iniState, round_demands(samples))

    def simulateLostSaleOnDemands(self, iniState, demands: np.ndarray):
        rec = self.recursion
        rec.getSurvProb(iniState)
        x, cash, _ = rec.functor.tuple_of(iniState)
        went_bankrupt, valid = rec.engine.simulate(demands, np.ones(rec.T), x, cash, 0.0)
        if not valid.all():
            raise RuntimeError("a sample path left the state grid")
        flags = rec.engine.last_sim_flags
        self.last_flags = flags
        lost = int(((flags >> 1) & 1).sum())
        sim_final = 1 - math.fsum(went_bankrupt.tolist()) / float(len(went_bankrupt))
        return [sim_final, lost / float(self.sampleNum)]

    def simulateOnHost(self, iniState, demands: np.ndarray):
        """The reference's loop verbatim through the host lambdas and getAction, for a few paths."""
        rec = self.recursion
        bankrupt = np.zeros(len(demands))
        lost = 0
        for i, row in enumerate(demands):
            state = iniState
            countBefore, countBeforeBankrupt = False, state.getBankruptBefore()
            for d in row:
                rec.getSurvProb(state)
                optQ = rec.getAction(state)
                if state.getIniCash() < 0:
                    optQ = 0.0
                if state.getIniInventory() + optQ < d and not countBefore:
                    lost += 1
                    countBefore = True
                thisValue = state.getIniCash() + rec.immediateValue(state, optQ, float(d))
                state = rec.stateTransition(state, optQ, float(d))
                if thisValue < 0 and not countBeforeBankrupt:
                    bankrupt[i] = 1
                    countBeforeBankrupt = True
        return [1 - bankrupt.sum() / float(len(demands)), lost / float(len(demands))]

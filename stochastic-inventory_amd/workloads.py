"""Synthetic workloads = the BASELINE.json configs, as laid out in SURVEY.md section 8(d).

Deterministic, closed-form inputs (no RNG, no data files): a D-point period has support
d = 0..D-1 with p_d proportional to the Poisson(lambda_t) pmf renormalised over the support
(the shape GetPmf.java:123-124 produces), lambda_t = (D/2)(1 + 0.25 sin(2 pi t / T)), so every
period has a distinct PMF tile.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List

import numpy as np

from .functors import BackorderFunctor, CashFunctor, LeadtimeFunctor
from .states import OptDirection


def truncated_poisson_tile(lam: float, D: int) -> np.ndarray:
    """[[d, p_d]] for d = 0..D-1, p_d = Poisson(lam) pmf / sum over the support."""
    d = np.arange(D, dtype=np.float64)
    logp = -lam + d * math.log(lam) - np.array([math.lgamma(k + 1.0) for k in range(D)])
    p = np.exp(logp - logp.max())
    p = p / p.sum()
    return np.stack([d, p], axis=1)


def seasonal_pmf(T: int, D: int) -> List[np.ndarray]:
    return [truncated_poisson_tile((D / 2.0) * (1.0 + 0.25 * math.sin(2.0 * math.pi * (t + 1) / T)), D)
            for t in range(T)]


@dataclass
class Workload:
    name: str
    functor: object
    direction: OptDirection
    pmf: List[np.ndarray]
    note: str = ""

    @property
    def T(self) -> int:
        return len(self.pmf)

    def desc(self):
        return self.functor.to_desc(self.T, self.direction)

    def overhead(self):
        return self.functor.overheads(self.T) if hasattr(self.functor, "overheads") else None


def cfg1_sS(T: int = 12) -> Workload:
    """configs[0]: single-item (s,S), Poisson(10) truncated at q = 0.9999 (support 0..24), 200 states."""
    f = BackorderFunctor(fixedOrderingCost=500, variOrderingCost=0, holdingCost=2, penaltyCost=10,
                         minInventory=-100, maxInventory=99, maxOrderQuantity=100, iniInventory=0)
    pmf = [truncated_poisson_tile(10.0, 25) for _ in range(T)]
    return Workload("cfg1_sS_200x101x25", f, OptDirection.MIN, pmf, "src/sdp Recursion plumbing case")


def cfg2_clsp(T: int = 52, S: int = 10000, A: int = 200, D: int = 100) -> Workload:
    """configs[1]: capacitated lot sizing, 1e4 states x 200 actions x 100 demands, 52 periods."""
    f = BackorderFunctor(fixedOrderingCost=500, variOrderingCost=1, holdingCost=2, penaltyCost=10,
                         minInventory=-(S // 2), maxInventory=S - S // 2 - 1, maxOrderQuantity=A - 1,
                         iniInventory=0)
    return Workload(f"cfg2_clsp_{S}x{A}x{D}x{T}", f, OptDirection.MIN, seasonal_pmf(T, D),
                    "src/capacitated CLSP.f")


def cfg3_cash(T: int = 6, NX: int = 200, NC: int = 5000, A: int = 300, D: int = 150) -> Workload:
    """configs[2]: cash-constrained 2-D state (inventory x cash), 1e6 states x <=300 actions x 150."""
    f = CashFunctor(price=10, fixOrderCost=0, variCost=1, holdingCost=0, depositeRate=0, overheadCost=0,
                    overheadRate=0, salvageValue=0.5, penaltyCost=0, discountFactor=1.0,
                    maxOrderQuantity=A - 1, minInventoryState=0, maxInventoryState=NX - 1,
                    minCashState=0, maxCashState=NC - 1, cashRoundMult=1.0, cashRoundDiv=1.0,
                    cashRoundIntDiv=True, cashFormula=0, iniInventory=0, iniCash=100)
    return Workload(f"cfg3_cash_{NX}x{NC}x{A}x{D}x{T}", f, OptDirection.MAX, seasonal_pmf(T, D),
                    "src/cash CashConstraint via CashRecursion, cash quantum 1")


def cfg4_leadtime(T: int = 50, NX: int = 1000, A: int = 200, D: int = 100) -> Workload:
    """configs[3]: the lead-time pipeline state of src/leadtime -- (period, inventory, preQ), the three
    fields of the reference's LeadtimeState (LeadtimeState.java:10-20) -- at about 1e7 state cells
    (50 periods x 1000 inventory points x 200 pipeline quantities), 200 actions, 100 demands; lead time 1
    exactly as Leadtime.java, on a clamped synthetic grid."""
    f = LeadtimeFunctor(fixedOrderingCost=0, variOrderingCost=1, holdingCost=2, penaltyCost=10,
                        maxOrderQuantity=A - 1, clampInventory=True, minInventory=-50,
                        maxInventory=NX - 51, iniInventory=0, iniPreQ=0)
    return Workload(f"cfg4_leadtime_{NX}x{A}q_{A}x{D}x{T}", f, OptDirection.MIN, seasonal_pmf(T, D),
                    "src/leadtime via LeadtimeRecursion")


def cfg4_pipeline(T: int = 4, NX: int = 250, A: int = 200, D: int = 100) -> Workload:
    """configs[3] as SURVEY.md section 8(d) lays it out: F2 generalised to lead time 2, 3-D state (x, q1, q2) with
    x in [-50, 199] and q1, q2 in [0, 199] = 1e7 states per period, 200 actions, 100 demands (2e11 cells per
    period), inventory clamped.  The reference has lead time 1 only (LeadtimeState.java:10-20): cfg4_leadtime is
    its exact shape, this one the synthetic generalisation."""
    f = LeadtimeFunctor(fixedOrderingCost=0, variOrderingCost=1, holdingCost=2, penaltyCost=10,
                        maxOrderQuantity=A - 1, clampInventory=True, minInventory=-50,
                        maxInventory=NX - 51, iniInventory=0, iniPreQ=0, leadTime=2, iniPreQ2=0)
    return Workload(f"cfg4_pipeline_{NX}x{A}q1x{A}q2_{A}x{D}x{T}", f, OptDirection.MIN, seasonal_pmf(T, D),
                    "src/leadtime generalised to a two-stage pipeline")


def cfg5_scaled(S: int, T: int = 3, A: int = 500, D: int = 200) -> Workload:
    """configs[4] / the north-star target grid: F1 scaled to S states, 500 actions, 200 demands."""
    f = BackorderFunctor(fixedOrderingCost=500, variOrderingCost=1, holdingCost=2, penaltyCost=10,
                         minInventory=0, maxInventory=S - 1, maxOrderQuantity=A - 1, iniInventory=0)
    return Workload(f"cfg5_f1_{S}x{A}x{D}x{T}", f, OptDirection.MIN, seasonal_pmf(T, D), "synthetic scaled F1")


def target_grid(T: int = 6, S: int = 1000000, A: int = 500, D: int = 200) -> Workload:
    """The grid BASELINE.json's target sentence is quoted on: 1e6 states x 500 actions x 200 demands (F1, the
    lambdas of capacitated.CLSP, CLSP.java:251-272), six periods = 6e11 cells per sweep."""
    w = cfg5_scaled(S=S, T=T, A=A, D=D)
    w.name = f"target_f1_{S}x{A}x{D}x{T}"
    w.note = "north-star target grid"
    return w


def cfg3_tenths(T: int = 6, NX: int = 501, maxCash: float = 2000.0, A: int = 101, D: int = 25) -> Workload:
    """configs[2]'s family at the size and cash quantum of the reference's own driver, cash.singleItem.CashConstraint
    .main (CashConstraint.java:44-68,131): cash in TENTHS (Math.round(cash * 10) / 10.0), inventory 0..500, cash
    0..2000 = 501 x 20001 states, orders 0..100, Poisson(10) demand truncated to 25 points, six periods.  Nothing is
    dyadic here, so the uniform-shift kernel does not apply: this is the cash row kernel's workload."""
    f = CashFunctor(price=10, fixOrderCost=0, variCost=1, holdingCost=0, depositeRate=0, overheadCost=0, overheadRate=0,
                    salvageValue=0.5, penaltyCost=0, discountFactor=1.0, maxOrderQuantity=A - 1, minInventoryState=0,
                    maxInventoryState=NX - 1, minCashState=0, maxCashState=maxCash, iniInventory=0, iniCash=100)
    pmf = [truncated_poisson_tile(10.0, D) for _ in range(T)]
    return Workload(f"cfg3t_cash_tenths_{NX}x{int(maxCash * 10) + 1}x{A}x{D}x{T}", f, OptDirection.MAX, pmf,
                    "CashConstraint.main: cash quantum 0.1")


def by_name(name: str, **kw) -> Workload:
    table = {"cfg1": cfg1_sS, "cfg2": cfg2_clsp, "cfg3": cfg3_cash, "cfg3t": cfg3_tenths, "cfg4": cfg4_leadtime,
             "cfg4p": cfg4_pipeline, "target": target_grid}
    if name in table:
        return table[name](**kw)
    if name == "cfg5":
        return cfg5_scaled(**kw)
    raise KeyError(name)

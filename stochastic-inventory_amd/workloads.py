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


def f5_single_product_leadtime(T: int = 4, mean: float = 20.0) -> Workload:
    """cash.overdraft.SingleProductLeadtime at the size its header calls the limit of the Java code -- "4 periods, mean
    demand 20 ... 50 s ... maximum computational capacity for java, otherwise memory errors" (SingleProductLeadtime.java:22-24):
    F5 (CashLeadtimeRecursion), state (inventory 0..60, cash -200..300 in HUNDREDTHS, preQ 0..30) = 9.5e7 states per period,
    orders 0..30, Poisson(20) truncated at 0.9999 (GetPmf.java:82-134)."""
    from .functors import CashLeadtimeFunctor
    from .pmf import GetPmf, PoissonDist
    f = CashLeadtimeFunctor(price=5, variCost=1, salvageValue=0.5, maxOrderQuantity=30, minInventoryState=0,
                            maxInventoryState=60, minCashState=-200, maxCashState=300, r0=0, r2=0.1, r3=2, limit=500,
                            interestFreeAmount=0, iniInventory=0, iniCash=0, iniPreQ=0, overheadCosts=[0.0] * T)
    pmf = [np.asarray(t, dtype=np.float64) for t in GetPmf([PoissonDist(mean)] * T, 0.9999, 1).getpmf()]
    return Workload(f"f5_spl_61x50001x31q_31x{len(pmf[0])}x{T}", f, OptDirection.MAX, pmf,
                    "SingleProductLeadtime.main via CashLeadtimeRecursion, cash quantum 0.01")


@dataclass
class StaffWorkload:
    """workforce.StaffRecursion: the pmf depends on the hire-up-to level (StaffRecursion.java:93-95), so the workload carries a
    (T, levels, stride) table instead of per-period tiles; `overhead()` hands the engine minStaffNum[t]."""
    name: str
    functor: object
    level_pmf: np.ndarray
    note: str = ""
    pmf = None
    direction = OptDirection.MIN

    @property
    def T(self) -> int:
        return int(self.level_pmf.shape[0])

    def desc(self):
        return self.functor.to_desc(self.T)

    def overhead(self):
        return [float(m) for m in self.functor.minStaffNum]


def staff_testing(T: int = 8, maxHire: int = 1000, rate: float = 0.1) -> StaffWorkload:
    """The first of WorkforceTesting.main's 216 instances (WorkforceTesting.java:45-107): T = 8, hires 0..1000, no clamp
    (staff 0..7000 by period 8), binomial turnover at rate turnoverRates[0] tabulated for 1001 hire-up-to levels."""
    from .pmf import staff_level_pmf
    from .workforce import StaffFunctor
    f = StaffFunctor(fixCost=50, unitVariCost=20, salary=30, unitPenalty=50, minStaffNum=[40] * T, maxHireNum=maxHire,
                     clampStaff=False, iniStaffNum=0)
    one = staff_level_pmf([rate], maxHire + 1)
    return StaffWorkload(f"staff_testing0_{maxHire + 1}hires_x{T}", f, np.repeat(one, T, axis=0), "WorkforceTesting.main[0] via StaffRecursion")


# capacitated.CLSP's three lambdas (CLSP.java:251-272) as the HIP device text a driver hands to sdpgpu_create_custom when its
# lambdas are its own: params = {K, v, h, pi, minInventory, maxInventory, maxOrderQuantity}.  bench.py's `custom_clsp` entry
# runs configs[1] through this text, i.e. through the reference's actual plugin API (arbitrary closures) instead of the
# built-in family.
CLSP_LAMBDAS_HIP = r"""
__device__ int sdp_feasible_count(const sdp_ctx& c, double x, double cash, double preq) {
  return (int)(c.params[6] / c.step) + 1;
}
__device__ double sdp_immediate(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand) {
  double fixedCost = action > 0 ? c.params[0] : 0;
  double variableCost = c.params[1] * action;
  double inventoryLevel = x + action - randomDemand;
  double holdingCosts = c.params[2] * sdp_max(inventoryLevel, 0);
  double penaltyCosts = c.params[3] * sdp_max(-inventoryLevel, 0);
  double totalCosts = fixedCost + variableCost + holdingCosts + penaltyCosts;
  return totalCosts;
}
__device__ void sdp_transition(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand,
                               double& nx, double& ncash, double& npreq) {
  double nextInventory = x + action - randomDemand;
  nextInventory = nextInventory > c.params[5] ? c.params[5] : nextInventory;
  nextInventory = nextInventory < c.params[4] ? c.params[4] : nextInventory;
  nx = nextInventory;
  ncash = 0;
  npreq = 0;
}
"""


# The same lambdas with the LEVEL SHAPE declared (include/sdpgpu.h, SDP_SHAPE_LEVEL): immediate value = the action's cost + the
# cost of the level the demand leaves, next inventory = that level, clamped.  The engine tabulates the two functions per period
# with the driver's own compiled code and runs the F1 window kernel from the tables.
CLSP_LAMBDAS_LEVEL_HIP = r"""
#define SDP_SHAPE_LEVEL 1
__device__ double sdp_action_cost(const sdp_ctx& c, double action) {
  double fixedCost = action > 0 ? c.params[0] : 0;
  double variableCost = c.params[1] * action;
  return fixedCost + variableCost;
}
__device__ double sdp_level_cost(const sdp_ctx& c, double inventoryLevel) {
  double holdingCosts = c.params[2] * sdp_max(inventoryLevel, 0);
  double penaltyCosts = c.params[3] * sdp_max(-inventoryLevel, 0);
  return holdingCosts + penaltyCosts;
}
"""


def clsp_lambda_params(w: Workload):
    f = w.functor
    return [f.fixedOrderingCost, f.variOrderingCost, f.holdingCost, f.penaltyCost, f.minInventory, f.maxInventory,
            f.maxOrderQuantity]


def custom_clsp(**kw) -> Workload:
    """configs[1] with CLSP's lambdas handed over as user text (hipRTC) instead of the built-in F1 family."""
    w = cfg2_clsp(**kw)
    w.name = "custom_clsp_" + w.name[len("cfg2_clsp_"):]
    w.note = "configs[1] through sdpgpu_create_custom (CLSP's lambdas as HIP text)"
    w.custom_source = CLSP_LAMBDAS_HIP
    w.custom_params = clsp_lambda_params(w)
    return w


def custom_clsp_level(**kw) -> Workload:
    """configs[1] with CLSP's lambdas as user text of the level shape: hipRTC compiles the two cost functions, the F1 window
    kernel runs them from tables."""
    w = cfg2_clsp(**kw)
    w.name = "custom_clsp_level_" + w.name[len("cfg2_clsp_"):]
    w.note = "configs[1] through sdpgpu_create_custom, lambdas of the declared level shape"
    w.custom_source = CLSP_LAMBDAS_LEVEL_HIP
    w.custom_params = clsp_lambda_params(w)
    return w


def by_name(name: str, **kw) -> Workload:
    table = {"cfg1": cfg1_sS, "cfg2": cfg2_clsp, "cfg3": cfg3_cash, "cfg3t": cfg3_tenths, "cfg4": cfg4_leadtime,
             "cfg4p": cfg4_pipeline, "target": target_grid, "f5_spl": f5_single_product_leadtime, "staff": staff_testing,
             "custom_clsp": custom_clsp, "custom_clsp_level": custom_clsp_level}
    if name in table:
        return table[name](**kw)
    if name == "cfg5":
        return cfg5_scaled(**kw)
    raise KeyError(name)

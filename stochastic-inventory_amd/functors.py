"""Functor descriptors: the closed-form lambda families the in-scope drivers close over.

A GPU cannot call the reference's three Java lambdas per cell (Recursion.java:49-52), so each
family is named by an enum + scalar parameters that the device code evaluates in the
reference's operation order.  Every descriptor class also restates its family on the HOST in
plain Python floats (IEEE fp64, no FMA) -- `feasibleActions`, `immediateValue`,
`stateTransition` -- for three uses only: (i) default lambdas for the simulators /
policy read-out, (ii) `Recursion.validateFunctor`, which samples cells and checks
user lambda == functor, (iii) host bookkeeping.  They are never used to compute value tables.

Field names follow the reference's local variable names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

from . import _abi
from .states import CashStateXR, CashLeadtimeState, CashState, LeadtimeState, OptDirection, RiskState, State


def java_round(x: float) -> int:
    """java.lang.Math.round(double): nearest long, ties toward +infinity."""
    f = math.floor(x)
    return int(f + 1) if (x - f) >= 0.5 else int(f)


def _d2i(x: float) -> int:
    if x != x:
        return 0
    return int(max(-2147483648.0, min(2147483647.0, x)))  # truncation toward zero


def _interest(f, before: float) -> float:
    # CashOverdraft.java:87-95
    if before >= 0:
        return -f.r0 * before
    if before >= -f.interestFreeAmount:
        return 0.0
    if before >= -f.limit:
        return f.r2 * (-before - f.interestFreeAmount)
    return f.r3 * (-before - f.limit) + f.r2 * (f.limit - f.interestFreeAmount)


@dataclass
class _Base:
    def _common(self, d, T: int, direction: OptDirection):
        d.periods = T
        d.direction = direction.value if isinstance(direction, OptDirection) else int(direction)
        return d


@dataclass
class BackorderFunctor(_Base):
    """F1: CLSP.java:251-272, CLSPTesting.java:78-106, CLSPforDraw.java:86-103, LevelFitsS.java:85-102."""

    fixedOrderingCost: float = 0.0
    variOrderingCost: float = 0.0  # proportionalOrderingCost in CLSP.java
    holdingCost: float = 0.0
    penaltyCost: float = 0.0
    minInventory: float = 0.0  # minState
    maxInventory: float = 0.0  # maxState
    maxOrderQuantity: float = 0.0
    stepSize: float = 1.0
    iniInventory: float = 0.0
    clampInventory: bool = True  # every in-scope F1 driver clamps (CLSP.java:257-258)

    state_type = State

    def to_desc(self, T: int, direction: OptDirection = OptDirection.MIN):
        d = _abi.desc_defaults()
        d.family = _abi.FAMILY_BACKORDER
        d.step = self.stepSize
        d.min_inventory, d.max_inventory = self.minInventory, self.maxInventory
        d.max_order_quantity = self.maxOrderQuantity
        d.clamp_inventory = 1 if self.clampInventory else 0
        d.fixed_order_cost, d.unit_order_cost = self.fixedOrderingCost, self.variOrderingCost
        d.holding_cost, d.penalty_cost = self.holdingCost, self.penaltyCost
        d.ini_inventory = self.iniInventory
        return self._common(d, T, direction)

    def make_state(self, period, x, cash=0.0, preq=0.0):
        return State(period, x)

    def tuple_of(self, s):
        return (s.getIniInventory(), 0.0, 0.0)

    def feasibleActions(self, s, T=None) -> List[float]:
        n = _d2i(self.maxOrderQuantity / self.stepSize) + 1
        return [k * self.stepSize for k in range(n)]

    def immediateValue(self, s, action, randomDemand, T=None) -> float:
        fixedCost = self.fixedOrderingCost if action > 0 else 0.0
        variableCost = self.variOrderingCost * action
        inventoryLevel = s.getIniInventory() + action - randomDemand
        holdingCosts = self.holdingCost * max(inventoryLevel, 0.0)
        penaltyCosts = self.penaltyCost * max(-inventoryLevel, 0.0)
        return fixedCost + variableCost + holdingCosts + penaltyCosts

    def stateTransition(self, s, action, randomDemand, T=None):
        nextInventory = s.getIniInventory() + action - randomDemand
        if self.clampInventory:
            nextInventory = self.maxInventory if nextInventory > self.maxInventory else nextInventory
            nextInventory = self.minInventory if nextInventory < self.minInventory else nextInventory
        return State(s.getPeriod() + 1, nextInventory)


@dataclass
class LeadtimeFunctor(_Base):
    """F2: Leadtime.java:50-81 (lead time 1; the inventory clamp is commented out there).

    leadTime = 2 is the synthetic two-stage pipeline of BASELINE configs[3] (the reference has lead time 1 only):
    state (x, q1, q2), x' = x + q1 - d, q1' = q2, q2' = action; same cost terms.  It exists at the engine level
    (SdpEngine / the C ABI); the reference-shaped LeadtimeRecursion mirror keeps the reference's two-field state.
    """

    fixedOrderingCost: float = 0.0
    variOrderingCost: float = 0.0
    holdingCost: float = 0.0
    penaltyCost: float = 0.0
    maxOrderQuantity: float = 0.0
    stepSize: float = 1.0
    clampInventory: bool = False
    minInventory: float = 0.0
    maxInventory: float = 0.0
    iniInventory: float = 0.0
    iniPreQ: float = 0.0
    leadTime: int = 1
    iniPreQ2: float = 0.0

    state_type = LeadtimeState

    def to_desc(self, T: int, direction: OptDirection = OptDirection.MIN):
        d = _abi.desc_defaults()
        d.family = _abi.FAMILY_LEADTIME
        d.step = self.stepSize
        d.min_inventory, d.max_inventory = self.minInventory, self.maxInventory
        d.max_order_quantity = self.maxOrderQuantity
        d.clamp_inventory = 1 if self.clampInventory else 0
        d.fixed_order_cost, d.unit_order_cost = self.fixedOrderingCost, self.variOrderingCost
        d.holding_cost, d.penalty_cost = self.holdingCost, self.penaltyCost
        d.ini_inventory, d.ini_preq = self.iniInventory, self.iniPreQ
        d.lead_time = int(self.leadTime)
        d.ini_preq2 = self.iniPreQ2 if self.leadTime == 2 else 0.0
        return self._common(d, T, direction)

    def make_state(self, period, x, cash=0.0, preq=0.0):
        return LeadtimeState(period, x, preq)

    def tuple_of(self, s):
        return (s.getIniInventory(), 0.0, s.getPreQ())

    def feasibleActions(self, s, T=None):
        n = _d2i(self.maxOrderQuantity / self.stepSize) + 1
        return [k * self.stepSize for k in range(n)]

    def immediateValue(self, s, action, randomDemand, T=None):
        fixedCost = self.fixedOrderingCost if action > 0 else 0.0
        variableCost = self.variOrderingCost * action
        inventoryLevel = s.getIniInventory() + s.getPreQ() - randomDemand
        holdingCosts = self.holdingCost * max(inventoryLevel, 0.0)
        penaltyCosts = self.penaltyCost * max(-inventoryLevel, 0.0)
        return fixedCost + variableCost + holdingCosts + penaltyCosts

    def stateTransition(self, s, action, randomDemand, T=None):
        nextInventory = s.getIniInventory() + s.getPreQ() - randomDemand
        if self.clampInventory:
            nextInventory = self.maxInventory if nextInventory > self.maxInventory else nextInventory
            nextInventory = self.minInventory if nextInventory < self.minInventory else nextInventory
        return LeadtimeState(s.getPeriod() + 1, nextInventory, action)


@dataclass
class CashFunctor(_Base):
    """F3: CashConstraint.java:95-133 (cashFormula 0) / CashConstraintTesting.java:110-148 (1)."""

    price: float = 0.0
    fixOrderCost: float = 0.0
    variCost: float = 1.0
    holdingCost: float = 0.0
    depositeRate: float = 0.0
    overheadCost: float = 0.0
    overheadRate: float = 0.0
    salvageValue: float = 0.0
    penaltyCost: float = 0.0
    discountFactor: float = 1.0
    maxOrderQuantity: float = 0.0
    stepSize: float = 1.0
    minInventoryState: float = 0.0
    maxInventoryState: float = 0.0
    minCashState: float = 0.0
    maxCashState: float = 0.0
    cashRoundMult: float = 10.0  # Math.round(nextCash * 10) / 10.0
    cashRoundDiv: float = 10.0
    cashRoundIntDiv: bool = False
    cashFormula: int = 0
    iniInventory: float = 0.0
    iniCash: float = 0.0
    overheadCosts: Optional[List[float]] = None  # per period, overrides overheadCost

    state_type = CashState
    family = _abi.FAMILY_CASH

    def to_desc(self, T: int, direction: OptDirection = OptDirection.MAX):
        d = _abi.desc_defaults()
        d.family = self.family
        d.step = self.stepSize
        d.min_inventory, d.max_inventory = self.minInventoryState, self.maxInventoryState
        d.max_order_quantity = self.maxOrderQuantity
        d.clamp_inventory = 1
        d.fixed_order_cost, d.unit_order_cost = self.fixOrderCost, self.variCost
        d.holding_cost, d.penalty_cost = self.holdingCost, self.penaltyCost
        d.price, d.salvage_value = self.price, self.salvageValue
        d.deposit_rate, d.overhead_cost, d.overhead_rate = self.depositeRate, self.overheadCost, self.overheadRate
        d.discount_factor = self.discountFactor
        d.min_cash, d.max_cash = self.minCashState, self.maxCashState
        d.cash_round_mult, d.cash_round_div = self.cashRoundMult, self.cashRoundDiv
        d.cash_round_int_div = 1 if self.cashRoundIntDiv else 0
        d.cash_formula = self.cashFormula
        d.ini_inventory, d.ini_cash = self.iniInventory, self.iniCash
        return self._common(d, T, direction)

    def overheads(self, T):
        return list(self.overheadCosts) if self.overheadCosts is not None else None

    def _oh(self, period):
        return self.overheadCosts[period - 1] if self.overheadCosts is not None else self.overheadCost

    def make_state(self, period, x, cash=0.0, preq=0.0):
        return CashState(period, x, cash)

    def tuple_of(self, s):
        return (s.getIniInventory(), s.getIniCash(), 0.0)

    def _round_cash(self, nextCash):
        r = java_round(nextCash * self.cashRoundMult)
        if self.cashRoundIntDiv:
            q = abs(r) // int(self.cashRoundDiv)
            return float(q if r >= 0 else -q)  # Java long division truncates toward zero
        return r / self.cashRoundDiv

    def feasibleActions(self, s, T=None):
        v = self.variCost
        num = s.getIniCash() - self._oh(s.getPeriod()) - self.fixOrderCost
        if v == 0:
            q = math.nan if num == 0 else math.copysign(math.inf, num)
        else:
            q = num / v
        m = max(0.0, q) if q == q else math.nan
        m = min(self.maxOrderQuantity, m) if m == m else math.nan
        maxQ = float(_d2i(m))
        return [k * self.stepSize for k in range(_d2i(maxQ) + 1)]

    def immediateValue(self, s, action, randomDemand, T=None):
        revenue = self.price * min(s.getIniInventory() + action, randomDemand)
        fixedCost = self.fixOrderCost if action > 0 else 0.0
        variableCost = self.variCost * action
        inventoryLevel = s.getIniInventory() + action - randomDemand
        holdCosts = self.holdingCost * max(inventoryLevel, 0.0)
        oh = self._oh(s.getPeriod())
        if self.cashFormula == 0:
            deposite = (s.getIniCash() - fixedCost - variableCost) * (1 + self.depositeRate)
            cashIncrement = (1 - self.overheadRate) * revenue + deposite - holdCosts - oh - s.getIniCash()
        else:
            cashIncrement = revenue - fixedCost - variableCost - holdCosts - oh
        salValue = self.salvageValue * max(inventoryLevel, 0.0) if s.getPeriod() == T else 0.0
        cashIncrement += salValue
        endCash = s.getIniCash() + cashIncrement
        if endCash < 0:
            cashIncrement += self.penaltyCost * endCash
        return cashIncrement

    def stateTransition(self, s, action, randomDemand, T=None):
        nextInventory = max(0.0, s.getIniInventory() + action - randomDemand)
        nextCash = s.getIniCash() + self.immediateValue(s, action, randomDemand, T)
        nextCash = self.maxCashState if nextCash > self.maxCashState else nextCash
        nextCash = self.minCashState if nextCash < self.minCashState else nextCash
        nextInventory = self.maxInventoryState if nextInventory > self.maxInventoryState else nextInventory
        nextInventory = self.minInventoryState if nextInventory < self.minInventoryState else nextInventory
        nextCash = self._round_cash(nextCash)
        return CashState(s.getPeriod() + 1, nextInventory, nextCash)


@dataclass
class CashXRFunctor(CashFunctor):
    """The lambdas of cash.singleItem.CashConstraintXR (CashConstraintXR.java:84-125; Chao 2008) under
    sdp.cash.CashRecursionXR: state (x, R) with R = cash + variCost * x, actions = order-up-to levels
    y = x, x + 1, ... up to max(x, R / variCost), `Math.round(nextCash * 1) / 1`.  maxOrderQuantity is declared in the
    driver (:50) and never read by its lambdas; iniCash is the R of the period-1 state (:132)."""

    cashRoundMult: float = 1.0
    cashRoundDiv: float = 1.0
    cashRoundIntDiv: bool = True
    cashFormula: int = 2

    state_type = CashStateXR

    def make_state(self, period, x, cash=0.0, preq=0.0):
        """`cash` is R (the engine's convention for this family, include/sdpgpu.h)."""
        return CashStateXR(period, x, cash, self.variCost)

    def tuple_of(self, s):
        return (s.getIniInventory(), s.getIniR(), 0.0)

    def feasibleActions(self, s, T=None):
        x = s.getIniInventory()
        maxY = x if s.getIniR() / self.variCost < x else s.getIniR() / self.variCost
        length = _d2i(maxY - x) + 1
        return [x + k * self.stepSize for k in range(length)]

    def immediateValue(self, s, actionY, randomDemand, T=None):
        revenue = self.price * min(actionY, randomDemand)
        action = actionY - s.getIniInventory()
        fixedCost = self.fixOrderCost if actionY > s.getIniInventory() else 0.0
        variableCost = self.variCost * action
        initCash = s.getIniR() - self.variCost * s.getIniInventory()
        deposite = (initCash - fixedCost - variableCost) * (1 + self.depositeRate)
        inventoryLevel = actionY - randomDemand
        holdCosts = self.holdingCost * max(inventoryLevel, 0.0)
        cashIncrement = (1 - self.overheadRate) * revenue + deposite - holdCosts - self._oh(s.getPeriod()) - initCash
        salValue = self.salvageValue * max(inventoryLevel, 0.0) if s.getPeriod() == T else 0.0
        cashIncrement += salValue
        return cashIncrement

    def stateTransition(self, s, actionY, randomDemand, T=None):
        nextInventory = max(0.0, actionY - randomDemand)
        initCash = s.getIniR() - self.variCost * s.getIniInventory()
        nextCash = initCash + self.immediateValue(s, actionY, randomDemand, T)
        nextCash = self.maxCashState if nextCash > self.maxCashState else nextCash
        nextCash = self.minCashState if nextCash < self.minCashState else nextCash
        nextInventory = self.maxInventoryState if nextInventory > self.maxInventoryState else nextInventory
        nextInventory = self.minInventoryState if nextInventory < self.minInventoryState else nextInventory
        nextCash = self._round_cash(nextCash)
        nextR = nextCash + self.variCost * nextInventory
        return CashStateXR(s.getPeriod() + 1, nextInventory, nextR, self.variCost)


@dataclass
class SurvivalFunctor(CashFunctor):
    """F6: the lambdas of cashSurvival.java:98-143 under RiskRecursion.getSurvProb (RiskRecursion.java:65-108):
    orders limited by cash / variCost (no overhead or fixed cost in the bound), immediate value without the
    end-cash penalty, `Math.round(nextCash * 1) / 1`.  price / variCost are arrays there, filled with one value
    (cashSurvival.java:52-55); scalars here."""

    cashRoundMult: float = 1.0
    cashRoundDiv: float = 1.0
    cashRoundIntDiv: bool = True

    state_type = RiskState
    family = _abi.FAMILY_SURVIVAL

    def make_state(self, period, x, cash=0.0, preq=0.0):
        return RiskState(period, x, cash, False)

    def feasibleActions(self, s, T=None):
        v = self.variCost
        c = s.getIniCash()
        q = (math.nan if c == 0 else math.copysign(math.inf, c)) if v == 0 else c / v
        maxQ = min(q, self.maxOrderQuantity) if q == q else math.nan
        maxQ = max(maxQ, 0.0) if maxQ == maxQ else math.nan
        return [k * self.stepSize for k in range(_d2i(maxQ) + 1)]

    def immediateValue(self, s, action, randomDemand, T=None):
        revenue = self.price * min(s.getIniInventory() + action, randomDemand)
        fixedCost = self.fixOrderCost if action > 0 else 0.0
        variableCost = self.variCost * action
        deposite = (s.getIniCash() - fixedCost - variableCost) * (1 + self.depositeRate)
        inventoryLevel = s.getIniInventory() + action - randomDemand
        holdCosts = self.holdingCost * max(inventoryLevel, 0.0)
        cashIncrement = revenue + deposite - holdCosts - self._oh(s.getPeriod()) - s.getIniCash()
        salValue = self.salvageValue * max(inventoryLevel, 0.0) if s.getPeriod() == T else 0.0
        cashIncrement += salValue
        return cashIncrement

    def stateTransition(self, s, action, randomDemand, T=None):
        nextInventory = max(0.0, s.getIniInventory() + action - randomDemand)
        nextCash = s.getIniCash() + self.immediateValue(s, action, randomDemand, T)
        nextCash = self.maxCashState if nextCash > self.maxCashState else nextCash
        nextCash = self.minCashState if nextCash < self.minCashState else nextCash
        nextInventory = self.maxInventoryState if nextInventory > self.maxInventoryState else nextInventory
        nextInventory = self.minInventoryState if nextInventory < self.minInventoryState else nextInventory
        nextCash = self._round_cash(nextCash)
        return RiskState(s.getPeriod() + 1, nextInventory, nextCash, nextCash < 0)


@dataclass
class OverdraftFunctor(CashFunctor):
    """F4: CashOverdraft.java:72-118."""

    r0: float = 0.0
    r2: float = 0.0
    r3: float = 0.0
    limit: float = 0.0
    interestFreeAmount: float = 0.0
    cashRoundIntDiv: bool = True  # `/ 10` long division at CashOverdraft.java:116

    family = _abi.FAMILY_OVERDRAFT

    def to_desc(self, T: int, direction: OptDirection = OptDirection.MAX):
        d = super().to_desc(T, direction)
        d.r0, d.r2, d.r3 = self.r0, self.r2, self.r3
        d.overdraft_limit, d.interest_free_amount = self.limit, self.interestFreeAmount
        return d

    def feasibleActions(self, s, T=None):
        return [k * self.stepSize for k in range(_d2i(self.maxOrderQuantity) + 1)]

    def immediateValue(self, s, action, randomDemand, T=None):
        revenue = self.price * min(s.getIniInventory() + action, randomDemand)
        fixedCost = self.fixOrderCost if action > 0 else 0.0
        variableCost = self.variCost * action
        inventoryLevel = s.getIniInventory() + action - randomDemand
        before = s.getIniCash() - fixedCost - variableCost - self._oh(s.getPeriod())
        interest = _interest(self, before)
        after = before - interest + revenue
        cashIncrement = after - s.getIniCash()
        salValue = self.salvageValue * max(inventoryLevel, 0.0) if s.getPeriod() == T else 0.0
        cashIncrement += salValue
        return cashIncrement


@dataclass
class CashLeadtimeFunctor(OverdraftFunctor):
    """F5: SingleProductLeadtime.java:72-119."""

    cashRoundMult: float = 100.0
    cashRoundDiv: float = 100.0
    cashRoundIntDiv: bool = False
    zeroOrderLastPeriod: bool = True  # SingleProductLeadtime.java:74-75
    iniPreQ: float = 0.0

    state_type = CashLeadtimeState
    family = _abi.FAMILY_CASH_LEADTIME

    def to_desc(self, T: int, direction: OptDirection = OptDirection.MAX):
        d = super().to_desc(T, direction)
        d.zero_order_last_period = 1 if self.zeroOrderLastPeriod else 0
        d.ini_preq = self.iniPreQ
        return d

    def make_state(self, period, x, cash=0.0, preq=0.0):
        return CashLeadtimeState(period, x, cash, preq)

    def tuple_of(self, s):
        return (s.getIniInventory(), s.getIniCash(), s.getPreQ())

    def feasibleActions(self, s, T=None):
        maxQ = self.maxOrderQuantity
        if self.zeroOrderLastPeriod and s.getPeriod() == T:
            maxQ = 0
        return [k * self.stepSize for k in range(_d2i(maxQ) + 1)]

    def immediateValue(self, s, action, randomDemand, T=None):
        revenue = self.price * min(s.getIniInventory() + s.getPreQ(), randomDemand)
        variableCost = self.variCost * action
        inventoryLevel = s.getIniInventory() + s.getPreQ() - randomDemand
        before = s.getIniCash() - variableCost - self._oh(s.getPeriod())
        interest = _interest(self, before)
        after = before - interest + revenue
        cashIncrement = after - s.getIniCash()
        salValue = self.salvageValue * max(inventoryLevel, 0.0) if s.getPeriod() == T else 0.0
        cashIncrement += salValue
        return cashIncrement

    def stateTransition(self, s, action, randomDemand, T=None):
        nextInventory = max(0.0, s.getIniInventory() + s.getPreQ() - randomDemand)
        nextCash = s.getIniCash() + self.immediateValue(s, action, randomDemand, T)
        nextCash = self.maxCashState if nextCash > self.maxCashState else nextCash
        nextCash = self.minCashState if nextCash < self.minCashState else nextCash
        nextInventory = self.maxInventoryState if nextInventory > self.maxInventoryState else nextInventory
        nextInventory = self.minInventoryState if nextInventory < self.minInventoryState else nextInventory
        nextCash = self._round_cash(nextCash)
        return CashLeadtimeState(s.getPeriod() + 1, nextInventory, nextCash, action)


@dataclass
class CustomFunctor(_Base):
    """User-defined lambdas (sdpgpu_create_custom): `source` is the HIP device text of sdp_feasible_count /
    sdp_immediate / sdp_transition, `params` the doubles it reads through c.params, `shape` the built-in functor
    whose STATE SHAPE, grid fields and loop it borrows (its cost fields are ignored).  The Python callables
    restate the same lambdas on the host for the mirror classes (policy read-out, validateFunctor); they take
    and return the reference's state objects, exactly like the Java lambdas."""

    shape: object = None
    source: str = ""
    params: List[float] = field(default_factory=list)
    getFeasibleAction: Optional[object] = None   # state -> list of actions
    stateTransitionFn: Optional[object] = None   # (state, action, demand) -> state
    immediateValueFn: Optional[object] = None    # (state, action, demand) -> float

    def __getattr__(self, name):  # stepSize, iniInventory, ... come from the shape
        if name in ("shape", "source", "params"):
            raise AttributeError(name)
        return getattr(self.shape, name)

    def to_desc(self, T, direction=OptDirection.MIN):
        return self.shape.to_desc(T, direction)

    def overheads(self, T):
        return self.shape.overheads(T) if hasattr(self.shape, "overheads") else None

    def make_state(self, *a, **k):
        return self.shape.make_state(*a, **k)

    def tuple_of(self, s):
        return self.shape.tuple_of(s)

    def feasibleActions(self, s, T=None):
        return list(self.getFeasibleAction(s))

    def immediateValue(self, s, action, randomDemand, T=None):
        return self.immediateValueFn(s, action, randomDemand)

    def stateTransition(self, s, action, randomDemand, T=None):
        return self.stateTransitionFn(s, action, randomDemand)

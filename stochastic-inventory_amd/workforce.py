"""Host-side mirror of the reference's workforce recursion, backed by the HIP engine (STAFF family).

    workforce.StaffState        src/workforce/StaffState.java:4-35
    workforce.StaffRecursion    src/workforce/StaffRecursion.java:16-118, 262-279

Same names, argument meaning and behaviour for the path the drivers solve -- `getExpectedValue(StaffState)`
(WorkforcePlanning.java:104-112, WorkforceTesting.java:116-124): the turnover pmf of a period is picked by the
hire-up-to level, `pmfs[t][min(iniStaffNum + orderQty, pmfs[t].length - 1)]` (StaffRecursion.java:92-95).  As with the
other mirrors the three lambdas are kept for the caller's own use and a `functor` descriptor names the closed-form
family the device evaluates.  The G(y)-drawing variants (`getExpectedValue(state, hireUpStaffNum)`,
`getExpectedValue2`, `getExpectedValueNoHireFirst`, StaffRecursion.java:127-330) are plotting helpers over special
period-1 lambdas and are not built.  There is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _abi
from .engine import SdpEngine


class StaffState:
    """workforce.StaffState: (period, iniStaffNum), both ints."""

    __slots__ = ("period", "iniStaffNum")

    def __init__(self, period: int, iniStaffNum: int):
        object.__setattr__(self, "period", int(period))
        object.__setattr__(self, "iniStaffNum", int(iniStaffNum))

    def __setattr__(self, *a):
        raise AttributeError("StaffState is immutable")

    def __eq__(self, o):
        return isinstance(o, StaffState) and o.period == self.period and o.iniStaffNum == self.iniStaffNum

    def __hash__(self):
        return hash((self.period, self.iniStaffNum))

    def __repr__(self):
        return f"period = {self.period}, initial staff number = {self.iniStaffNum}"


@dataclass
class StaffFunctor:
    """F7: the lambdas of WorkforcePlanning.java:72-101 (clampStaff True) / WorkforceTesting.java:80-107 (False)."""

    fixCost: float = 0.0
    unitVariCost: float = 0.0
    salary: float = 0.0
    unitPenalty: float = 0.0
    minStaffNum: Sequence[int] = field(default_factory=list)  # per period
    maxHireNum: int = 0
    minX: int = 0
    maxX: int = 0
    clampStaff: bool = True
    iniStaffNum: int = 0

    stepSize = 1

    def to_desc(self, T: int):
        if len(self.minStaffNum) != T:
            raise ValueError("minStaffNum needs one entry per period")
        d = _abi.desc_defaults()
        d.family = _abi.FAMILY_STAFF
        d.direction = _abi.MIN
        d.periods = T
        d.step = 1.0
        d.min_inventory, d.max_inventory = float(self.minX), float(self.maxX)
        d.clamp_inventory = 1 if self.clampStaff else 0
        d.ini_inventory = float(self.iniStaffNum)
        d.max_order_quantity = float(self.maxHireNum)
        d.fixed_order_cost, d.unit_order_cost = self.fixCost, self.unitVariCost
        d.holding_cost, d.penalty_cost = self.salary, self.unitPenalty
        return d

    # host restatement (default lambdas; never used to compute tables)
    def feasibleActions(self, s) -> List[int]:
        return list(range(0, self.maxHireNum + 1))

    def immediateValue(self, s, action: int, randomDemand: int) -> float:
        fixHireCost = self.fixCost if action > 0 else 0.0
        variHireCost = self.unitVariCost * action
        nextStaffNum = s.iniStaffNum + action - randomDemand
        salaryCost = self.salary * nextStaffNum
        t = s.period - 1
        penaltyCost = 0.0 if nextStaffNum > self.minStaffNum[t] else self.unitPenalty * (self.minStaffNum[t] - nextStaffNum)
        return fixHireCost + variHireCost + salaryCost + penaltyCost

    def stateTransition(self, s, action: int, randomDemand: int) -> StaffState:
        nextStaffNum = s.iniStaffNum + action - randomDemand
        if self.clampStaff:
            nextStaffNum = self.maxX if nextStaffNum > self.maxX else nextStaffNum
            nextStaffNum = self.minX if nextStaffNum < self.minX else nextStaffNum
        return StaffState(s.period + 1, nextStaffNum)


def pack_level_pmf(pmf) -> tuple:
    """The reference's `double[T][rows][len][2]` (pmfs[t][y][j] = {j, prob}) -> (table (T, rows, stride), row_len or
    None).  An ndarray (T, rows, stride) is passed through (row y then has y + 1 entries)."""
    if isinstance(pmf, np.ndarray) and pmf.ndim == 3:
        return np.ascontiguousarray(pmf, dtype=np.float64), None
    T, rows = len(pmf), len(pmf[0])
    lens = [len(r) for r in pmf[0]]
    stride = max(max(len(r) for r in per) for per in pmf)
    table = np.zeros((T, rows, stride))
    for t, per in enumerate(pmf):
        if len(per) != rows or [len(r) for r in per] != lens:
            raise ValueError("every period must have the same table shape")
        for y, row in enumerate(per):
            arr = np.asarray(row, dtype=np.float64).reshape(-1, 2)
            if not np.array_equal(arr[:, 0], np.arange(len(arr))):
                raise ValueError(f"pmfs[{t}][{y}][j][0] must be j (WorkforcePlanning.java:64)")
            table[t, y, :len(arr)] = arr[:, 1]
    default = lens == [y + 1 for y in range(rows)]
    return table, (None if default else np.asarray(lens, dtype=np.int32))


class StaffRecursion:
    """workforce.StaffRecursion(getFeasibleAction, stateTransition, immediateValue, pmf, T) -- StaffRecursion.java:42-56."""

    def __init__(self, getFeasibleAction: Optional[Callable] = None, stateTransition: Optional[Callable] = None,
                 immediateValue: Optional[Callable] = None, pmf=None, T: Optional[int] = None, *,
                 functor: Optional[StaffFunctor] = None, device: int = -1):
        if functor is None:
            raise TypeError("a StaffFunctor descriptor is required: the GPU engine evaluates closed-form lambda families")
        if pmf is None:
            raise TypeError("pmf: the level-dependent table pmfs[t][y][j]")
        table, row_len = pack_level_pmf(pmf)
        self.T = int(T) if T is not None else table.shape[0]
        if table.shape[0] != self.T:
            raise ValueError(f"pmf has {table.shape[0]} periods, T = {self.T}")
        self.pmfs = pmf
        self.functor = functor
        self.getFeasibleAction = getFeasibleAction or functor.feasibleActions
        self.stateTransition = stateTransition or functor.stateTransition
        self.immediateValue = immediateValue or functor.immediateValue
        desc = functor.to_desc(self.T)
        desc.device = device
        self._engine = SdpEngine(desc, None, [float(m) for m in functor.minStaffNum], level_pmf=table,
                                 level_row_len=row_len)
        self._solved = False
        self._tables = {}

    def getStateTransitionFunction(self):
        return self.stateTransition

    def getImmediateValueFunction(self):
        return self.immediateValue

    @property
    def engine(self) -> SdpEngine:
        return self._engine

    def _lookup(self, state: StaffState):
        if not self._solved:
            self._engine.solve(sync=True)
            self._solved = True
        period = state.period
        if period < 1 or period > self.T:
            raise IndexError(f"period {period} outside 1..{self.T}")
        if period not in self._tables:
            self._tables[period] = (self._engine.values(period), self._engine.policy(period))
        idx = self._engine.state_index(period, float(state.iniStaffNum), 0.0, 0.0)
        if idx < 0:
            raise KeyError(f"{state!r} lies outside the staff numbers the recursion can reach")
        v, p = self._tables[period]
        return float(v[idx]), int(p[idx])

    def getExpectedValue(self, state: StaffState, hireUpStaffNum: Optional[int] = None) -> float:
        if hireUpStaffNum is not None:
            raise NotImplementedError("the G(y)-drawing variant (StaffRecursion.java:127-172) is not built")
        return self._lookup(state)[0]

    def getAction(self, state: StaffState) -> int:
        return self._lookup(state)[1]

    def getOptTable(self) -> np.ndarray:
        """Rows {period, iniStaffNum, action} of the visited states in comparator order (StaffRecursion.java:245-254)."""
        self._lookup(StaffState(1, self.functor.iniStaffNum))
        rows = []
        for period in range(1, self.T + 1):
            idx = np.nonzero(self._engine.reachable(period))[0]
            if len(idx) == 0:
                continue
            x_lo = self._engine.grid(period)[0]
            if period not in self._tables:
                self._tables[period] = (self._engine.values(period), self._engine.policy(period))
            pol = self._tables[period][1]
            rows.append(np.stack([np.full(len(idx), float(period)), x_lo + idx.astype(np.float64),
                                  pol[idx].astype(np.float64)], axis=1))
        return np.concatenate(rows, axis=0) if rows else np.zeros((0, 3))

    def close(self):
        self._engine.close()

"""State tuples of the reference, same names and getters.

State            src/sdp/inventory/State.java:12-76
LeadtimeState    src/sdp/inventory/LeadtimeState.java:10-52
CashState        src/sdp/cash/CashState.java:12-48
CashLeadtimeState src/sdp/cash/CashLeadtimeState.java:11-46
RiskState        src/sdp/cash/RiskState.java:12-52

Equality is exact `==` on the doubles, as in the reference's equals(); the objects are
immutable and hashable so they can key Python dicts the way they key the Java maps.
"""
from __future__ import annotations

import enum


class OptDirection(enum.Enum):
    """Recursion.java:44-47 / CashRecursion.java:34-37."""

    MIN = 0
    MAX = 1


class State:
    __slots__ = ("period", "initialInventory")

    def __init__(self, period: int, initialInventory: float):
        object.__setattr__(self, "period", int(period))
        object.__setattr__(self, "initialInventory", float(initialInventory))

    def __setattr__(self, *a):
        raise AttributeError("State is immutable")

    def getPeriod(self) -> int:
        return self.period

    def getIniInventory(self) -> float:
        return self.initialInventory

    def _key(self):
        return (self.period, self.initialInventory)

    def __eq__(self, o):
        return type(o) is type(self) and o._key() == self._key()

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return f"period = {self.period}, initialInventory = {self.initialInventory}"


class LeadtimeState(State):
    __slots__ = ("preQ",)

    def __init__(self, period: int, initialInventory: float, preQ: float):
        super().__init__(period, initialInventory)
        object.__setattr__(self, "preQ", float(preQ))

    def getPreQ(self) -> float:
        return self.preQ

    def _key(self):
        return (self.period, self.initialInventory, self.preQ)

    def __repr__(self):
        return f"period = {self.period}, initialInventory = {self.initialInventory}, preQ = {self.preQ}"


class CashState(State):
    __slots__ = ("iniCash",)

    def __init__(self, period: int, initialInventory: float, iniCash: float):
        super().__init__(period, initialInventory)
        object.__setattr__(self, "iniCash", float(iniCash))

    def getIniCash(self) -> float:
        return self.iniCash

    def _key(self):
        return (self.period, self.initialInventory, self.iniCash)

    def __repr__(self):
        return f"period = {self.period}, iniInventory = {self.initialInventory}, iniCash = {self.iniCash}"


class CashStateXR(State):
    """sdp.cash.CashStateXR (CashStateXR.java:14-57): (period, inventory x, working capital R = cash + variCost * x);
    the unit cost rides along for getOptTable's S column and takes no part in equality (:43-50)."""
    __slots__ = ("iniR", "unitVariCost")

    def __init__(self, period: int, iniInventory: float, R: float, variCost: float = 0.0):
        super().__init__(period, iniInventory)
        object.__setattr__(self, "iniR", float(R))
        object.__setattr__(self, "unitVariCost", float(variCost))

    def getIniR(self) -> float:
        return self.iniR

    def _key(self):
        return (self.period, self.initialInventory, self.iniR)

    def __repr__(self):
        return f"period = {self.period}, iniInventory = {self.initialInventory}, iniR = {self.iniR}"


class RiskState(CashState):
    """RiskState.java:12-52.  The constructor there ignores its `bankruptBefore` argument (`:17` assigns the
    literal false), and RiskRecursion's comparator (RiskRecursion.java:39-42) does not look at the flag."""
    __slots__ = ("bankruptBefore",)

    def __init__(self, period: int, initialInventory: float, iniCash: float, bankruptBefore: bool = False):
        super().__init__(period, initialInventory, iniCash)
        object.__setattr__(self, "bankruptBefore", False)

    def getBankruptBefore(self) -> bool:
        return self.bankruptBefore

    def __repr__(self):
        return (f"period = {self.period}, iniInventory = {self.initialInventory}, iniCash = {self.iniCash}, "
                f"bankuptBefore = {self.bankruptBefore}")


class CashLeadtimeState(CashState):
    __slots__ = ("preQ",)

    def __init__(self, period: int, initialInventory: float, iniCash: float, preQ: float):
        super().__init__(period, initialInventory, iniCash)
        object.__setattr__(self, "preQ", float(preQ))

    def getPreQ(self) -> float:
        return self.preQ

    def _key(self):
        return (self.period, self.initialInventory, self.iniCash, self.preQ)

    def __repr__(self):
        return (f"period = {self.period}, initialInventory = {self.initialInventory}, "
                f"iniCash = {self.iniCash}, preQ = {self.preQ}")

"""Two-product lead-time family on the reachable-set engine (sdpgpu_multilead_solve).

Host mirror of `sdp.cash.multiItem.CashRecursionMultiLead` as `cash.overdraft.MultiProductLeadtime.main`
uses it (src/cash/overdraft/MultiProductLeadtime.java:87-239): the lambdas there are fixed in form, so the
mirror takes their parameters, not closures.  This is the only family the reference records outputs
for (the comment block at MultiProductLeadtime.java:30-50).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List

from . import _abi
import numpy as np

from ._abi import SdpgpuError, SdpgpuMulticash, SdpgpuMultilead


@dataclass
class MultiLeadResult:
    finalValue: float          # iniCash + getExpectedValue(iniState)
    firstAction: int           # getAction(iniState).getFirstAction()
    secondAction: int
    statesPerPeriod: List[int]
    cells: int
    gpu_ms: float
    # rows {period, i1, i2, q1, q2, cash (R for the XR family), value, action 1, action 2} of every visited state in the
    # order of the reference's TreeMap keys -- what getCacheActions() / getOptTable() iterate; None unless asked for
    table: object = None


def fill_multilead(k, *, T, q_bound, price, vari_cost, sal_value, ini_cash, ini_i1, ini_i2, r0, r1, r2, limit,
                   interest_free, min_inventory, max_inventory, min_cash, max_cash, discount, overhead, values, probs,
                   cash_int_cast=False):
    k.T, k.q_bound = T, q_bound
    k.cash_int_cast = 1 if cash_int_cast else 0
    for name, val in (("price", price), ("vari_cost", vari_cost), ("sal_value", sal_value)):
        getattr(k, name)[0], getattr(k, name)[1] = val
    for name, val in (("ini_cash", ini_cash), ("ini_i1", ini_i1), ("ini_i2", ini_i2), ("r0", r0), ("r1", r1), ("r2", r2),
                      ("limit", limit), ("interest_free", interest_free), ("min_inventory", min_inventory),
                      ("max_inventory", max_inventory), ("min_cash", min_cash), ("max_cash", max_cash),
                      ("discount", discount)):
        setattr(k, name, float(val))
    for t, v in enumerate(overhead):
        k.overhead[t] = float(v)
    k.n1, k.n2 = len(values[0]), len(values[1])
    for i, (v, p) in enumerate(zip(values[0], probs[0])):
        k.v1[i], k.p1[i] = float(v), float(p)
    for i, (v, p) in enumerate(zip(values[1], probs[1])):
        k.v2[i], k.p2[i] = float(v), float(p)
    return k


def _run(call, T: int, want_table: bool) -> MultiLeadResult:
    """call(fv, q1, q2, states, cells, ms) -> rc.  With want_table the solve is run a second time with a table large
    enough for every visited state (the first run says how many there are)."""
    lib = _abi.load()

    def once():
        fv, q1, q2 = C.c_double(), C.c_int32(), C.c_int32()
        states = (C.c_int64 * T)()
        cells, ms = C.c_int64(), C.c_double()
        rc = call(C.byref(fv), C.byref(q1), C.byref(q2), states, C.byref(cells), C.byref(ms))
        if rc:
            raise SdpgpuError(rc, lib.sdpgpu_multilead_last_error().decode())
        return MultiLeadResult(fv.value, q1.value, q2.value, list(states), cells.value, ms.value)

    res = once()
    if want_table:
        t, arrs = _abi.make_multi_table(sum(res.statesPerPeriod))
        lib.sdpgpu_multi_set_table(C.byref(t))
        try:
            res = once()
        finally:
            lib.sdpgpu_multi_set_table(None)
        res.table = _abi.multi_table_rows(t, arrs)
    return res


def multilead_solve(table: bool = False, **kw) -> MultiLeadResult:
    """One call = `recursion.getExpectedValue(iniState)` + `getAction(iniState)` of MultiProductLeadtime.main;
    table=True also returns the whole memo (see MultiLeadResult.table)."""
    lib = _abi.load()
    k = fill_multilead(SdpgpuMultilead(), **kw)
    return _run(lambda *out: lib.sdpgpu_multilead_solve(C.byref(k), *out), k.T, table)


def fill_multicash(k, *, T, q_bound, price, vari_cost, sal_price, ini_cash, ini_i1, ini_i2, min_inventory, max_inventory,
                   min_cash, max_cash, discount, pmf):
    """pmf[t] = rows {d1, d2, probability}: what GetPmfMulti.getPmf(t) returns (GetPmfMulti.java:41-69).  The
    arrays the struct points to are kept alive on `k._keep`."""
    k.T, k.q_bound = T, q_bound
    for name, val in (("price", price), ("vari_cost", vari_cost), ("sal_price", sal_price)):
        getattr(k, name)[0], getattr(k, name)[1] = val
    for name, val in (("ini_cash", ini_cash), ("ini_i1", ini_i1), ("ini_i2", ini_i2), ("min_inventory", min_inventory),
                      ("max_inventory", max_inventory), ("min_cash", min_cash), ("max_cash", max_cash),
                      ("discount", discount)):
        setattr(k, name, float(val))
    if len(pmf) != T:
        raise ValueError(f"pmf has {len(pmf)} periods, T = {T}")
    tiles = [np.asarray(t, dtype=np.float64).reshape(-1, 3) for t in pmf]
    off = np.concatenate([[0], np.cumsum([len(t) for t in tiles])]).astype(np.int32)
    allp = np.concatenate(tiles, axis=0)
    d1, d2, p = (np.ascontiguousarray(allp[:, c]) for c in range(3))
    k._keep = (off, d1, d2, p)
    k.pmf_off = off.ctypes.data_as(C.POINTER(C.c_int32))
    k.d1, k.d2, k.p = (a.ctypes.data_as(C.POINTER(C.c_double)) for a in (d1, d2, p))
    return k


def multicash_solve(table: bool = False, **kw) -> MultiLeadResult:
    """`sdp.cash.multiItem.CashRecursionMulti` as `cash.multiItem.MultiItemCash.main` sets it up
    (MultiItemCash.java:66-132; the lambdas are fixed in form, so the mirror takes their parameters):
    one call = `iniCash + recursion.getExpectedValue(iniState)` + `getAction(iniState)`; table=True also returns the
    rows `getOptTable` is built from."""
    lib = _abi.load()
    k = fill_multicash(SdpgpuMulticash(), **kw)
    return _run(lambda *out: lib.sdpgpu_multicash_solve(C.byref(k), *out), k.T, table)


def multixr_solve(depositeRate: float = 0.0, table: bool = False, **kw) -> MultiLeadResult:
    """`sdp.cash.multiItem.CashRecursionMultiXR` as `cash.multiItem.MultiItemCashXR.main` sets it up
    (MultiItemCashXR.java:92-164): state (x1, x2, R), actions = order-up-to levels; `ini_cash` is the R of the period-1
    state.  firstAction / secondAction of the result are y1 / y2 (`recursion.getAction(iniState)[0]`, `[1]`)."""
    lib = _abi.load()
    k = fill_multicash(SdpgpuMulticash(), **kw)
    return _run(lambda *out: lib.sdpgpu_multixr_solve(C.byref(k), C.c_double(depositeRate), *out), k.T, table)


# ---------------------------------------------------------------------------------------------------------------
# Mirror classes: the reference's names and methods over the solvers above.  As elsewhere the lambdas are accepted
# and kept for the caller's own use (simulators), and a functor -- here simply the keyword arguments of the solver --
# names the closed-form family the device evaluates.  The first getExpectedValue runs the solve and reads the whole
# memo back, so every state the reference's recursion would have visited can be asked for afterwards.
# ---------------------------------------------------------------------------------------------------------------
class Actions:
    """sdp.cash.multiItem.Actions (Actions.java:17-33)."""

    def __init__(self, action1: int, action2: int):
        self.action1, self.action2 = int(action1), int(action2)

    def getFirstAction(self) -> int:
        return self.action1

    def getSecondAction(self) -> int:
        return self.action2

    def __eq__(self, o):
        return isinstance(o, Actions) and (o.action1, o.action2) == (self.action1, self.action2)

    def __repr__(self):
        return f"Actions({self.action1}, {self.action2})"


class CashStateMulti:
    """sdp.cash.multiItem.CashStateMulti (CashStateMulti.java:14-70)."""

    def __init__(self, period: int, iniInventory1: float, iniInventory2: float, iniCash: float):
        self.period, self.iniInventory1, self.iniInventory2, self.iniCash = int(period), float(iniInventory1), float(iniInventory2), float(iniCash)

    def getPeriod(self): return self.period
    def getIniInventory1(self): return self.iniInventory1
    def getIniInventory2(self): return self.iniInventory2
    def getIniCash(self): return self.iniCash

    def getIniR(self, variCosts):
        return self.iniCash + variCosts[0] * self.iniInventory1 + variCosts[1] * self.iniInventory2

    def _key(self): return (self.period, self.iniInventory1, self.iniInventory2, 0.0, 0.0, self.iniCash)


class CashStateMultiXR:
    """sdp.cash.multiItem.CashStateMultiXR (CashStateMultiXR.java:21-73): (period, x1, x2, R)."""

    def __init__(self, period: int, iniInventory1: float, iniInventory2: float, R: float):
        self.period, self.iniInventory1, self.iniInventory2, self.iniR = int(period), float(iniInventory1), float(iniInventory2), float(R)

    def getPeriod(self): return self.period
    def getIniInventory1(self): return self.iniInventory1
    def getIniInventory2(self): return self.iniInventory2
    def getIniR(self): return self.iniR
    def _key(self): return (self.period, self.iniInventory1, self.iniInventory2, 0.0, 0.0, self.iniR)


class CashStateMultiLead:
    """sdp.cash.multiItem.CashStateMultiLead (CashStateMultiLead.java:10-80)."""

    def __init__(self, period: int, iniInventory1: float, iniInventory2: float, preQ1: float, preQ2: float, iniCash: float):
        self.period = int(period)
        self.iniInventory1, self.iniInventory2 = float(iniInventory1), float(iniInventory2)
        self.preQ1, self.preQ2, self.iniCash = float(preQ1), float(preQ2), float(iniCash)

    def getPeriod(self): return self.period
    def getIniInventory1(self): return self.iniInventory1
    def getIniInventory2(self): return self.iniInventory2
    def getPreQ1(self): return self.preQ1
    def getPreQ2(self): return self.preQ2
    def getIniCash(self): return self.iniCash

    def getIniR(self, variCosts):
        return self.iniCash + variCosts[0] * self.iniInventory1 + variCosts[1] * self.iniInventory2

    def _key(self): return (self.period, self.iniInventory1, self.iniInventory2, self.preQ1, self.preQ2, self.iniCash)


class _MultiRecursionBase:
    _solver = None       # multilead_solve / multicash_solve / multixr_solve
    _state_type = None

    def __init__(self, discountFactor, Pmf, buildActionList=None, stateTransition=None, immediateValue=None,
                 TLength=None, *, functor=None):
        if functor is None:
            raise TypeError("functor: the parameters of the driver's lambdas (the keyword arguments of the solver)")
        self.discountFactor, self.Pmf, self.TLength = discountFactor, Pmf, TLength
        self.buildActionList, self.stateTransition, self.immediateValue = buildActionList, stateTransition, immediateValue
        self.functor = dict(functor)
        self.functor["discount"] = discountFactor
        if TLength is not None:
            self.functor["T"] = int(TLength)
        self._result = None
        self._memo = None

    def _solve(self, state):
        if self._result is None:
            f = self.functor
            ini = state._key()
            if ini[0] != 1:
                raise ValueError("the first getExpectedValue must be asked for a period-1 state (it fixes the reachable set)")
            self._result = self._run(f, ini)
            t = self._result.table
            self._memo = {tuple(r[:6]): (float(r[6]), int(r[7]), int(r[8])) for r in t}
        return self._result

    def _lookup(self, state):
        self._solve(state)
        k = state._key()
        k = (float(k[0]),) + k[1:]
        if k not in self._memo:
            raise KeyError(f"state {k} was not visited from the initial state (the reference's getAction would return null)")
        return self._memo[k]

    def getExpectedValue(self, state) -> float:
        return self._lookup(state)[0]

    def getCacheActions(self):
        """{state key (period, i1, i2, preQ1, preQ2, cash or R): action pair} in the reference's key order."""
        return {k: v[1:] for k, v in sorted(self._memo.items())} if self._memo else {}

    @property
    def result(self) -> MultiLeadResult:
        return self._result


class CashRecursionMultiLead(_MultiRecursionBase):
    """sdp.cash.multiItem.CashRecursionMultiLead (CashRecursionMultiLead.java:31-101) for the lambdas of
    MultiProductLeadtime.main; functor = the keyword arguments of multilead_solve (initial state and T come from the call)."""

    def _run(self, f, ini):
        kw = dict(f, ini_i1=ini[1], ini_i2=ini[2], ini_cash=ini[5])
        return multilead_solve(table=True, **kw)

    def getAction(self, state: CashStateMultiLead) -> Actions:
        _, a1, a2 = self._lookup(state)
        return Actions(a1, a2)


class CashRecursionMulti(_MultiRecursionBase):
    """sdp.cash.multiItem.CashRecursionMulti (CashRecursionMulti.java:39-211) for the lambdas of MultiItemCash.main;
    functor = the keyword arguments of multicash_solve; `Pmf` = the per-period lists GetPmfMulti.getPmf(t) returns."""

    def _run(self, f, ini):
        kw = dict(f, ini_i1=ini[1], ini_i2=ini[2], ini_cash=ini[5], pmf=self.Pmf)
        return multicash_solve(table=True, **kw)

    def getAction(self, state: CashStateMulti) -> Actions:
        _, a1, a2 = self._lookup(state)
        return Actions(a1, a2)

    def getOptTable(self, variCost):
        """CashRecursionMulti.java:171-199: rows {period, x1, x2, w, R, boolAlpha, alpha, Q1, Q2, c1, c2}."""
        rows = []
        for (period, x1, x2, _, _, w), (_, Q1, Q2) in sorted(self._memo.items()):
            alpha, boolAlpha = 10000.0, 0.0
            R = w + x1 * variCost[0] + x2 * variCost[1]
            if w <= variCost[0] * Q1 + variCost[1] * Q2 and Q1 > 0 and Q2 > 0:
                boolAlpha = 1.0
                alpha = variCost[0] * Q1 / w
            rows.append([period, x1, x2, w, R, boolAlpha, alpha, float(Q1), float(Q2), variCost[0], variCost[1]])
        return np.array(rows)


class CashRecursionMultiXR(_MultiRecursionBase):
    """sdp.cash.multiItem.CashRecursionMultiXR (CashRecursionMultiXR.java:39-148) for the lambdas of
    MultiItemCashXR.main; functor = the keyword arguments of multixr_solve (+ "depositeRate")."""

    def _run(self, f, ini):
        kw = dict(f, ini_i1=ini[1], ini_i2=ini[2], ini_cash=ini[5], pmf=self.Pmf)
        dep = kw.pop("depositeRate", 0.0)
        return multixr_solve(dep, table=True, **kw)

    def getAction(self, state: CashStateMultiXR):
        """the optimal order-up-to levels {y1, y2} (CashRecursionMultiXR.java:103-105)."""
        _, y1, y2 = self._lookup(state)
        return [float(y1), float(y2)]

    def getOptTable(self, variCost):
        """CashRecursionMultiXR.java:122-148: rows {period, x1, x2, w, R, boolAlpha, alpha, y1, y2, c1, c2}."""
        rows = []
        for (period, x1, x2, _, _, R), (_, y1, y2) in sorted(self._memo.items()):
            Q1, Q2 = y1 - x1, y2 - x2
            alpha, boolAlpha = 10000.0, 0.0
            w = R - x1 * variCost[0] - x2 * variCost[1]
            if R <= variCost[0] * y1 + variCost[1] * y2 + 0.1 and Q1 > 0.1 and Q2 > 0.1:
                boolAlpha = 1.0
                alpha = variCost[0] * y1 / R
            rows.append([period, x1, x2, w, R, boolAlpha, alpha, float(y1), float(y2), variCost[0], variCost[1]])
        return np.array(rows)

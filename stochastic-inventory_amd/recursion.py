"""Host-side mirror of the reference's recursion classes, backed by the HIP engine.

Same public names, argument meaning and behaviour as

    sdp.inventory.Recursion              src/sdp/inventory/Recursion.java:33-188
    sdp.inventory.LeadtimeRecursion      src/sdp/inventory/LeadtimeRecursion.java:19-104
    sdp.cash.CashRecursion               src/sdp/cash/CashRecursion.java:23-218
    sdp.cash.CashLeadtimeRecursion       src/sdp/cash/CashLeadtimeRecursion.java:19-107
    capacitated.CLSP (f / cacheActions)  src/capacitated/CLSP.java:61-138

with ONE addition: a `functor` descriptor (functors.py) naming the closed-form family the three
lambdas belong to -- a GPU cannot call host closures per cell.  The lambdas are still accepted
and kept (getStateTransitionFunction / getImmediateValueFunction serve the simulators exactly
as in Simulation.java:39-40); `validateFunctor` checks on sampled cells that they compute what
the functor computes, and runs by itself before the first solve whenever a lambda was supplied
(`recursion.strict = False` turns that off).

`getExpectedValue(state)` runs the full backward sweep on the GPU on its first call (the
reference fills its memo maps on the first call too, Recursion.java:89-163) and answers from
the device tables afterwards.  There is no CPU fallback.
"""
from __future__ import annotations

import random
from typing import Callable, Dict, Optional

import numpy as np

from .engine import SdpEngine
from .states import OptDirection


class _GpuRecursionBase:
    _direction_fixed: Optional[OptDirection] = None

    def __init__(self, optDirection, pmf, getFeasibleAction=None, stateTransition=None, immediateValue=None,
                 functor=None, discountFactor: float = 1.0, device: int = -1, kernel: int = 0):
        if functor is None:
            raise TypeError(
                "a functor descriptor is required: the GPU engine evaluates closed-form lambda families, "
                "see stochastic-inventory_amd/functors.py")
        self.optDirection = optDirection
        self.pmf = pmf  # shared by reference, never copied (Recursion.java:54)
        self.functor = functor
        self.T = len(pmf)
        T = self.T
        # lambdas the caller supplied are checked against the functor before the first solve (`strict`, on by
        # default): a driver whose lambdas are NOT the closed-form family would otherwise get tables for a
        # different model without a word, and the simulators would then roll the host lambdas against that policy
        self._user_lambdas = any(f is not None for f in (getFeasibleAction, stateTransition, immediateValue))
        self.strict = True
        self.getFeasibleActions = getFeasibleAction or (lambda s: functor.feasibleActions(s, T))
        self.stateTransition = stateTransition or (lambda s, a, r: functor.stateTransition(s, a, r, T))
        self.immediateValue = immediateValue or (lambda s, a, r: functor.immediateValue(s, a, r, T))
        self.discountFactor = discountFactor
        holder = getattr(functor, "shape", None) or functor  # a CustomFunctor borrows its grid from `shape`
        if hasattr(holder, "discountFactor"):
            holder.discountFactor = discountFactor
        desc = functor.to_desc(T, optDirection)
        desc.device = device
        desc.kernel = kernel
        overhead = functor.overheads(T) if hasattr(functor, "overheads") else None
        source = getattr(functor, "source", None) or None  # CustomFunctor: the lambdas as HIP device text
        self._engine = SdpEngine(desc, pmf, overhead, custom_source=source,
                                 custom_params=getattr(functor, "params", None) if source else None)
        self._solved = False
        self._values: Dict[int, np.ndarray] = {}
        self._policy: Dict[int, np.ndarray] = {}
        self._extra: Dict[object, tuple] = {}  # off-grid states answered by eval_states

    # -- reference API ----------------------------------------------------------------------
    def getStateTransitionFunction(self) -> Callable:
        return self.stateTransition

    def getImmediateValueFunction(self) -> Callable:
        return self.immediateValue

    def setTreeMapCacheAction(self):
        """Recursion.java:80-86 swaps the action map for a TreeMap; the dense tables are already
        in comparator order, so there is nothing to do."""

    def _solve(self):
        if not self._solved:
            if self._user_lambdas and self.strict and all(hasattr(self.functor, m) for m in
                                                          ("feasibleActions", "immediateValue", "stateTransition")):
                self.validateFunctor(nSamples=96)  # raises ValueError on the first cell that differs
            self._engine.solve(sync=True)
            self._solved = True

    def _table(self, period: int):
        if period not in self._values:
            self._values[period] = self._engine.values(period)
            self._policy[period] = self._engine.policy(period)
        return self._values[period], self._policy[period]

    def _lookup(self, state):
        self._solve()
        period = state.getPeriod()
        if period < 1 or period > self.T:
            raise IndexError(f"period {period} outside 1..{self.T}")
        x, cash, preq = self.functor.tuple_of(state)
        idx = self._engine.state_index(period, x, cash, preq)
        if idx >= 0:
            v, p = self._table(period)
            return float(v[idx]), int(p[idx])
        if state not in self._extra:
            val, act = self._engine.eval_states(period, [x], [cash], [preq])
            self._extra[state] = (float(val[0]), int(act[0]))
        return self._extra[state]

    def getExpectedValue(self, state) -> float:
        return self._lookup(state)[0]

    def getAction(self, state) -> float:
        return self._lookup(state)[1] * self.functor.stepSize

    def getCacheActions(self) -> Dict[object, float]:
        """The reference returns its (state -> action) map of VISITED states; here: the reachable set."""
        return {self.functor.make_state(int(r[0]), *self._row_tuple(r)): float(r[-1]) for r in self.getOptTable()}

    def _row_tuple(self, row):
        raise NotImplementedError

    def _opt_columns(self, period, idx, pol):
        raise NotImplementedError

    def getOptTable(self) -> np.ndarray:
        """Rows in the comparator order of the reference's sorted map, reachable states only
        (Recursion.java:177-186 iterates cacheActions, which holds exactly the visited states)."""
        self._solve()
        rows = []
        for period in range(1, self.T + 1):
            mask = self._engine.reachable(period)
            idx = np.nonzero(mask)[0]
            if len(idx) == 0:
                continue
            _, pol = self._table(period)
            rows.append(self._opt_columns(period, idx, pol[idx].astype(np.float64) * self.functor.stepSize))
        ini = self._initial_state()
        if self._engine.state_index(1, *self.functor.tuple_of(ini)) < 0:  # off-grid period-1 state
            rows.insert(0, self._row_of_state(ini, self.getAction(ini)))
        out = np.concatenate(rows, axis=0) if rows else np.zeros((0, self._ncols()))
        return out

    def _initial_state(self):
        f = self.functor
        return f.make_state(1, getattr(f, "iniInventory", 0.0), getattr(f, "iniCash", 0.0), getattr(f, "iniPreQ", 0.0))

    def _row_of_state(self, s, action):
        raise NotImplementedError

    def _ncols(self):
        raise NotImplementedError

    # -- additions --------------------------------------------------------------------------
    @property
    def engine(self) -> SdpEngine:
        return self._engine

    def validateFunctor(self, nSamples: int = 256, seed: int = 1) -> int:
        """Check on sampled (state, action, demand) cells that the lambdas passed to the constructor
        compute exactly what the functor family computes.  Returns the number of cells checked;
        raises ValueError on the first mismatch."""
        rng = random.Random(seed)
        f, T = self.functor, self.T
        checked = 0
        for _ in range(nSamples):
            period = rng.randint(1, T)
            x_lo, nx, nc, nq = self._engine.grid(period)
            ix, ic, iq = rng.randrange(nx), rng.randrange(nc), rng.randrange(nq)
            cash = self._engine.cash_value(ic) if nc > 1 or hasattr(f, "minCashState") else 0.0
            s = f.make_state(period, x_lo + ix * f.stepSize, cash, iq * f.stepSize)
            acts_ref = list(self.getFeasibleActions(s))
            acts = f.feasibleActions(s, T)
            if [float(a) for a in acts_ref] != [float(a) for a in acts]:
                raise ValueError(f"feasible actions differ at {s}: lambda {acts_ref[:5]}.. vs functor {acts[:5]}..")
            a = rng.choice(acts)
            d = rng.choice(self.pmf[period - 1])[0]
            i1, i2 = self.immediateValue(s, a, d), f.immediateValue(s, a, d, T)
            if i1 != i2:
                raise ValueError(f"immediateValue differs at {s}, a={a}, d={d}: lambda {i1!r} vs functor {i2!r}")
            if period < T:
                n1, n2 = self.stateTransition(s, a, d), f.stateTransition(s, a, d, T)
                if f.tuple_of(n1) != f.tuple_of(n2) or n1.getPeriod() != n2.getPeriod():
                    raise ValueError(f"stateTransition differs at {s}, a={a}, d={d}: {n1} vs {n2}")
            checked += 1
        return checked


class Recursion(_GpuRecursionBase):
    """sdp.inventory.Recursion (Recursion.java:49-63): rows of getOptTable are {period, x, Q}."""

    def __init__(self, optDirection, pmf, getFeasibleAction=None, stateTransition=None, immediateValue=None, *,
                 functor=None, device: int = -1, kernel: int = 0):
        super().__init__(optDirection, pmf, getFeasibleAction, stateTransition, immediateValue, functor=functor,
                         device=device, kernel=kernel)

    def _opt_columns(self, period, idx, q):
        x_lo, nx, nc, nq = self._engine.grid(period)
        x = x_lo + idx.astype(np.float64) * self.functor.stepSize
        return np.stack([np.full(len(idx), float(period)), x, q], axis=1)

    def _row_tuple(self, r):
        return (r[1],)

    def _row_of_state(self, s, action):
        return np.array([[float(s.getPeriod()), s.getIniInventory(), action]])

    def _ncols(self):
        return 3


class CLSP(Recursion):
    """capacitated.CLSP: the self-contained copy of the same loop (CLSP.java:88-138), MIN only.
    `f(state)` is its name for getExpectedValue; `cacheActions` its public action map."""

    def __init__(self, pmf, *, functor=None, device: int = -1, kernel: int = 0):
        super().__init__(OptDirection.MIN, pmf, functor=functor, device=device, kernel=kernel)

    def f(self, state) -> float:
        return self.getExpectedValue(state)

    @property
    def cacheActions(self):
        return self.getCacheActions()


class LeadtimeRecursion(_GpuRecursionBase):
    """sdp.inventory.LeadtimeRecursion (LeadtimeRecursion.java:28-45): MIN only; rows
    {period, x, preQ, Q} ordered by (period, x, preQ) (LeadtimeRecursion.java:37-40,93-102)."""

    def __init__(self, pmf, getFeasibleAction=None, stateTransition=None, immediateValue=None, *, functor=None,
                 device: int = -1, kernel: int = 0):
        super().__init__(OptDirection.MIN, pmf, getFeasibleAction, stateTransition, immediateValue, functor=functor,
                         device=device, kernel=kernel)

    def getCacheValues(self):
        self._solve()
        out = {}
        for r in self.getOptTable():
            s = self.functor.make_state(int(r[0]), r[1], 0.0, r[2])
            out[s] = self.getExpectedValue(s)
        return out

    def _opt_columns(self, period, idx, q):
        x_lo, nx, nc, nq = self._engine.grid(period)
        iq, ix = idx // nx, idx % nx
        step = self.functor.stepSize
        rows = np.stack([np.full(len(idx), float(period)), x_lo + ix * step, iq * step, q], axis=1)
        order = np.lexsort((rows[:, 2], rows[:, 1]))  # by x, then preQ
        return rows[order]

    def _row_tuple(self, r):
        return (r[1], 0.0, r[2])

    def _row_of_state(self, s, action):
        return np.array([[float(s.getPeriod()), s.getIniInventory(), s.getPreQ(), action]])

    def _ncols(self):
        return 4


class CashRecursion(_GpuRecursionBase):
    """sdp.cash.CashRecursion (CashRecursion.java:39-56): discounted future term
    `p * discountFactor * V` (CashRecursion.java:120); rows {period, x, cash, Q} (:209-218)."""

    def __init__(self, optDirection, pmf, getFeasibleAction=None, stateTransition=None, immediateValue=None,
                 discountFactor: float = 1.0, *, functor=None, device: int = -1, kernel: int = 0):
        super().__init__(optDirection, pmf, getFeasibleAction, stateTransition, immediateValue, functor=functor,
                         discountFactor=discountFactor, device=device, kernel=kernel)

    def _opt_columns(self, period, idx, q):
        x_lo, nx, nc, nq = self._engine.grid(period)
        ix, ic = idx // nc, idx % nc
        cash = np.array([self._engine.cash_value(int(c)) for c in np.unique(ic)])
        cash_of = dict(zip(np.unique(ic).tolist(), cash.tolist()))
        return np.stack([np.full(len(idx), float(period)), x_lo + ix * self.functor.stepSize,
                         np.array([cash_of[int(c)] for c in ic]), q], axis=1)

    def _row_tuple(self, r):
        return (r[1], r[2])

    def _row_of_state(self, s, action):
        return np.array([[float(s.getPeriod()), s.getIniInventory(), s.getIniCash(), action]])

    def _ncols(self):
        return 4


class CashRecursionXR(CashRecursion):
    """sdp.cash.CashRecursionXR (CashRecursionXR.java:39-56,79-126): the discounted loop of CashRecursion on
    CashStateXR keys (period, x, R), R = cash + variCost * x; the actions ARE order-up-to levels, so getAction
    returns y (`bestY`, :96,:103-113).  Rows of getOptTable are {period, x, S, R, y} with S = R - unitVariCost * x
    (:185-200).  Functor: CashXRFunctor (the lambdas of cash.singleItem.CashConstraintXR)."""

    def getAction(self, state) -> float:
        return state.getIniInventory() + self._lookup(state)[1] * self.functor.stepSize

    def _opt_columns(self, period, idx, q):
        x_lo, nx, nc, nq = self._engine.grid(period)
        ix, ic = idx // nc, idx % nc
        x = x_lo + ix * self.functor.stepSize
        cash_of = {int(c): self._engine.cash_value(int(c)) for c in np.unique(ic)}
        cash = np.array([cash_of[int(c)] for c in ic])
        R = cash + self.functor.variCost * x  # what the transition stores (CashConstraintXR.java:121)
        return np.stack([np.full(len(idx), float(period)), x, R - self.functor.variCost * x, R, x + q], axis=1)

    def _row_tuple(self, r):
        return (r[1], r[3])

    def _row_of_state(self, s, action):
        return np.array([[float(s.getPeriod()), s.getIniInventory(), s.getIniR() - self.functor.variCost * s.getIniInventory(),
                          s.getIniR(), action]])

    def _ncols(self):
        return 5

    def getCacheActions(self):
        return {self.functor.make_state(int(r[0]), r[1], r[3]): float(r[4]) for r in self.getOptTable()}


class RiskRecursion(CashRecursion):
    """sdp.cash.RiskRecursion (RiskRecursion.java:31-46): the survival-probability recursion, MAX only, no
    discount; `getSurvProb` (:65-108) takes the place of getExpectedValue.  Rows of getOptTable are
    {period, x, cash, bankruptBefore, Q} (:123-132; the flag is always 0, RiskState.java:17)."""

    def __init__(self, pmf, getFeasibleAction=None, stateTransition=None, immediateValue=None, *, functor=None,
                 device: int = -1, kernel: int = 0):
        super().__init__(OptDirection.MAX, pmf, getFeasibleAction, stateTransition, immediateValue, 1.0,
                         functor=functor, device=device, kernel=kernel)

    def getSurvProb(self, state) -> float:
        return self._lookup(state)[0]

    def getExpectedValue(self, state):
        raise AttributeError("RiskRecursion has getSurvProb, not getExpectedValue (RiskRecursion.java:65)")

    def _opt_columns(self, period, idx, q):
        c = super()._opt_columns(period, idx, q)
        return np.concatenate([c[:, :3], np.zeros((len(idx), 1)), c[:, 3:]], axis=1)

    def _row_tuple(self, r):
        return (r[1], r[2])

    def _row_of_state(self, s, action):
        return np.array([[float(s.getPeriod()), s.getIniInventory(), s.getIniCash(), 0.0, action]])

    def _ncols(self):
        return 5


class CashLeadtimeRecursion(_GpuRecursionBase):
    """sdp.cash.CashLeadtimeRecursion (CashLeadtimeRecursion.java:28-46): MAX only, no discount;
    rows {period, x, cash, preQ, Q} (:97-106).  The reference's comparator (:37-41) is malformed
    (its cash test compares a value with itself); this mirror keys on the full tuple."""

    def __init__(self, pmf, getFeasibleAction=None, stateTransition=None, immediateValue=None, *, functor=None,
                 device: int = -1, kernel: int = 0):
        super().__init__(OptDirection.MAX, pmf, getFeasibleAction, stateTransition, immediateValue, functor=functor,
                         device=device, kernel=kernel)

    def getCacheValues(self):
        self._solve()
        out = {}
        for r in self.getOptTable():
            s = self.functor.make_state(int(r[0]), r[1], r[2], r[3])
            out[s] = self.getExpectedValue(s)
        return out

    def _opt_columns(self, period, idx, q):
        x_lo, nx, nc, nq = self._engine.grid(period)
        ic = idx % nc
        r = idx // nc
        ix, iq = r % nx, r // nx
        step = self.functor.stepSize
        ucs = np.unique(ic)
        cash_of = {int(c): self._engine.cash_value(int(c)) for c in ucs}
        rows = np.stack([np.full(len(idx), float(period)), x_lo + ix * step,
                         np.array([cash_of[int(c)] for c in ic]), iq * step, q], axis=1)
        order = np.lexsort((rows[:, 2], rows[:, 3], rows[:, 1]))  # by x, preQ, cash
        return rows[order]

    def _row_tuple(self, r):
        return (r[1], r[2], r[3])

    def _row_of_state(self, s, action):
        return np.array([[float(s.getPeriod()), s.getIniInventory(), s.getIniCash(), s.getPreQ(), action]])

    def _ncols(self):
        return 5

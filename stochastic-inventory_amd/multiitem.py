"""Two-product lead-time family on the reachable-set engine (sdpgpu_multilead_solve).

Host mirror of `sdp.cash.multiItem.CashRecursionMultiLead` as `cash.overdraft.MultiProductLeadtime.main`
uses it (src/cash/overdraft/MultiProductLeadtime.java:87-239): the lambdas there are fixed in form, so the
mirror takes their parameters, not closures.  This is the only family the reference records outputs
for (the comment block at MultiProductLeadtime.java:30-50).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Sequence

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

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
from ._abi import SdpgpuError, SdpgpuMultilead


@dataclass
class MultiLeadResult:
    finalValue: float          # iniCash + getExpectedValue(iniState)
    firstAction: int           # getAction(iniState).getFirstAction()
    secondAction: int
    statesPerPeriod: List[int]
    cells: int
    gpu_ms: float


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


def multilead_solve(**kw) -> MultiLeadResult:
    """One call = `recursion.getExpectedValue(iniState)` + `getAction(iniState)` of MultiProductLeadtime.main."""
    lib = _abi.load()
    k = fill_multilead(SdpgpuMultilead(), **kw)
    fv, q1, q2 = C.c_double(), C.c_int32(), C.c_int32()
    states = (C.c_int64 * k.T)()
    cells, ms = C.c_int64(), C.c_double()
    rc = lib.sdpgpu_multilead_solve(C.byref(k), C.byref(fv), C.byref(q1), C.byref(q2), states, C.byref(cells),
                                    C.byref(ms))
    if rc:
        raise SdpgpuError(rc, lib.sdpgpu_multilead_last_error().decode())
    return MultiLeadResult(fv.value, q1.value, q2.value, list(states), cells.value, ms.value)

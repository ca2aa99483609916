"""User-defined lambdas (sdpgpu_create_custom / oracle custom_functor), CPU side: the hipRTC compile needs no GPU,
and the oracle running the host-compiled text must reproduce its own built-in families."""
import ctypes as C

import numpy as np
import pytest

import cases
import custom_sources as cs


def _params_backorder(f):
    return [f.fixedOrderingCost, f.variOrderingCost, f.holdingCost, f.penaltyCost, f.minInventory, f.maxInventory,
            f.maxOrderQuantity]


@pytest.mark.parametrize("make,src", [(cases.f1_small, cs.BACKORDER), (cases.f1_max, cs.BACKORDER),
                                      (cases.f2_clamped, cs.LEADTIME)], ids=["f1_small", "f1_max", "f2_clamped"])
def test_oracle_with_user_lambdas_equals_builtin_family(oracle, make, src):
    w = make()
    P = oracle.Problem(w.desc(), w.pmf)
    V, pol, cells = P.solve()
    with oracle.custom_functor(src, _params_backorder(w.functor)):
        V2, pol2, cells2 = P.solve()
        m2 = P.memo()
    m = P.memo()
    assert cells == cells2 and m["value"] == m2["value"] and m["n"] == m2["n"]
    for a, b in zip(V + pol, V2 + pol2):
        assert np.array_equal(a, b)


def test_create_custom_compiles_without_a_gpu_and_reports_compile_errors(sia):
    lib = sia._abi.load()
    w = cases.f1_small()
    h = C.c_void_p()
    prm = (C.c_double * 7)(*_params_backorder(w.functor))
    d = w.desc()
    assert lib.sdpgpu_create_custom(C.byref(d), cs.BACKORDER.encode(), prm, 7, C.byref(h)) == 0
    assert h.value  # compiled; the code object is loaded when the first period runs
    lib.sdpgpu_destroy(h)
    bad = cs.BACKORDER.replace("double fixedCost", "double fixedCost = nonsense(); double fixedCost2")
    assert lib.sdpgpu_create_custom(C.byref(d), bad.encode(), prm, 7, C.byref(h)) == 1
    msg = lib.sdpgpu_last_error(None).decode()
    assert "does not compile" in msg and "nonsense" in msg and "user_functor" in msg
    d.lead_time = 2
    d.family = 2
    assert lib.sdpgpu_create_custom(C.byref(d), cs.LEADTIME.encode(), prm, 7, C.byref(h)) == 4

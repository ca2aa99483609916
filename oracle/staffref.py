"""ctypes binding of the CPU oracle for workforce.StaffRecursion (oracle/staffref.c).  TEST INFRASTRUCTURE ONLY:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDPREF_SANITIZE=1: the AddressSanitizer + UBSan build of the same file (oracle/Makefile); the process must then have
# gcc's libasan preloaded (tests/test_sanitizers.py runs the oracle tests that way)
_SAN = os.environ.get("SDPREF_SANITIZE") == "1"
LIB_NAME = "libstaffref_asan.so" if _SAN else "libstaffref.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)
_DP, _IP, _LP = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class _Problem(C.Structure):
    _fields_ = [("T", C.c_int32), ("min_x", C.c_int32), ("max_x", C.c_int32), ("clamp", C.c_int32),
                ("ini_x", C.c_int32), ("max_hire", C.c_int32),
                ("fix_cost", C.c_double), ("unit_vari_cost", C.c_double), ("salary", C.c_double),
                ("unit_penalty", C.c_double), ("min_staff", _IP), ("n_rows", C.c_int32), ("row_stride", C.c_int32),
                ("prob", _DP), ("row_len", _IP)]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("staffref.c", "staffref.h", "Makefile")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(f) > os.path.getmtime(LIB_PATH) for f in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", LIB_NAME], check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


class Problem:
    """prob: (T, n_rows, row_stride) float64, prob[t, y, j] = P(turnover j | hire-up-to level y) in period t+1."""

    def __init__(self, *, T, min_x, max_x, clamp, ini_x, max_hire, fix_cost, unit_vari_cost, salary, unit_penalty,
                 min_staff, prob, row_len=None):
        self.prob = np.ascontiguousarray(prob, dtype=np.float64)
        assert self.prob.ndim == 3 and self.prob.shape[0] == T
        self.min_staff = np.ascontiguousarray(min_staff, dtype=np.int32)
        assert len(self.min_staff) == T
        self.row_len = None if row_len is None else np.ascontiguousarray(row_len, dtype=np.int32)
        self.T = T
        self.c = _Problem(T, min_x, max_x, int(bool(clamp)), ini_x, max_hire, fix_cost, unit_vari_cost, salary,
                          unit_penalty, self.min_staff.ctypes.data_as(_IP), self.prob.shape[1], self.prob.shape[2],
                          self.prob.ctypes.data_as(_DP),
                          None if self.row_len is None else self.row_len.ctypes.data_as(_IP))
        x_lo = np.zeros(T, dtype=np.int32)
        nx = np.zeros(T, dtype=np.int32)
        if lib().staffref_layout(C.byref(self.c), x_lo.ctypes.data_as(_IP), nx.ctypes.data_as(_IP)):
            raise ValueError("staffref_layout rejected the problem")
        self.x_lo, self.nx = x_lo, nx
        self.off = np.concatenate([[0], np.cumsum(nx)]).astype(np.int64)

    def solve(self, nthreads: int = 1):
        """-> (V[t] arrays, policy[t] arrays, cells)"""
        values = np.zeros(int(self.off[-1]))
        policy = np.zeros(int(self.off[-1]), dtype=np.int32)
        cells = C.c_int64(0)
        rc = lib().staffref_solve(C.byref(self.c), values.ctypes.data_as(_DP), policy.ctypes.data_as(_IP),
                                  self.off.ctypes.data_as(_LP), nthreads, C.byref(cells))
        if rc:
            raise RuntimeError(f"staffref_solve: {rc}")
        V = [values[self.off[t]: self.off[t + 1]] for t in range(self.T)]
        P = [policy[self.off[t]: self.off[t + 1]] for t in range(self.T)]
        return V, P, cells.value

    def period(self, period, v_next, lo, hi, nthreads, v_cur, pol):
        cells = C.c_int64(0)
        rc = lib().staffref_period(C.byref(self.c), period, None if v_next is None else v_next.ctypes.data_as(_DP),
                                   v_cur.ctypes.data_as(_DP), pol.ctypes.data_as(_IP), C.c_int64(lo), C.c_int64(hi),
                                   nthreads, C.byref(cells))
        if rc:
            raise RuntimeError(f"staffref_period: {rc}")
        return cells.value

    def memo(self, width: int):
        """The literal recursion from (1, ini_x) -> (root value, root action, val[T, width], act, seen, cells)."""
        val = np.zeros((self.T, width))
        act = np.zeros((self.T, width), dtype=np.int32)
        seen = np.zeros((self.T, width), dtype=np.uint8)
        rv, ra, cells = C.c_double(), C.c_int32(), C.c_int64()
        rc = lib().staffref_memo(C.byref(self.c), C.byref(rv), C.byref(ra), width, val.ctypes.data_as(_DP),
                                 act.ctypes.data_as(_IP), seen.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(cells))
        if rc:
            raise RuntimeError(f"staffref_memo: {rc}")
        return rv.value, ra.value, val, act, seen.astype(bool), cells.value

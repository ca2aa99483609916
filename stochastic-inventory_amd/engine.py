"""Thin object wrapper over one sdpgpu_handle (include/sdpgpu.h).

`pmf` follows the reference's `double[][][] pmf` (Recursion.java:38): pmf[t][j] = [demand, prob].
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _abi
from ._abi import SdpgpuDesc, SdpgpuError, SdpgpuStats


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def split_pmf(pmf):
    """pmf[t][j] = (demand, prob) -> list of (demand array, prob array) per period."""
    out = []
    for tile in pmf:
        arr = np.asarray(tile, dtype=np.float64)
        if arr.ndim != 2 or arr.shape[1] != 2 or arr.shape[0] < 1:
            raise ValueError("pmf[t] must be an array of [demand, prob] pairs")
        out.append((np.ascontiguousarray(arr[:, 0]), np.ascontiguousarray(arr[:, 1])))
    return out


class SdpEngine:
    """One backward-recursion problem on one GPU (or one rank's state slab of it)."""

    def __init__(self, desc: SdpgpuDesc, pmf, overhead: Optional[Sequence[float]] = None,
                 custom_source: Optional[str] = None, custom_params: Optional[Sequence[float]] = None,
                 level_pmf=None, level_row_len=None):
        """custom_source: HIP device text of the three lambdas (see sdpgpu_create_custom in include/sdpgpu.h),
        custom_params: the doubles they read through `c.params`.
        level_pmf (STAFF family, instead of pmf): array (T, rows, stride), level_pmf[t, y, j] = P(turnover j | level y);
        level_row_len: entries per row (default y + 1).  `overhead` then carries minStaffNum[t]."""
        self._lib = _abi.load()
        self._h = C.c_void_p()
        self.desc = desc
        staff = desc.family == _abi.FAMILY_STAFF
        if staff:
            if level_pmf is None:
                raise ValueError("the STAFF family needs level_pmf")
            level_pmf = np.ascontiguousarray(level_pmf, dtype=np.float64)
            if level_pmf.ndim != 3 or level_pmf.shape[0] != desc.periods:
                raise ValueError("level_pmf must have shape (periods, rows, stride)")
            if level_row_len is not None:
                level_row_len = np.ascontiguousarray(level_row_len, dtype=np.int32)
                if level_row_len.shape != (level_pmf.shape[1],):
                    raise ValueError("level_row_len must have one entry per row")
        tiles = [] if staff else split_pmf(pmf)
        if not staff and len(tiles) != desc.periods:
            raise ValueError(f"pmf has {len(tiles)} periods, descriptor says {desc.periods}")
        if custom_source is None:
            rc = self._lib.sdpgpu_create(C.byref(desc), C.byref(self._h))
        else:
            prm = np.ascontiguousarray(custom_params if custom_params is not None else [], dtype=np.float64)
            rc = self._lib.sdpgpu_create_custom(C.byref(desc), custom_source.encode(), _dp(prm) if len(prm) else None,
                                                len(prm), C.byref(self._h))
        if rc:
            raise SdpgpuError(rc, self._lib.sdpgpu_last_error(None).decode())
        try:
            for t, (d, p) in enumerate(tiles):
                self._check(self._lib.sdpgpu_set_pmf(self._h, t, _dp(d), _dp(p), len(d)))
            if staff:
                for t in range(desc.periods):
                    self._check(self._lib.sdpgpu_set_level_pmf(
                        self._h, t, _dp(level_pmf[t]), None if level_row_len is None else _ip(level_row_len),
                        level_pmf.shape[1], level_pmf.shape[2]))
            if overhead is not None:
                for t, oh in enumerate(overhead):
                    self._check(self._lib.sdpgpu_set_overhead(self._h, t, float(oh)))
        except Exception:
            self.close()
            raise
        self.T = desc.periods

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc:
            raise SdpgpuError(rc, self._lib.sdpgpu_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.sdpgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- geometry ---------------------------------------------------------------------------
    def num_states(self, period: int) -> int:
        n = self._lib.sdpgpu_num_states(self._h, period)
        if n < 0:
            raise SdpgpuError(2, self._lib.sdpgpu_last_error(self._h).decode() or "layout not available")
        return int(n)

    def slab(self, period: int):
        pad, lo, hi = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.sdpgpu_slab(self._h, period, C.byref(pad), C.byref(lo), C.byref(hi)))
        return pad.value, lo.value, hi.value

    def grid(self, period: int):
        x_lo = C.c_double()
        nx, nc, nq = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.sdpgpu_grid(self._h, period, C.byref(x_lo), C.byref(nx), C.byref(nc), C.byref(nq)))
        return x_lo.value, nx.value, nc.value, nq.value

    def grid2(self, period: int):
        """(x_lo, nx, nc, nq1, nq2): the pipeline axis split in two (lead_time 2), nq2 = 1 otherwise."""
        x_lo = C.c_double()
        nx, nc, q1, q2 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.sdpgpu_grid2(self._h, period, C.byref(x_lo), C.byref(nx), C.byref(nc), C.byref(q1),
                                           C.byref(q2)))
        return x_lo.value, nx.value, nc.value, q1.value, q2.value

    def cash_value(self, ic: int) -> float:
        return float(self._lib.sdpgpu_cash_value(self._h, ic))

    def state_index(self, period: int, x: float, cash: float = 0.0, preq: float = 0.0, preq2: float = 0.0) -> int:
        return int(self._lib.sdpgpu_state_index2(self._h, period, float(x), float(cash), float(preq), float(preq2)))

    # -- execution --------------------------------------------------------------------------
    def set_action_counts(self, t: int, counts):
        """The caller's own action-list lengths of every grid state of period t+1 (sdpgpu_set_action_counts)."""
        c = np.ascontiguousarray(counts, dtype=np.int32)
        self._check(self._lib.sdpgpu_set_action_counts(self._h, t, _ip(c), len(c)))

    def set_stream(self, hip_stream: int):
        self._check(self._lib.sdpgpu_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_profiling(self, on: bool):
        self._check(self._lib.sdpgpu_set_profiling(self._h, 1 if on else 0))

    def solve(self, sync: bool = True):
        self._check(self._lib.sdpgpu_solve(self._h, 1 if sync else 0))

    def run_period(self, period: int):
        self._check(self._lib.sdpgpu_run_period(self._h, period))

    def keys_bytes(self) -> int:
        return int(self._lib.sdpgpu_keys_bytes(self._h))

    def attach_keys(self, device_ptr: int, nbytes: int):
        self._check(self._lib.sdpgpu_attach_keys(self._h, C.c_void_p(device_ptr), nbytes))

    def exchange_ptr(self, period: int) -> int:
        p = self._lib.sdpgpu_exchange_ptr(self._h, period)
        if not p:
            raise SdpgpuError(3, self._lib.sdpgpu_last_error(self._h).decode() or "no exchange table")
        return int(p)

    def finalize(self):
        """Enqueue the deferred read-out of every period run so far (no host wait)."""
        self._check(self._lib.sdpgpu_finalize(self._h))

    def footprint(self, period: int):
        """(left, right): state i of `period` reads V_{period+1}[i - left .. i + right]; None if unbounded."""
        l, r = C.c_int64(), C.c_int64()
        rc = self._lib.sdpgpu_footprint(self._h, period, C.byref(l), C.byref(r))
        if rc == 4:
            return None
        self._check(rc)
        return l.value, r.value

    def plan(self, period: int):
        """sdpgpu_plan_period: the launch plan of `period` (host arithmetic only; raises SdpgpuError when the plan -- e.g.
        one forced through SDPGPU_WIN_R / _S / _NCH -- cannot run)."""
        from ._abi import SdpgpuPlan
        out = SdpgpuPlan()
        self._check(self._lib.sdpgpu_plan_period(self._h, period, C.byref(out)))
        return out

    def set_halo(self, halo: int):
        self._check(self._lib.sdpgpu_set_halo(self._h, int(halo)))

    def run_period_range(self, period: int, lo: int, hi: int):
        self._check(self._lib.sdpgpu_run_period_range(self._h, period, int(lo), int(hi)))

    def run_period_part(self, period: int, part: int):
        self._check(self._lib.sdpgpu_run_period_part(self._h, period, part))

    # -- multi-GPU through the C ABI (csrc/sdpgpu_comm.hip) -------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: ONE rank calls this and hands the 128 bytes to the others."""
        lib = _abi.load()
        _abi.share_rccl_with_torch()
        buf = C.create_string_buffer(_abi.UNIQUE_ID_BYTES)
        rc = lib.sdpgpu_comm_unique_id(buf)
        if rc:
            raise SdpgpuError(rc, lib.sdpgpu_last_error(None).decode())
        return buf.raw

    def comm_prepare(self):
        """sdpgpu_comm_prepare: what can fail on this rank ALONE (device tables, RCCL load) -- before the ranks agree to
        enter the collective comm_init."""
        _abi.share_rccl_with_torch()
        self._check(self._lib.sdpgpu_comm_prepare(self._h))

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """Collective over the world's ranks (ncclCommInitRank on this handle's device)."""
        if len(unique_id) != _abi.UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        _abi.share_rccl_with_torch()
        self._check(self._lib.sdpgpu_comm_init(self._h, C.c_char_p(unique_id), rank, world))

    def comm_destroy(self):
        self._check(self._lib.sdpgpu_comm_destroy(self._h))

    def exchange(self, period: int):
        """All-gather of the row of `period` on the handle's stream (needs comm_init)."""
        self._check(self._lib.sdpgpu_exchange(self._h, period))

    def solve_sharded(self, overlap: bool = False, sync: bool = True, gather_first: bool = False):
        """This rank's whole sweep with the per-period all-gathers in between; every rank calls it."""
        flags = (_abi.SHARDED_SYNC if sync else 0) | (_abi.SHARDED_OVERLAP if overlap else 0) | \
                (_abi.SHARDED_GATHER_FIRST if gather_first else 0)
        self._check(self._lib.sdpgpu_solve_sharded(self._h, flags))

    @staticmethod
    def solve_multi(engines, sync: bool = True, gather_first: bool = False, threads: bool = False):
        """One process driving every rank: engines[r] is rank r of len(engines) (sdpgpu_solve_multi).  threads: one host
        thread per rank inside the library (SDPGPU_SHARDED_THREADS) instead of one thread issuing for all."""
        lib = _abi.load()
        _abi.share_rccl_with_torch()
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        flags = (_abi.SHARDED_SYNC if sync else 0) | (_abi.SHARDED_GATHER_FIRST if gather_first else 0) | \
                (_abi.SHARDED_THREADS if threads else 0)
        rc = lib.sdpgpu_solve_multi(arr, len(engines), flags)
        if rc:
            raise SdpgpuError(rc, lib.sdpgpu_last_error(engines[0]._h).decode())

    def synchronize(self):
        self._check(self._lib.sdpgpu_synchronize(self._h))

    def values_bytes(self) -> int:
        return int(self._lib.sdpgpu_values_bytes(self._h))

    def attach_values(self, device_ptr: int, nbytes: int):
        self._check(self._lib.sdpgpu_attach_values(self._h, C.c_void_p(device_ptr), nbytes))

    def values_device_ptr(self, period: int) -> int:
        p = self._lib.sdpgpu_values_device_ptr(self._h, period)
        if not p:
            raise SdpgpuError(3, self._lib.sdpgpu_last_error(self._h).decode() or "no device table")
        return int(p)

    # -- results ----------------------------------------------------------------------------
    def values(self, period: int) -> np.ndarray:
        n = self.num_states(period)
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.sdpgpu_values(self._h, period, _dp(out), n))
        return out

    def policy(self, period: int) -> np.ndarray:
        """Arg-opt action INDEX of this rank's slab (action = index * step)."""
        _, lo, hi = self.slab(period)
        out = np.empty(hi - lo, dtype=np.int32)
        self._check(self._lib.sdpgpu_policy(self._h, period, _ip(out), lo, hi - lo))
        return out

    def eval_states(self, period: int, x, cash=None, preq=None, preq2=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = len(x)
        cash_a = None if cash is None else np.ascontiguousarray(cash, dtype=np.float64)
        preq_a = None if preq is None else np.ascontiguousarray(preq, dtype=np.float64)
        preq2_a = None if preq2 is None else np.ascontiguousarray(preq2, dtype=np.float64)
        val = np.empty(n, dtype=np.float64)
        act = np.empty(n, dtype=np.int32)
        self._check(
            self._lib.sdpgpu_eval_states2(
                self._h, period, n, _dp(x), None if cash_a is None else _dp(cash_a),
                None if preq_a is None else _dp(preq_a), None if preq2_a is None else _dp(preq2_a), _dp(val),
                _ip(act)))
        return val, act

    def reachable(self, period: int) -> np.ndarray:
        n = self.num_states(period)
        out = np.empty(n, dtype=np.uint8)
        self._check(self._lib.sdpgpu_reachable(self._h, period, out.ctypes.data_as(C.POINTER(C.c_uint8)), n))
        return out.astype(bool)

    def simulate(self, demand, discount, ini_x: float, ini_cash: float = 0.0, ini_preq: float = 0.0):
        """Roll the policy along demand paths: demand[n][T] (rounded), discount[T] -> (sums[n], valid[n])."""
        dem = np.ascontiguousarray(demand, dtype=np.float64)
        if dem.ndim != 2 or dem.shape[1] != self.T:
            raise ValueError(f"demand must be [n_paths][{self.T}]")
        disc = np.ascontiguousarray(discount, dtype=np.float64)
        if disc.shape != (self.T,):
            raise ValueError("discount must have one entry per period")
        n = dem.shape[0]
        out = np.empty(n, dtype=np.float64)
        valid = np.empty(n, dtype=np.uint8)
        self._check(self._lib.sdpgpu_simulate(self._h, n, _dp(dem), _dp(disc), float(ini_x), float(ini_cash),
                                              float(ini_preq), _dp(out), valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        self.last_sim_flags = valid  # bit 0: path stayed on the grid; bit 1 (family SURVIVAL): a demand was lost
        return out, (valid & 1).astype(bool)

    def stats(self) -> SdpgpuStats:
        st = SdpgpuStats()
        self._check(self._lib.sdpgpu_stats_get(self._h, C.byref(st)))
        return st

    def period_ms(self, period: int) -> float:
        return float(self._lib.sdpgpu_period_ms(self._h, period))

    def period_cells(self, period: int) -> int:
        """Cells of `period` on this rank's slab (-1: not counted)."""
        return int(self._lib.sdpgpu_period_cells(self._h, period))

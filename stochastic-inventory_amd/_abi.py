"""ctypes binding of the C ABI declared in include/sdpgpu.h.

The shared library `libsdpgpu.so` (hand-written HIP for gfx950, built in-tree by
`build.py`) is the product.  There is no Python or CPU fallback: if the library is
missing or no HIP device is present every compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDPGPU_LIB: load another build of the SAME library instead (tests/test_sanitizers.py points it at the host-ASan build)
LIB_PATH = os.environ.get("SDPGPU_LIB") or os.path.join(_HERE, "libsdpgpu.so")

SDPGPU_ABI_VERSION = 6

FAMILY_BACKORDER = 1
FAMILY_LEADTIME = 2
FAMILY_CASH = 3
FAMILY_OVERDRAFT = 4
FAMILY_CASH_LEADTIME = 5
FAMILY_SURVIVAL = 6
FAMILY_STAFF = 7

MIN = 0
MAX = 1

KERNEL_AUTO = 0
KERNEL_GATHER = 1
KERNEL_WINDOW = 2
KERNEL_SEPARABLE = 3  # opt-in.  F1: reassociated sum, values to 1e-9, arg-opt may differ on near-ties; F2: exact

SHARDED_SYNC = 1
SHARDED_OVERLAP = 2
SHARDED_GATHER_FIRST = 4
SHARDED_THREADS = 8  # sdpgpu_solve_multi: one host thread per rank
UNIQUE_ID_BYTES = 128

PART_ALL = 0
PART_INTERIOR = 1
PART_BOUNDARY = 2


class SdpgpuDesc(C.Structure):
    """struct sdpgpu_desc (include/sdpgpu.h), field for field."""

    _fields_ = [
        ("abi_version", C.c_int32),
        ("family", C.c_int32),
        ("direction", C.c_int32),
        ("periods", C.c_int32),
        ("step", C.c_double),
        ("min_inventory", C.c_double),
        ("max_inventory", C.c_double),
        ("max_order_quantity", C.c_double),
        ("clamp_inventory", C.c_int32),
        ("zero_order_last_period", C.c_int32),
        ("ini_inventory", C.c_double),
        ("ini_cash", C.c_double),
        ("ini_preq", C.c_double),
        ("fixed_order_cost", C.c_double),
        ("unit_order_cost", C.c_double),
        ("holding_cost", C.c_double),
        ("penalty_cost", C.c_double),
        ("price", C.c_double),
        ("salvage_value", C.c_double),
        ("deposit_rate", C.c_double),
        ("overhead_cost", C.c_double),
        ("overhead_rate", C.c_double),
        ("discount_factor", C.c_double),
        ("min_cash", C.c_double),
        ("max_cash", C.c_double),
        ("cash_round_mult", C.c_double),
        ("cash_round_div", C.c_double),
        ("cash_round_int_div", C.c_int32),
        ("cash_formula", C.c_int32),
        ("r0", C.c_double),
        ("r2", C.c_double),
        ("r3", C.c_double),
        ("overdraft_limit", C.c_double),
        ("interest_free_amount", C.c_double),
        ("kernel", C.c_int32),
        ("device", C.c_int32),
        ("rank", C.c_int32),
        ("world_size", C.c_int32),
        ("store_all_values", C.c_int32),
        ("lead_time", C.c_int32),
        ("ini_preq2", C.c_double),
        ("reserved1", C.c_double),
    ]


class SdpgpuStats(C.Structure):
    _fields_ = [
        ("states_total", C.c_int64),
        ("cells_evaluated", C.c_int64),
        ("cells_all_ranks", C.c_int64),
        ("solve_ms", C.c_double),
        ("kernel_ms_sum", C.c_double),
        ("periods_run", C.c_int32),
        ("kernel_used", C.c_int32),
        ("window_r", C.c_int32),
        ("window_s", C.c_int32),
        ("fp64_ops_executed", C.c_double),
        ("lds_bytes", C.c_double),
        ("l1_bytes", C.c_double),
        ("graph_replays", C.c_int64),
    ]


class SdpgpuPlan(C.Structure):
    """struct sdpgpu_plan (include/sdpgpu.h, ABI 5)."""

    _fields_ = [("kernel", C.c_int32), ("r", C.c_int32), ("s", C.c_int32), ("chunks", C.c_int32),
                ("chunk_blocks", C.c_int32), ("tiles", C.c_int32), ("tasks", C.c_int32),
                ("workgroups_per_cu", C.c_int32), ("lds_bytes", C.c_int64)]


def desc_defaults() -> SdpgpuDesc:
    """Python twin of sdpgpu_desc_init (usable without loading the library)."""
    d = SdpgpuDesc()
    d.abi_version = SDPGPU_ABI_VERSION
    d.family = FAMILY_BACKORDER
    d.direction = MIN
    d.periods = 1
    d.step = 1.0
    d.clamp_inventory = 1
    d.discount_factor = 1.0
    d.cash_round_mult = 10.0
    d.cash_round_div = 10.0
    d.kernel = KERNEL_AUTO
    d.device = -1
    d.rank = 0
    d.world_size = 1
    d.store_all_values = 1
    return d


class SdpgpuMultilead(C.Structure):
    """struct sdpgpu_multilead (include/sdpgpu.h)."""

    _fields_ = [
        ("T", C.c_int32), ("q_bound", C.c_int32),
        ("price", C.c_double * 2), ("vari_cost", C.c_double * 2), ("sal_value", C.c_double * 2),
        ("ini_cash", C.c_double), ("ini_i1", C.c_double), ("ini_i2", C.c_double),
        ("r0", C.c_double), ("r1", C.c_double), ("r2", C.c_double), ("limit", C.c_double),
        ("interest_free", C.c_double),
        ("min_inventory", C.c_double), ("max_inventory", C.c_double), ("min_cash", C.c_double),
        ("max_cash", C.c_double), ("discount", C.c_double),
        ("overhead", C.c_double * 16),
        ("n1", C.c_int32), ("n2", C.c_int32),
        ("v1", C.c_double * 16), ("p1", C.c_double * 16), ("v2", C.c_double * 16), ("p2", C.c_double * 16),
        ("cash_int_cast", C.c_int32), ("reserved", C.c_int32),
    ]


class SdpgpuMulticash(C.Structure):
    """struct sdpgpu_multicash (include/sdpgpu.h)."""

    _fields_ = [
        ("T", C.c_int32), ("q_bound", C.c_int32),
        ("price", C.c_double * 2), ("vari_cost", C.c_double * 2), ("sal_price", C.c_double * 2),
        ("ini_cash", C.c_double), ("ini_i1", C.c_double), ("ini_i2", C.c_double),
        ("min_inventory", C.c_double), ("max_inventory", C.c_double), ("min_cash", C.c_double),
        ("max_cash", C.c_double), ("discount", C.c_double),
        ("pmf_off", C.POINTER(C.c_int32)), ("d1", C.POINTER(C.c_double)), ("d2", C.POINTER(C.c_double)),
        ("p", C.POINTER(C.c_double)),
    ]


class SdpgpuMultiTable(C.Structure):
    """struct sdpgpu_multi_table (include/sdpgpu.h)."""

    _fields_ = [("capacity", C.c_int64), ("rows", C.c_int64), ("period", C.POINTER(C.c_int32)),
                ("i1", C.POINTER(C.c_double)), ("i2", C.POINTER(C.c_double)), ("q1", C.POINTER(C.c_double)),
                ("q2", C.POINTER(C.c_double)), ("cash", C.POINTER(C.c_double)), ("value", C.POINTER(C.c_double)),
                ("a1", C.POINTER(C.c_int32)), ("a2", C.POINTER(C.c_int32))]


def make_multi_table(capacity: int):
    """-> (struct, dict of numpy arrays it points to)."""
    import numpy as np
    arrs = {"period": np.zeros(capacity, np.int32), "a1": np.zeros(capacity, np.int32), "a2": np.zeros(capacity, np.int32)}
    for n in ("i1", "i2", "q1", "q2", "cash", "value"):
        arrs[n] = np.zeros(capacity)
    t = SdpgpuMultiTable()
    t.capacity, t.rows = capacity, 0
    for n, a in arrs.items():
        setattr(t, n, a.ctypes.data_as(C.POINTER(C.c_int32 if a.dtype == np.int32 else C.c_double)))
    return t, arrs


def multi_table_rows(t, arrs):
    """The filled rows as an array sorted the way the reference's TreeMap orders its keys:
    columns period, i1, i2, q1, q2, cash, value, a1, a2."""
    import numpy as np
    n = int(t.rows)
    m = np.stack([arrs[c][:n].astype(np.float64) for c in ("period", "i1", "i2", "q1", "q2", "cash", "value", "a1", "a2")], axis=1)
    order = np.lexsort((m[:, 5], m[:, 4], m[:, 3], m[:, 2], m[:, 1], m[:, 0]))
    return m[order]


class SdpgpuDistSpec(C.Structure):
    """struct sdpgpu_dist_spec (include/sdpgpu.h)."""

    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("a", C.c_double), ("b", C.c_double)]


DIST_POISSON, DIST_NORMAL, DIST_UNIFORM_INT, DIST_GAMMA = 1, 2, 3, 4
PMF_GETPMF, PMF_CLSP = 0, 1


# every symbol include/sdpgpu.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)
_LP = C.POINTER(C.c_int64)
EXPORTS = {
    "sdpgpu_abi_version": (C.c_int, []),
    "sdpgpu_build_id": (C.c_char_p, []),
    "sdpgpu_desc_init": (None, [C.POINTER(SdpgpuDesc)]),
    "sdpgpu_create": (C.c_int, [C.POINTER(SdpgpuDesc), C.POINTER(_P)]),
    "sdpgpu_create_custom": (C.c_int, [C.POINTER(SdpgpuDesc), C.c_char_p, _DP, C.c_int32, C.POINTER(_P)]),
    "sdpgpu_destroy": (None, [_P]),
    "sdpgpu_last_error": (C.c_char_p, [_P]),
    "sdpgpu_set_pmf": (C.c_int, [_P, C.c_int32, _DP, _DP, C.c_int32]),
    "sdpgpu_set_level_pmf": (C.c_int, [_P, C.c_int32, _DP, _IP, C.c_int32, C.c_int32]),
    "sdpgpu_getpmf": (C.c_int, [C.POINTER(SdpgpuDistSpec), C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32, _DP, _DP,
                                C.c_int32, _IP]),
    "sdpgpu_set_overhead": (C.c_int, [_P, C.c_int32, C.c_double]),
    "sdpgpu_set_action_counts": (C.c_int, [_P, C.c_int32, _IP, C.c_int64]),
    "sdpgpu_set_stream": (C.c_int, [_P, _P]),
    "sdpgpu_set_profiling": (C.c_int, [_P, C.c_int32]),
    "sdpgpu_num_states": (C.c_int64, [_P, C.c_int32]),
    "sdpgpu_slab": (C.c_int, [_P, C.c_int32, _LP, _LP, _LP]),
    "sdpgpu_grid": (C.c_int, [_P, C.c_int32, _DP, _LP, _LP, _LP]),
    "sdpgpu_grid2": (C.c_int, [_P, C.c_int32, _DP, _LP, _LP, _LP, _LP]),
    "sdpgpu_cash_value": (C.c_double, [_P, C.c_int64]),
    "sdpgpu_state_index": (C.c_int64, [_P, C.c_int32, C.c_double, C.c_double, C.c_double]),
    "sdpgpu_state_index2": (C.c_int64, [_P, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]),
    "sdpgpu_solve": (C.c_int, [_P, C.c_int32]),
    "sdpgpu_run_period": (C.c_int, [_P, C.c_int32]),
    "sdpgpu_run_period_part": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "sdpgpu_footprint": (C.c_int, [_P, C.c_int32, _LP, _LP]),
    "sdpgpu_set_halo": (C.c_int, [_P, C.c_int64]),
    "sdpgpu_run_period_range": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64]),
    "sdpgpu_comm_unique_id": (C.c_int, [_P]),
    "sdpgpu_comm_prepare": (C.c_int, [_P]),
    "sdpgpu_comm_init": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "sdpgpu_comm_destroy": (C.c_int, [_P]),
    "sdpgpu_exchange": (C.c_int, [_P, C.c_int32]),
    "sdpgpu_solve_sharded": (C.c_int, [_P, C.c_int32]),
    "sdpgpu_solve_multi": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int32]),
    "sdpgpu_values_device_ptr": (_P, [_P, C.c_int32]),
    "sdpgpu_values_bytes": (C.c_size_t, [_P]),
    "sdpgpu_attach_values": (C.c_int, [_P, _P, C.c_size_t]),
    "sdpgpu_exchange_ptr": (_P, [_P, C.c_int32]),
    "sdpgpu_keys_bytes": (C.c_size_t, [_P]),
    "sdpgpu_attach_keys": (C.c_int, [_P, _P, C.c_size_t]),
    "sdpgpu_finalize": (C.c_int, [_P]),
    "sdpgpu_synchronize": (C.c_int, [_P]),
    "sdpgpu_values": (C.c_int, [_P, C.c_int32, _DP, C.c_int64]),
    "sdpgpu_policy": (C.c_int, [_P, C.c_int32, _IP, C.c_int64, C.c_int64]),
    "sdpgpu_eval_states": (C.c_int, [_P, C.c_int32, C.c_int64, _DP, _DP, _DP, _DP, _IP]),
    "sdpgpu_eval_states2": (C.c_int, [_P, C.c_int32, C.c_int64, _DP, _DP, _DP, _DP, _DP, _IP]),
    "sdpgpu_reachable": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_uint8), C.c_int64]),
    "sdpgpu_simulate": (C.c_int, [_P, C.c_int64, _DP, _DP, C.c_double, C.c_double, C.c_double, _DP,
                                  C.POINTER(C.c_uint8)]),
    "sdpgpu_stats_get": (C.c_int, [_P, C.POINTER(SdpgpuStats)]),
    "sdpgpu_period_ms": (C.c_double, [_P, C.c_int32]),
    "sdpgpu_period_cells": (C.c_int64, [_P, C.c_int32]),
    "sdpgpu_plan_period": (C.c_int, [_P, C.c_int32, C.POINTER(SdpgpuPlan)]),
    "sdpgpu_multilead_solve": (C.c_int, [C.POINTER(SdpgpuMultilead), _DP, _IP, _IP, _LP, _LP, _DP]),
    "sdpgpu_multilead_last_error": (C.c_char_p, []),
    "sdpgpu_multi_set_table": (None, [C.POINTER(SdpgpuMultiTable)]),
    "sdpgpu_multicash_solve": (C.c_int, [C.POINTER(SdpgpuMulticash), _DP, _IP, _IP, _LP, _LP, _DP]),
    "sdpgpu_multixr_solve": (C.c_int, [C.POINTER(SdpgpuMulticash), C.c_double, _DP, _IP, _IP, _LP, _LP, _DP]),
}

_lib = None


class SdpgpuError(RuntimeError):
    """A C-ABI call returned a non-zero status (message from sdpgpu_last_error)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"sdpgpu error {code}: {message}")
        self.code = code
        self.message = message


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  libsdpgpu.so needs `libamdhip64.so.7` (the system ROCm's); PyTorch's
    libraries need the unversioned `libamdhip64.so` and find their own bundled copy through RPATH.  If this
    library is loaded first the process ends up with two runtimes and the second one to initialise sees no
    GPU ("No HIP GPUs are available").  Loading PyTorch's copy first (when PyTorch is installed; it is not
    imported here) makes both resolve to the same object, because its SONAME is libamdhip64.so.7."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # already loaded: our DT_NEEDED resolves to it by SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime


def share_rccl_with_torch():
    """One RCCL (and one rocm_smi under it) per process: libsdpgpu.so opens `librccl.so.1` on first use of a communicator
    (sdpgpu_comm.hip) and takes a copy the process already holds.  When PyTorch is installed, PyTorch is IMPORTED here,
    so that its bundled ROCm libraries -- libamdhip64, librccl, librocm_smi64 -- come in in PyTorch's own order and
    the library then finds that RCCL.  (Loading PyTorch's librccl.so by hand BEFORE a later `import torch` changed the
    libraries' load order and with it the order of their exit-time destructors: librocm_smi64's global map was freed
    twice and the process aborted at exit, after every test had passed.)  Without PyTorch the system RCCL is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None:
        return
    try:
        import torch  # noqa: F401  (for its libraries only)
    except Exception:
        pass  # the system RCCL then


def load():
    """Load libsdpgpu.so and type every export.  Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
        )
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sdpgpu_abi_version() != SDPGPU_ABI_VERSION:
        raise ImportError("libsdpgpu.so ABI version mismatch")
    _lib = lib
    return lib

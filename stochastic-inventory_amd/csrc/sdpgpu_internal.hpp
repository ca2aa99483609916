// sdpgpu_internal.hpp -- what the translation units of libsdpgpu.so share: the handle, the per-period layout
// record, small host helpers and the functions one unit offers the others.  Not installed, not part of the ABI.
//
//   sdpgpu.hip          descriptor validation, grid layout, device tables, period dispatch, the C ABI
//   sdpgpu_generic.hip  generic per-cell kernel (every family), reachable set, rollout, user-defined functors
//   sdpgpu_window.hip   F1 window kernel (plan, chunk/key bookkeeping, finalize) and F2 row-window kernel
//   sdpgpu_cash.hip     F3 uniform-shift kernel and the cash row kernel (F3-F6)
//   sdpgpu_staff.hip    STAFF family (workforce.StaffRecursion): level-dependent pmf tables, its period kernel
//   sdpgpu_sparse.hip   reachable-set engine of the two-product lead-time family (own entry point)
//   sdpgpu_pmf.hip      GetPmf.getpmf / CLSP.main's inline pmf (host arithmetic) behind the ABI
//   sdpgpu_comm.hip     multi-GPU: RCCL communicators (loaded on first use), per-period all-gather, sharded sweeps
#pragma once
#include "../../include/sdpgpu.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <mutex>
#include <string>
#include <vector>

#include "sdp_device.hpp"

namespace sdp {
struct FinalizeJob;
struct QueryStates;
struct SimPeriod;
}  // namespace sdp

using sdp::DevParams;
using sdp::Grid;



constexpr size_t kPmfPad = 16;  // zero-probability tail: demand loop in blocks of R <= 8, one block of prefetch

// register block / chunking of the F1 window kernel for one period (sdpgpu_window.hip)
struct WinPlan {
  int R = 0, S = 1, d_pad = 0, n_chunks = 1, chunk_blocks = 0, n_tiles = 0, n_tasks = 0;
  size_t smem = 0;
  int tile_states() const { return 64 * S; }
};

// plan_window's answer for one period, kept per (slab, forced block): the search walks 8 block shapes x up to blocks_total
// chunk counts on the issuing thread, and launch_window, run_period_impl's pre-check and sdpgpu_plan_period all ask for it
// (at 30 us per period the second search sat on the critical path).  Dropped by layout().
struct WinPlanCache {
  bool valid = false;
  int64_t lo = 0, hi = 0;
  int win_r = 0, win_s = 0, win_nch = 0;
  bool may_chunk = false;
  WinPlan plan;
  std::string why;
};

struct PeriodInfo {
  Grid g{};
  int64_t S = 0;      // grid states
  int64_t S_pad = 0;  // padded to a multiple of world_size
  int64_t lo = 0, hi = 0;  // this rank's slab
  int32_t nD = 0;
  size_t pmf_off = 0;   // element offset of this period's demand array inside d_pmf
  // The window kernels index demands by j = (d - d_0) / step.  A support with gaps (DiscreteDistribution-style
  // {2, 5, 9}) is laid out on the unit-stride grid with probability 0 in the gaps: a zero-probability step adds
  // exact zeros to the accumulator, so the sums are unchanged.  nD_win = 0: too sparse, generic kernel instead.
  int32_t nD_win = 0;      // demand steps of the unit-stride layout
  size_t pmf_win_off = 0;  // element offset of its probability array (== pmf_off + nD when there are no gaps)
  size_t v_off = 0;     // element offset of V_t inside the value arena
  size_t pol_off = 0;   // element offset of this rank's policy slab
  double overhead = 0;
  bool overhead_set = false;
  int64_t cells_rank = 0, cells_all = 0;
  bool cells_counted = false;  // count_cells has run for this slab, overhead and action counts (reset by whatever changes them)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  int32_t kernel_used = 0;
  // executed fp64 add/mul operations per cell of the kernel plan that ran this period (identical operations of
  // neighbouring cells are formed once, see sdp_window.hpp / sdp_cash.hpp); 0 = the kernel has no such model
  double ops_cell = 0;
  double lds_cell = 0, l1_cell = 0;  // bytes per cell through the LDS / the vector L1 of the kernel that ran the period (0: no model)
  mutable WinPlanCache win_plan;     // (a cache: filled through const handles by plan_window)
};

struct sdpgpu_handle {
  sdpgpu_desc d{};
  int32_t T = 0;
  int32_t n_actions_full = 0;
  std::vector<PeriodInfo> per;  // index period-1
  std::vector<std::vector<double>> pmf_d, pmf_p;
  std::vector<char> pmf_set;
  bool laid_out = false;
  bool allocated = false;
  double* d_pmf = nullptr;
  double* d_values = nullptr;
  size_t values_elems = 0;
  bool values_external = false;
  int32_t* d_policy = nullptr;
  size_t policy_elems = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool stream_given = false;  // sdpgpu_set_stream was called (NULL then means the legacy default stream)
  bool profiling = false;
  hipEvent_t ev_solve0 = nullptr, ev_solve1 = nullptr;
  bool solve_timed = false;
  std::vector<char> period_done;  // V_t valid (a ping-pong table may have been overwritten since)
  std::vector<char> policy_done;  // the policy slab of period t has been computed
  double* d_part_val[2] = {nullptr, nullptr};  // window kernels: partial arg-opt rows [chunk][slab], by period parity
  int32_t* d_part_idx[2] = {nullptr, nullptr};
  size_t part_elems[2] = {0, 0};
  // F1 window kernel with several tasks per tile (small grids): V_t is reduced into order-preserving
  // keys by atomics and the (value, action) rows of the chunks are kept until flush_pending() turns
  // them into the final V_t / policy rows in ONE launch (see sdp_window.hpp finalize_kernel).
  unsigned long long* d_keys = nullptr;  // [T][key_stride]
  bool keys_external = false;            // caller memory (sdpgpu_attach_keys), e.g. a tensor RCCL can address
  size_t key_stride = 0;
  std::vector<char> key_row_clean;       // row t holds the reduction identity (+-Double.MAX_VALUE)
  double* d_chunk_val = nullptr;         // arena of chunk rows, period t at chunk_off[t-1]
  int32_t* d_chunk_idx = nullptr;
  std::vector<size_t> chunk_off;
  std::vector<int> pending_chunks;       // >0: period's final rows not written yet (value = n_chunks)
  int n_pending = 0;
  sdp::FinalizeJob* d_jobs = nullptr;
  std::vector<unsigned char> jobs_host;  // upload source of d_jobs: T FinalizeJobs at a fixed address (flush_pending)
  int32_t* d_rowperm = nullptr;  // cash row kernel, F5: rows of the launch ordered by level x + preQ (RowTiling::perm)
  std::vector<int32_t> rowperm_host;
  int64_t rowperm_key[4] = {-1, -1, -1, -1};
  int2* d_units = nullptr;       // F5 pair kernel: (row group, tile) units in diagonal order (RowTiling::units)
  int64_t units_key[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  char* d_rowtab = nullptr;  // cash_row_table_kernel's (row, action) blocks of the period being run (cash_row_pair_kernel<TAB>)
  size_t rowtab_bytes = 0;
  void* d_diag = nullptr;  // cash_diag_kernel: the DiagStep table of the period being run, and its padded p * gamma row
  size_t diag_bytes = 0;
  bool fuse_combine = true;
  bool use_cash_shift = true;
  bool use_cash_row = true;   // SDPGPU_CASH_ROW=0 turns the cash row kernel off (generic kernel instead)
  int win_prio_fair = 1;  // window kernel: s_setprio by progress (SDPGPU_WIN_PRIO=0 turns it off)
  int win_r = 0, win_nch = 0, win_s = 0;  // tuning overrides (SDPGPU_WIN_R / SDPGPU_WIN_NCH / SDPGPU_WIN_S), 0 = heuristic
  uint8_t* d_reach = nullptr;      // reachable masks, period t at reach_off[t-1]
  std::vector<size_t> reach_off;
  bool reach_done = false;
  // user-defined functor (sdpgpu_create_custom): code object compiled by hipRTC at create time, loaded at
  // first use; every period then runs sdp_custom_period instead of a built-in kernel
  bool custom = false;
  std::vector<char> custom_code;
  std::vector<double> custom_params;
  hipModule_t custom_mod = nullptr;
  hipFunction_t custom_period[3] = {nullptr, nullptr, nullptr};  // 64 / 16 / 4 states per workgroup
  hipFunction_t custom_reach = nullptr;
  // user lambdas of the LEVEL SHAPE (SDP_SHAPE_LEVEL in the text): the F1 window kernel runs them from per-period tables
  // M(m), c(a) filled once by the user's own compiled functions (sdp_custom_tabulate)
  bool level_shape = false;
  hipFunction_t custom_tabulate = nullptr;
  double* d_level_tabs = nullptr;             // per period: [m_tab (n_m)] [c_tab (n_a)]
  std::vector<size_t> level_tab_off;          // element offset of period t's m_tab
  std::vector<int32_t> level_m_min, level_m_n;
  double* d_custom_params = nullptr;
  unsigned long long* d_custom_cells = nullptr;  // [T]
  int* d_custom_err = nullptr;
  // sdpgpu_set_halo: a sharded caller may run a period on its slab widened by up to `halo` states on either side
  // (sdpgpu_run_period_range); the chunk-row arenas are sized for that
  int64_t halo = 0;
  // STAFF family (sdpgpu_set_level_pmf): per period the table transposed, lvl_p[t][j * rows + y], and its row lengths
  std::vector<std::vector<double>> lvl_p;
  std::vector<std::vector<int32_t>> lvl_len;
  std::vector<int32_t> lvl_rows, lvl_maxj;
  std::vector<double*> d_lvl_p;
  std::vector<int32_t*> d_lvl_len;
  std::vector<void*> staff_owned;   // the distinct device tables (periods given the same table share one)
  double* d_staff_val = nullptr;    // partial arg-min rows [group][slab]
  int32_t* d_staff_idx = nullptr;
  size_t staff_part_elems = 0;
  // sdpgpu_set_action_counts: per period the caller's action-list lengths of every grid state (empty: family rule)
  std::vector<std::vector<int32_t>> counts;
  std::vector<int32_t*> d_counts;
  double* d_sep_val = nullptr;      // opt-in separable mode, lead-time family: the table G[q2][y] and its arg-min
  int32_t* d_sep_idx = nullptr;
  size_t sep_elems = 0;
  // multi-GPU (sdpgpu_comm.hip): the communicator of this rank, a second stream for the overlapped schedule, and --
  // sdpgpu_solve_multi with several ranks on ONE device -- the sibling handles whose rows are exchanged by copies
  void* comm = nullptr;  // ncclComm_t
  bool comm_prepared = false;  // sdpgpu_comm_prepare has succeeded (tables allocated, device pinned, RCCL loads)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_comp = nullptr, ev_comm = nullptr;
  std::vector<sdpgpu_handle*> siblings;  // handles[0..n) of the last sdpgpu_solve_multi (set on every member)
  bool multi_copy = false;               // exchange by device-to-device copies (ranks share a device)
  // sdpgpu_solve as ONE HIP graph (small grids: a sweep of configs[1] is 52 launches of 30 us -- replayed from a graph the
  // sweep no longer depends on how promptly the host thread issues them).  0: no eager sweep yet; 1: one eager sweep done
  // (every lazy allocation, attribute and table upload has happened); 2: graph captured and in use; -1: off (capture refused
  // or SDPGPU_GRAPH=0).  Dropped by whatever changes what a sweep launches (graph_drop).
  int graph_state = 0;
  bool in_solve = false;
  hipGraph_t sweep_graph = nullptr;
  hipGraphExec_t sweep_exec = nullptr;
  int64_t graph_replays = 0;
  int flush_uploads = 0;  // job-list uploads of flush_pending since the counter was last reset (sdpgpu_solve's capture)
  std::string err;
  std::string plan_error;  // set by a launcher that rejects a period's plan (run_period_impl reports it as SDPGPU_ERR_ARG)
  int device = -1;
};

namespace sdpgpu_detail {

extern thread_local std::string g_create_error;
int fail(sdpgpu_handle* h, int code, const char* fmt, ...);

#define HIP_TRY(h, expr)                                                                        \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

inline bool has_cash(int f) {
  return f == SDPGPU_FAMILY_CASH || f == SDPGPU_FAMILY_OVERDRAFT || f == SDPGPU_FAMILY_CASH_LEADTIME ||
         f == SDPGPU_FAMILY_SURVIVAL;
}
inline bool has_preq(int f) { return f == SDPGPU_FAMILY_LEADTIME || f == SDPGPU_FAMILY_CASH_LEADTIME; }

// Java semantics needed on the host for the layout only.
inline int64_t java_round(double x) {
  double f = std::floor(x);
  return (int64_t)((x - f >= 0.5) ? f + 1.0 : f);
}
inline int32_t java_d2i(double x) {
  if (x != x) return 0;
  if (x >= 2147483647.0) return INT32_MAX;
  if (x <= -2147483648.0) return INT32_MIN;
  return (int32_t)x;
}

inline int64_t cash_key_of_bound(const sdpgpu_desc& d, double bound) {
  // key of the grid point the reference's rounding maps `bound` to
  int64_t r = java_round(bound * d.cash_round_mult);
  if (d.cash_round_int_div) return r / (int64_t)d.cash_round_div;
  return r;
}

inline bool is_pow2_int(double s) {
  if (!(s >= 1) || s != std::floor(s) || s > 1073741824.0) return false;
  int64_t v = (int64_t)s;
  return (v & (v - 1)) == 0;
}

// A dispatch carries at most 2^32 - 1 work-items (AQL grid_size is 32 bits); beyond that the launch is
// silently truncated.  Every launcher below sends 256-thread workgroups and refuses a grid over the limit.
inline bool grid_ok(int64_t blocks) { return blocks > 0 && blocks * 256 < 4294967296LL; }

// LDS of a gfx950 compute unit: 160 KiB, all of which one workgroup may take.  A launch that asks for more than the
// 64 KiB of earlier parts raises the kernel's dynamic-LDS limit first (lds_allow); planners budget against kLdsPerCU
// divided by the workgroups they want resident together on a CU.
constexpr size_t kLdsPerCU = 160 * 1024;
constexpr size_t kLdsLegacy = 64 * 1024;
inline int lds_workgroups(size_t smem) { return smem == 0 ? 32 : (int)(kLdsPerCU / smem); }

// Allow a kernel more than 64 KiB of dynamic LDS.  The attribute belongs to the function on the current device: it is
// raised ONCE per kernel instantiation and device, straight to the whole 160 KiB of the CU, under a lock -- so the limit never
// decreases and two host threads (SDPGPU_SHARDED_THREADS with ranks sharing a device, slabs planning different sizes) cannot
// lower it under each other's launch.  `mark` is the caller's per-instantiation record.
struct LdsMark {
  std::mutex m;
  bool raised[64] = {};
};
template <class K>
inline hipError_t lds_allow(K kernel, size_t smem, LdsMark* mark) {
  if (smem <= kLdsLegacy) return hipSuccess;
  if (smem > kLdsPerCU) return hipErrorInvalidValue;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mark->m);
  bool* slot = (dev >= 0 && dev < 64) ? &mark->raised[dev] : nullptr;
  if (slot && *slot) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCU);
  if (e == hipSuccess && slot) *slot = true;
  return e;
}

// the handle evaluates the backorder family's loop shape on the F1 window kernel: the built-in family, or user lambdas that
// declared the level shape
inline bool f1_like(const sdpgpu_handle* h) { return !h->custom || h->level_shape; }

// ---- sdpgpu.hip ----------------------------------------------------------------------------------------
int run_period_impl(sdpgpu_handle* h, int period, int part = SDPGPU_PART_ALL, int64_t range_lo = -1, int64_t range_hi = -1);
void graph_drop(sdpgpu_handle* h);  // forget the captured sweep (the next sdpgpu_solve runs eagerly, the one after captures again)
void count_cells(sdpgpu_handle* h, int period);
int flush_api(sdpgpu_handle* h);
int layout(sdpgpu_handle* h);
int ensure_device(sdpgpu_handle* h);
int allocate(sdpgpu_handle* h);
DevParams make_params(const sdpgpu_handle* h, int period);

// ---- sdpgpu_comm.hip -----------------------------------------------------------------------------------
void comm_release(sdpgpu_handle* h);  // sdpgpu_destroy: communicator, second stream, events

// ---- sdpgpu_generic.hip ----------------------------------------------------------------------------------
hipError_t launch_gather_grid(const DevParams& P, const double* v_next, double* v_cur, int32_t* pol, const double* pmf_d,
                              const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st);
hipError_t launch_gather_query(const DevParams& P, const double* v_next, double* out_val, int32_t* out_act,
                               const double* pmf_d, const double* pmf_p, int64_t n, const double* x, const double* cash,
                               const double* preq, const double* preq2, hipStream_t st);
hipError_t launch_custom_period(sdpgpu_handle* h, int period, const double* v_next, double* v_cur, int32_t* pol,
                                int64_t lo, int64_t hi, const double* qx, const double* qcash, const double* qpreq,
                                bool count);
int custom_check(sdpgpu_handle* h);
int fill_level_tables(sdpgpu_handle* h);  // user lambdas of the level shape: M(m), c(a) of every period (sdp_custom_tabulate)
int compute_reachable(sdpgpu_handle* h);
hipError_t launch_simulate(sdpgpu_handle* h, const sdp::SimPeriod* d_per, const double* d_dem, const double* d_disc,
                           int64_t n_paths, int64_t idx0, double ini_x, double ini_cash, double ini_preq,
                           double ini_preq2, int first_k, double* d_sum, uint8_t* d_valid);

// ---- sdpgpu_staff.hip -----------------------------------------------------------------------------------
int staff_upload(sdpgpu_handle* h);
hipError_t launch_staff(sdpgpu_handle* h, int period, const double* v_next, double* v_cur, int32_t* pol, int64_t lo,
                        int64_t hi, hipStream_t st);
int64_t staff_cells(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi);
void staff_reach_intervals(const sdpgpu_handle* h, std::vector<int64_t>* lo_out, std::vector<int64_t>* hi_out);

// ---- sdpgpu_cash.hip ---------------------------------------------------------------------------------------
bool cash_shift_eligible(const sdpgpu_handle* h, int period);
hipError_t launch_cash_shift(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                             int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st);
bool cash_row_eligible(const sdpgpu_handle* h, int period);
size_t cash_row_lds(int nD, int tile_pts);
hipError_t launch_cash_row(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                           int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st, bool levels_only = false);
// F5, OPT-IN separable mode: the rows of one level x + preQ hold identical tables; launch_cash_row(levels_only) evaluates one
// representative row per level, this copies it to the level's other rows
hipError_t launch_level_fill(sdpgpu_handle* h, int period, double* v_cur, int32_t* pol, hipStream_t st);

// ---- sdpgpu_window.hip -------------------------------------------------------------------------------------
bool window_eligible(const sdpgpu_handle* h, int period);
// (why: receives the reason when no plan exists -- a forced plan that is infeasible, or a period too big for the LDS)
WinPlan plan_window(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi, std::string* why = nullptr);
hipError_t flush_pending(sdpgpu_handle* h);
bool window_interior_tiles(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi, int* first, int* count);
hipError_t launch_separable_f2(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                               int32_t* pol, const double* pd, const double* pp);
hipError_t launch_separable(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                            int32_t* pol, const double* pd, const double* pp, bool* too_big);
// geometry of a period's chunk rows: element stride between chunks and the state index of element 0
inline int64_t chunk_row_cap(const sdpgpu_handle* h, const PeriodInfo& p) { return (p.hi - p.lo) + 2 * h->halo; }
inline int64_t chunk_row_lo(const sdpgpu_handle* h, const PeriodInfo& p) { return p.lo - h->halo; }
hipError_t launch_window(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                         int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st,
                         int part);

}  // namespace sdpgpu_detail

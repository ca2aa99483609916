// sdpgpu.hip -- C-ABI implementation (include/sdpgpu.h) of the MI355X SDP engine.
//
// Host side: descriptor validation, per-period grid layout, device tables, launch logic.
// Device side: sdp_gather.hpp (generic functor + gather kernel), sdp_window.hpp (F1/F2
// LDS-window kernel).  gfx950 only; there is NO CPU fallback anywhere in this library --
// without a HIP device every compute entry point fails with SDPGPU_ERR_DEVICE.
#include "../../include/sdpgpu.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "sdp_device.hpp"
#include "sdp_gather.hpp"
#include "sdp_window.hpp"
#include "sdp_cash.hpp"
#include "sdp_custom_src.hpp"

using sdp::DevParams;
using sdp::Grid;

namespace {

thread_local std::string g_create_error;

constexpr size_t kPmfPad = 16;  // zero-probability tail: demand loop in blocks of R <= 8, one block of prefetch

struct PeriodInfo {
  Grid g{};
  int64_t S = 0;      // grid states
  int64_t S_pad = 0;  // padded to a multiple of world_size
  int64_t lo = 0, hi = 0;  // this rank's slab
  int32_t nD = 0;
  size_t pmf_off = 0;   // element offset of this period's demand array inside d_pmf
  size_t v_off = 0;     // element offset of V_t inside the value arena
  size_t pol_off = 0;   // element offset of this rank's policy slab
  double overhead = 0;
  bool overhead_set = false;
  int64_t cells_rank = 0, cells_all = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  int32_t kernel_used = 0;
};

}  // namespace

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
  hipFunction_t custom_period = nullptr, custom_reach = nullptr;
  double* d_custom_params = nullptr;
  unsigned long long* d_custom_cells = nullptr;  // [T]
  int* d_custom_err = nullptr;
  std::string err;
  int device = -1;
};

namespace {

int fail(sdpgpu_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h)
    h->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define HIP_TRY(h, expr)                                                                        \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

bool has_cash(int f) {
  return f == SDPGPU_FAMILY_CASH || f == SDPGPU_FAMILY_OVERDRAFT || f == SDPGPU_FAMILY_CASH_LEADTIME ||
         f == SDPGPU_FAMILY_SURVIVAL;
}
bool has_preq(int f) { return f == SDPGPU_FAMILY_LEADTIME || f == SDPGPU_FAMILY_CASH_LEADTIME; }

// Java semantics needed on the host for the layout only.
int64_t java_round(double x) {
  double f = std::floor(x);
  return (int64_t)((x - f >= 0.5) ? f + 1.0 : f);
}
int32_t java_d2i(double x) {
  if (x != x) return 0;
  if (x >= 2147483647.0) return INT32_MAX;
  if (x <= -2147483648.0) return INT32_MIN;
  return (int32_t)x;
}

int64_t cash_key_of_bound(const sdpgpu_desc& d, double bound) {
  // key of the grid point the reference's rounding maps `bound` to
  int64_t r = java_round(bound * d.cash_round_mult);
  if (d.cash_round_int_div) return r / (int64_t)d.cash_round_div;
  return r;
}

bool is_pow2_int(double s) {
  if (!(s >= 1) || s != std::floor(s) || s > 1073741824.0) return false;
  int64_t v = (int64_t)s;
  return (v & (v - 1)) == 0;
}

int validate(const sdpgpu_desc& d) {
  if (d.abi_version != SDPGPU_ABI_VERSION) return fail(nullptr, SDPGPU_ERR_ARG, "abi_version %d != %d", d.abi_version, SDPGPU_ABI_VERSION);
  if (d.family < 1 || d.family > 6) return fail(nullptr, SDPGPU_ERR_ARG, "unknown family %d", d.family);
  if (d.direction != SDPGPU_MIN && d.direction != SDPGPU_MAX) return fail(nullptr, SDPGPU_ERR_ARG, "bad direction %d", d.direction);
  if (d.periods < 1 || d.periods > 4096) return fail(nullptr, SDPGPU_ERR_ARG, "periods %d out of range", d.periods);
  if (!is_pow2_int(d.step)) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "step %g: only power-of-two integer step sizes are supported (every in-scope driver uses 1)", d.step);
  if (!(d.max_order_quantity >= 0) || d.max_order_quantity > 1e9) return fail(nullptr, SDPGPU_ERR_ARG, "max_order_quantity %g", d.max_order_quantity);
  if (d.world_size < 1 || d.rank < 0 || d.rank >= d.world_size) return fail(nullptr, SDPGPU_ERR_ARG, "rank %d / world_size %d", d.rank, d.world_size);
  if (d.clamp_inventory) {
    if (!(d.max_inventory >= d.min_inventory)) return fail(nullptr, SDPGPU_ERR_ARG, "max_inventory < min_inventory");
    if (std::fmod(d.min_inventory, d.step) != 0 || std::fmod(d.max_inventory, d.step) != 0) return fail(nullptr, SDPGPU_ERR_ARG, "inventory bounds must be multiples of step");
  } else {
    if (d.family != SDPGPU_FAMILY_BACKORDER && d.family != SDPGPU_FAMILY_LEADTIME) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "clamp_inventory = 0 only for the backorder / lead-time families");
    if (std::fmod(d.ini_inventory, d.step) != 0) return fail(nullptr, SDPGPU_ERR_ARG, "ini_inventory must be a multiple of step");
  }
  if (has_cash(d.family)) {
    if (!(d.max_cash >= d.min_cash)) return fail(nullptr, SDPGPU_ERR_ARG, "max_cash < min_cash");
    if (!(d.cash_round_mult > 0) || !(d.cash_round_div > 0)) return fail(nullptr, SDPGPU_ERR_ARG, "cash rounding factors must be positive");
    if (!d.cash_round_int_div && d.cash_round_mult != d.cash_round_div) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "Math.round(c*m)/d with m != d does not map onto a uniform grid");
    if (d.cash_round_int_div && d.cash_round_div != std::floor(d.cash_round_div)) return fail(nullptr, SDPGPU_ERR_ARG, "integer cash divisor must be integral");
    if (std::fabs(d.min_cash * d.cash_round_mult) > 2.0e9 || std::fabs(d.max_cash * d.cash_round_mult) > 2.0e9) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "cash keys exceed 32 bits");
  }
  if (d.lead_time < 0 || d.lead_time > 2) return fail(nullptr, SDPGPU_ERR_ARG, "lead_time %d (0/1 = the reference's lead time 1, 2 = two-stage pipeline)", d.lead_time);
  if (d.lead_time == 2) {
    if (d.family != SDPGPU_FAMILY_LEADTIME) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "lead_time 2 exists for the LEADTIME family only");
    if (!d.clamp_inventory) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "lead_time 2 needs clamp_inventory = 1");
  }
  if ((d.family == SDPGPU_FAMILY_LEADTIME) && d.direction != SDPGPU_MIN) return fail(nullptr, SDPGPU_ERR_ARG, "LeadtimeRecursion is MIN only (LeadtimeRecursion.java:52,66)");
  if ((d.family == SDPGPU_FAMILY_SURVIVAL) && d.direction != SDPGPU_MAX) return fail(nullptr, SDPGPU_ERR_ARG, "getSurvProb maximises (RiskRecursion.java:70,101)");
  if ((d.family == SDPGPU_FAMILY_CASH_LEADTIME) && d.direction != SDPGPU_MAX) return fail(nullptr, SDPGPU_ERR_ARG, "CashLeadtimeRecursion is MAX only (CashLeadtimeRecursion.java:53,70)");
  return SDPGPU_OK;
}

int32_t full_action_count(const sdpgpu_desc& d) {
  if (d.family == SDPGPU_FAMILY_OVERDRAFT || d.family == SDPGPU_FAMILY_CASH_LEADTIME) return java_d2i(d.max_order_quantity) + 1;
  return java_d2i(d.max_order_quantity / d.step) + 1;
}

// Per-period grids.  Clamped families: one fixed box.  Unclamped (Leadtime.java:65-66): the box
// of period t+1 is the image of the period-t box under every action and demand.
int layout(sdpgpu_handle* h) {
  if (h->laid_out) return SDPGPU_OK;
  const sdpgpu_desc& d = h->d;
  for (int t = 0; t < h->T; ++t)
    if (!h->pmf_set[t]) return fail(h, SDPGPU_ERR_STATE, "pmf of period %d not set", t + 1);
  int64_t nc = 1, k_lo = 0, nq = 1;
  if (has_cash(d.family)) {
    k_lo = cash_key_of_bound(d, d.min_cash);
    nc = cash_key_of_bound(d, d.max_cash) - k_lo + 1;
  }
  h->n_actions_full = full_action_count(d);
  if (has_preq(d.family)) nq = java_d2i(d.max_order_quantity / d.step) + 1;
  const int64_t nq1 = nq;
  if (d.lead_time == 2) nq = nq1 * nq1;  // (q1, q2): iq = iq2 * nq1 + iq1
  double lo = d.min_inventory, hi = d.max_inventory;
  if (!d.clamp_inventory) lo = hi = d.ini_inventory;
  size_t v_off = 0, pol_off = 0, pmf_off = 0;
  int64_t s_pad_max = 0;
  for (int t = 0; t < h->T; ++t) {
    PeriodInfo& p = h->per[t];
    p.g.x_lo = lo;
    p.g.nx = (int64_t)((hi - lo) / d.step) + 1;
    p.g.nc = nc;
    p.g.nq = nq;
    p.g.nq1 = nq1;
    p.g.k_lo = k_lo;
    if (p.g.nx >= 2147483647LL || nc >= 2147483647LL) return fail(h, SDPGPU_ERR_UNSUPPORTED, "axis longer than 2^31");
    p.S = p.g.nx * p.g.nc * p.g.nq;
    int64_t w = d.world_size;
    p.S_pad = (p.S + w - 1) / w * w;
    int64_t slab = p.S_pad / w;
    p.lo = std::min<int64_t>(p.S, slab * d.rank);
    p.hi = std::min<int64_t>(p.S, slab * (d.rank + 1));
    p.nD = (int32_t)h->pmf_d[t].size();
    p.pmf_off = pmf_off;
    pmf_off += 2 * (size_t)p.nD + kPmfPad;  // probabilities are followed by kPmfPad zeros (window kernel)
    p.v_off = v_off;
    p.pol_off = pol_off;
    pol_off += (size_t)slab;
    if (d.store_all_values) v_off += (size_t)p.S_pad;
    s_pad_max = std::max(s_pad_max, p.S_pad);
    if (!p.overhead_set) p.overhead = d.overhead_cost;
    if (!d.clamp_inventory) {
      double dmin = h->pmf_d[t][0], dmax = dmin;
      for (double v : h->pmf_d[t]) {
        dmin = std::min(dmin, v);
        dmax = std::max(dmax, v);
      }
      double qmax = (double)(h->n_actions_full - 1) * d.step;
      lo = lo - dmax;
      hi = hi + qmax - dmin;
    }
  }
  if (!d.store_all_values) {
    // two ping-pong tables: V_t lives in table (t & 1)
    for (int t = 0; t < h->T; ++t) h->per[t].v_off = (size_t)((t + 1) & 1) * (size_t)s_pad_max;
    v_off = 2 * (size_t)s_pad_max;
  }
  h->values_elems = v_off;
  h->policy_elems = pol_off;
  h->laid_out = true;
  return SDPGPU_OK;
}

// A dispatch carries at most 2^32 - 1 work-items (AQL grid_size is 32 bits); beyond that the launch is
// silently truncated.  Every launcher below sends 256-thread workgroups and refuses a grid over the limit.
inline bool grid_ok(int64_t blocks) { return blocks > 0 && blocks * 256 < 4294967296LL; }

int ensure_device(sdpgpu_handle* h) {
  if (h->device >= 0) HIP_TRY(h, hipSetDevice(h->device));
  return SDPGPU_OK;
}

int allocate(sdpgpu_handle* h) {
  if (h->allocated) return SDPGPU_OK;
  int rc = layout(h);
  if (rc) return rc;
  rc = ensure_device(h);
  if (rc) return rc;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(h, SDPGPU_ERR_DEVICE, "no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (!h->stream && !h->stream_given) {
    HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  if (!h->d_values) {
    HIP_TRY(h, hipMalloc((void**)&h->d_values, std::max<size_t>(h->values_elems, 1) * sizeof(double)));
    h->values_external = false;
  }
  HIP_TRY(h, hipMalloc((void**)&h->d_policy, std::max<size_t>(h->policy_elems, 1) * sizeof(int32_t)));
  size_t pmf_elems = 0;
  for (auto& p : h->per) pmf_elems += 2 * (size_t)p.nD + kPmfPad;
  HIP_TRY(h, hipMalloc((void**)&h->d_pmf, std::max<size_t>(pmf_elems, 1) * sizeof(double)));
  std::vector<double> host(pmf_elems, 0.0);
  for (int t = 0; t < h->T; ++t) {
    const PeriodInfo& p = h->per[t];
    std::memcpy(&host[p.pmf_off], h->pmf_d[t].data(), (size_t)p.nD * sizeof(double));
    std::memcpy(&host[p.pmf_off + p.nD], h->pmf_p[t].data(), (size_t)p.nD * sizeof(double));
  }
  HIP_TRY(h, hipMemcpy(h->d_pmf, host.data(), pmf_elems * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(h, hipEventCreate(&h->ev_solve0));
  HIP_TRY(h, hipEventCreate(&h->ev_solve1));
  if (h->custom) {
    HIP_TRY(h, hipModuleLoadData(&h->custom_mod, h->custom_code.data()));
    HIP_TRY(h, hipModuleGetFunction(&h->custom_period, h->custom_mod, "sdp_custom_period"));
    HIP_TRY(h, hipModuleGetFunction(&h->custom_reach, h->custom_mod, "sdp_custom_reach"));
    HIP_TRY(h, hipMalloc((void**)&h->d_custom_params, std::max<size_t>(h->custom_params.size(), 1) * sizeof(double)));
    if (!h->custom_params.empty())
      HIP_TRY(h, hipMemcpy(h->d_custom_params, h->custom_params.data(), h->custom_params.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc((void**)&h->d_custom_cells, (size_t)h->T * sizeof(unsigned long long)));
    HIP_TRY(h, hipMemset(h->d_custom_cells, 0, (size_t)h->T * sizeof(unsigned long long)));
    HIP_TRY(h, hipMalloc((void**)&h->d_custom_err, sizeof(int)));
    HIP_TRY(h, hipMemset(h->d_custom_err, 0, sizeof(int)));
  }
  h->allocated = true;
  return SDPGPU_OK;
}

// ---- user-defined functor: parameter block and launches ---------------------------------------------
sdp::CustomParams make_custom_params(const sdpgpu_handle* h, int period) {
  const sdpgpu_desc& d = h->d;
  const PeriodInfo& p = h->per[period - 1];
  sdp::CustomParams C{};
  C.has_cash = has_cash(d.family);
  C.has_preq = has_preq(d.family);
  C.maxdir = d.direction == SDPGPU_MAX;
  C.is_last = period == h->T;
  C.n_demand = p.nD;
  C.survival = d.family == SDPGPU_FAMILY_SURVIVAL;
  C.cash_int_div = d.cash_round_int_div;
  C.period = period;
  C.T = h->T;
  C.step = d.step;
  C.inv_step = 1.0 / d.step;
  const bool cash_loop = d.family == SDPGPU_FAMILY_CASH || d.family == SDPGPU_FAMILY_OVERDRAFT || d.family == SDPGPU_FAMILY_SURVIVAL;
  C.gamma = cash_loop ? d.discount_factor : 1.0;
  C.round_mult = d.cash_round_mult;
  C.round_div = d.cash_round_div;
  auto grid_of = [](const Grid& g) { return sdp::CustomGrid{g.x_lo, (long long)g.nx, (long long)g.nc, (long long)g.nq, (long long)g.k_lo}; };
  C.cur = grid_of(p.g);
  if (period < h->T) C.next = grid_of(h->per[period].g);
  C.user = h->d_custom_params;
  return C;
}

hipError_t launch_custom_period(sdpgpu_handle* h, int period, const double* v_next, double* v_cur, int32_t* pol,
                                int64_t lo, int64_t hi, const double* qx, const double* qcash, const double* qpreq,
                                bool count) {
  if (hi <= lo) return hipSuccess;
  const PeriodInfo& p = h->per[period - 1];
  sdp::CustomParams C = make_custom_params(h, period);
  const double* pd = h->d_pmf + p.pmf_off;
  const double* pp = pd + p.nD;
  long long llo = lo, lhi = hi;
  unsigned long long* cells = count ? h->d_custom_cells + (period - 1) : nullptr;
  if (count) {
    hipError_t e0 = hipMemsetAsync(cells, 0, sizeof(unsigned long long), h->stream);
    if (e0 != hipSuccess) return e0;
  }
  int* err = h->d_custom_err;
  void* args[] = {&C, &v_next, &v_cur, &pol, &pd, &pp, &llo, &lhi, &qx, &qcash, &qpreq, &cells, &err};
  const int64_t blocks = (hi - lo + 15) / 16;
  if (!grid_ok(blocks)) return hipErrorInvalidValue;
  const size_t smem = (size_t)p.nD * 16 + 4 * 16 * (sizeof(double) + sizeof(int));
  return hipModuleLaunchKernel(h->custom_period, (unsigned)blocks, 1, 1, 256, 1, 1, (unsigned)smem, h->stream, args, nullptr);
}

hipError_t launch_custom_reach(sdpgpu_handle* h, int period, const uint8_t* mcur, uint8_t* mnext, int64_t n,
                               const double* qx, const double* qcash, const double* qpreq) {
  if (n <= 0) return hipSuccess;
  const PeriodInfo& p = h->per[period - 1];
  sdp::CustomParams C = make_custom_params(h, period);
  const double* pd = h->d_pmf + p.pmf_off;
  long long ln = n;
  int* err = h->d_custom_err;
  void* args[] = {&C, &mcur, &mnext, &pd, &ln, &qx, &qcash, &qpreq, &err};
  const int64_t blocks = (n + 63) / 64;
  if (!grid_ok(blocks)) return hipErrorInvalidValue;
  return hipModuleLaunchKernel(h->custom_reach, (unsigned)blocks, 1, 1, 256, 1, 1, 0, h->stream, args, nullptr);
}

// After a synchronisation point: did a user transition return a state that is not a grid point?
int custom_check(sdpgpu_handle* h) {
  if (!h->custom || !h->d_custom_err) return SDPGPU_OK;
  int flag = 0;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(&flag, h->d_custom_err, sizeof flag, hipMemcpyDeviceToHost));
  if (flag) {
    (void)hipMemset(h->d_custom_err, 0, sizeof(int));
    return fail(h, SDPGPU_ERR_ARG,
                "user functor: sdp_transition returned a state that is not a grid point of the next period "
                "(the lambda must clamp and round as the descriptor says; results are invalid)");
  }
  return SDPGPU_OK;
}

DevParams make_params(const sdpgpu_handle* h, int period) {
  const sdpgpu_desc& d = h->d;
  const PeriodInfo& p = h->per[period - 1];
  DevParams P{};
  P.family = d.family;
  P.maxdir = d.direction == SDPGPU_MAX;
  P.is_last = period == h->T;
  P.n_demand = p.nD;
  P.clamp_inventory = d.clamp_inventory;
  P.cash_formula = d.cash_formula;
  P.cash_round_int_div = d.cash_round_int_div;
  P.n_actions_full = h->n_actions_full;
  P.lead2 = d.lead_time == 2;
  if (d.family == SDPGPU_FAMILY_CASH_LEADTIME && d.zero_order_last_period && period == h->T) P.n_actions_full = 1;
  P.step = d.step;
  P.inv_step = 1.0 / d.step;  // exact: step is a power of two
  P.min_inventory = d.min_inventory;
  P.max_inventory = d.max_inventory;
  P.max_order_quantity = d.max_order_quantity;
  P.K = d.fixed_order_cost;
  P.v = d.unit_order_cost;
  P.h = d.holding_cost;
  P.pi = d.penalty_cost;
  P.price = d.price;
  P.salvage = d.salvage_value;
  P.one_plus_deposit = 1 + d.deposit_rate;
  P.overhead = p.overhead;
  P.one_minus_overhead_rate = 1 - d.overhead_rate;
  // Recursion / LeadtimeRecursion / CashLeadtimeRecursion have no discount (p * V); p * 1.0 == p
  // exactly, so one code path serves both loop shapes.
  bool cash_loop = d.family == SDPGPU_FAMILY_CASH || d.family == SDPGPU_FAMILY_OVERDRAFT || d.family == SDPGPU_FAMILY_SURVIVAL;
  P.gamma = cash_loop ? d.discount_factor : 1.0;
  P.min_cash = d.min_cash;
  P.max_cash = d.max_cash;
  P.round_mult = d.cash_round_mult;
  P.round_div = d.cash_round_div;
  P.r0 = d.r0;
  P.r2 = d.r2;
  P.r3 = d.r3;
  P.limit = d.overdraft_limit;
  P.interest_free = d.interest_free_amount;
  P.cur = p.g;
  if (period < h->T) P.next = h->per[period].g;
  return P;
}

// ---- launch helpers --------------------------------------------------------------------------


template <int FAM, bool MAXDIR, int SX, bool QUERY>
hipError_t launch_gather_sx(const DevParams& P, const double* v_next, double* v_cur, int32_t* pol, const double* pmf_d,
                            const double* pmf_p, int64_t lo, int64_t hi, sdp::QueryStates q, hipStream_t st) {
  int64_t n = hi - lo;
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + SX - 1) / SX;
  if (!grid_ok(blocks)) return hipErrorInvalidValue;
  size_t smem = (size_t)P.n_demand * 16 + 4 * 64 * (sizeof(double) + sizeof(int));
  hipLaunchKernelGGL((sdp::gather_period_kernel<FAM, MAXDIR, SX, QUERY>), dim3((unsigned)blocks), dim3(256), smem, st, P,
                     v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, q);
  return hipGetLastError();
}

template <int FAM, bool MAXDIR, bool QUERY>
hipError_t launch_gather_dir(const DevParams& P, const double* v_next, double* v_cur, int32_t* pol, const double* pmf_d,
                             const double* pmf_p, int64_t lo, int64_t hi, sdp::QueryStates q, hipStream_t st) {
  // Enough workgroups to fill 256 CUs several times over: shrink the state tile (and widen the
  // action split) for small grids.
  int64_t n = hi - lo;
  if (n >= 64 * 2048) return launch_gather_sx<FAM, MAXDIR, 64, QUERY>(P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, q, st);
  if (n >= 16 * 1024) return launch_gather_sx<FAM, MAXDIR, 16, QUERY>(P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, q, st);
  return launch_gather_sx<FAM, MAXDIR, 4, QUERY>(P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, q, st);
}

template <bool QUERY>
hipError_t launch_gather(const DevParams& P, const double* v_next, double* v_cur, int32_t* pol, const double* pmf_d,
                         const double* pmf_p, int64_t lo, int64_t hi, sdp::QueryStates q, hipStream_t st) {
#define SDP_CASE(F)                                                                                        \
  case F:                                                                                                  \
    return P.maxdir ? launch_gather_dir<F, true, QUERY>(P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, q, st)  \
                    : launch_gather_dir<F, false, QUERY>(P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, q, st);
  switch (P.family) {
    SDP_CASE(sdp::FAM_BACKORDER)
    SDP_CASE(sdp::FAM_LEADTIME)
    SDP_CASE(sdp::FAM_CASH)
    SDP_CASE(sdp::FAM_OVERDRAFT)
    SDP_CASE(sdp::FAM_CASH_LEADTIME)
    SDP_CASE(sdp::FAM_SURVIVAL)
  }
#undef SDP_CASE
  return hipErrorInvalidValue;
}

// cells of one period: sum over states of nA(s) * D  (host arithmetic, no device work)
void count_cells(sdpgpu_handle* h, int period) {
  PeriodInfo& p = h->per[period - 1];
  if (h->custom) return;  // counted on the device (sdpgpu_stats_get)
  const sdpgpu_desc& d = h->d;
  int64_t nD = p.nD;
  auto range_cells = [&](int64_t lo, int64_t hi) -> int64_t {
    if (hi <= lo) return 0;
    if (d.family != SDPGPU_FAMILY_CASH && d.family != SDPGPU_FAMILY_SURVIVAL) {
      int64_t nA = h->n_actions_full;
      if (d.family == SDPGPU_FAMILY_CASH_LEADTIME && d.zero_order_last_period && period == h->T) nA = 1;
      return (hi - lo) * nA * nD;
    }
    // F3: nA depends on the cash index only; flat = ix * nc + ic
    int64_t nc = p.g.nc;
    std::vector<int64_t> pre((size_t)nc + 1, 0);
    for (int64_t ic = 0; ic < nc; ++ic) {
      double k = (double)(p.g.k_lo + ic);
      double cash = d.cash_round_int_div ? k : k / d.cash_round_div;
      double m = std::fmin(d.max_order_quantity, std::fmax(0.0, (cash - p.overhead - d.fixed_order_cost) / d.unit_order_cost));
      if (d.family == SDPGPU_FAMILY_SURVIVAL) m = std::fmax(std::fmin(cash / d.unit_order_cost, d.max_order_quantity), 0.0);
      int64_t nA = (int64_t)java_d2i(m) + 1;
      pre[(size_t)ic + 1] = pre[(size_t)ic] + nA;
    }
    auto upto = [&](int64_t idx) { return (idx / nc) * pre[(size_t)nc] + pre[(size_t)(idx % nc)]; };
    return (upto(hi) - upto(lo)) * nD;
  };
  p.cells_rank = range_cells(p.lo, p.hi);
  p.cells_all = range_cells(0, p.S);
}

bool window_eligible(const sdpgpu_handle* h, int period);
bool cash_shift_eligible(const sdpgpu_handle* h, int period);
bool cash_row_eligible(const sdpgpu_handle* h, int period);
hipError_t launch_cash_row(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                           int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st);
hipError_t flush_pending(sdpgpu_handle* h);
hipError_t launch_cash_shift(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                             int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st);
hipError_t launch_window(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                         int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st,
                         int part);
bool window_interior_tiles(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi, int* first, int* count);

// part: SDPGPU_PART_ALL, or the two halves a sharded caller overlaps with the all-gather of V_{t+1}:
// INTERIOR = the states whose cells read only THIS rank's slab of V_{t+1}, BOUNDARY = the rest.
int run_period_impl(sdpgpu_handle* h, int period, int part = SDPGPU_PART_ALL) {
  int rc = allocate(h);
  if (rc) return rc;
  if (period < 1 || period > h->T) return fail(h, SDPGPU_ERR_ARG, "period %d out of 1..%d", period, h->T);
  if (period < h->T && !h->period_done[period]) return fail(h, SDPGPU_ERR_STATE, "V_%d has not been computed yet (periods run T..1)", period + 1);
  rc = ensure_device(h);
  if (rc) return rc;
  PeriodInfo& p = h->per[period - 1];
  DevParams P = make_params(h, period);
  const double* v_next = period < h->T ? h->d_values + h->per[period].v_off : nullptr;
  double* v_cur = h->d_values + p.v_off;
  int32_t* pol = h->d_policy + p.pol_off - p.lo;  // kernels index the policy by flat state index
  const double* pd = h->d_pmf + p.pmf_off;
  const double* pp = pd + p.nD;
  if (p.nD > 4000) return fail(h, SDPGPU_ERR_UNSUPPORTED, "pmf with %d points exceeds the LDS tile", p.nD);
  if (h->profiling) {
    if (!p.ev0) {
      HIP_TRY(h, hipEventCreate(&p.ev0));
      HIP_TRY(h, hipEventCreate(&p.ev1));
    }
    HIP_TRY(h, hipEventRecord(p.ev0, h->stream));
  }
  if (h->custom) {
    if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;  // no bounded footprint is known for user lambdas
    hipError_t ec = launch_custom_period(h, period, v_next, v_cur, pol, p.lo, p.hi, nullptr, nullptr, nullptr, true);
    if (ec != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "period %d user-functor kernel: %s", period, hipGetErrorString(ec));
    p.kernel_used = SDPGPU_KERNEL_GATHER;
    if (h->profiling) {
      HIP_TRY(h, hipEventRecord(p.ev1, h->stream));
      p.timed = true;
    } else {
      p.timed = false;
    }
    h->period_done[period - 1] = 1;
    h->policy_done[period - 1] = 1;
    if (!h->d.store_all_values && period + 2 <= h->T) h->period_done[period + 1] = 0;
    return SDPGPU_OK;
  }
  if (h->d.kernel == SDPGPU_KERNEL_SEPARABLE) {
    if (h->d.family != SDPGPU_FAMILY_BACKORDER) return fail(h, SDPGPU_ERR_UNSUPPORTED, "the separable mode exists for the backorder family only");
    if (h->n_actions_full > 6000) return fail(h, SDPGPU_ERR_UNSUPPORTED, "separable mode: action range exceeds the LDS tile");
    if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;
    hipError_t es = flush_pending(h);
    if (es == hipSuccess) {
      sdp::SepParams S{};
      S.x_lo = p.g.x_lo;
      S.step = h->d.step;
      S.h = h->d.holding_cost;
      S.pi = h->d.penalty_cost;
      S.K = h->d.fixed_order_cost;
      S.v = h->d.unit_order_cost;
      S.inv_step = 1.0 / h->d.step;
      S.min_inventory = h->d.min_inventory;
      S.max_inventory = h->d.max_inventory;
      S.clamp_inventory = h->d.clamp_inventory;
      S.n_actions = h->n_actions_full;
      S.n_demand = p.nD;
      S.d_min = h->pmf_d[period - 1].front();  // demands are strictly ascending (checked at set_pmf)
      S.d_range = (int32_t)((h->pmf_d[period - 1].back() - S.d_min) / h->d.step);
      const bool future = period < h->T;
      if (future) {
        S.next_x_lo = h->per[period].g.x_lo;
        S.next_last = (int32_t)(h->per[period].g.nx - 1);
      }
      const int64_t n = p.hi - p.lo;
      if (n > 0) {
        dim3 grid((unsigned)((n + 63) / 64));
        size_t smem = (size_t)(64 + S.n_actions + S.d_range) * 16 + (size_t)(64 + S.n_actions) * 8 +
                      4 * 64 * (sizeof(double) + sizeof(int));
        if (smem > 64 * 1024) return fail(h, SDPGPU_ERR_UNSUPPORTED, "separable mode: action + demand range exceeds the LDS tile");
        const bool mx = P.maxdir != 0;
        if (mx && future) hipLaunchKernelGGL((sdp::separable_f1_kernel<true, true>), grid, dim3(256), smem, h->stream, S, v_next, v_cur, pol, pd, pp, p.lo, p.hi);
        else if (mx) hipLaunchKernelGGL((sdp::separable_f1_kernel<true, false>), grid, dim3(256), smem, h->stream, S, v_next, v_cur, pol, pd, pp, p.lo, p.hi);
        else if (future) hipLaunchKernelGGL((sdp::separable_f1_kernel<false, true>), grid, dim3(256), smem, h->stream, S, v_next, v_cur, pol, pd, pp, p.lo, p.hi);
        else hipLaunchKernelGGL((sdp::separable_f1_kernel<false, false>), grid, dim3(256), smem, h->stream, S, v_next, v_cur, pol, pd, pp, p.lo, p.hi);
        es = hipGetLastError();
      }
    }
    if (es != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "period %d separable kernel: %s", period, hipGetErrorString(es));
    p.kernel_used = SDPGPU_KERNEL_SEPARABLE;
    if (h->profiling) {
      HIP_TRY(h, hipEventRecord(p.ev1, h->stream));
      p.timed = true;
    }
    h->period_done[period - 1] = 1;
    h->policy_done[period - 1] = 1;
    if (!h->d.store_all_values && period + 2 <= h->T) h->period_done[period + 1] = 0;
    return SDPGPU_OK;
  }
  bool use_window = false;
  if (h->d.kernel == SDPGPU_KERNEL_WINDOW) {
    if (!window_eligible(h, period)) return fail(h, SDPGPU_ERR_UNSUPPORTED, "window kernel needs the backorder / lead-time family with a unit-stride demand grid");
    use_window = true;
  } else if (h->d.kernel == SDPGPU_KERNEL_AUTO) {
    use_window = window_eligible(h, period);
  }
  if (part != SDPGPU_PART_ALL) {
    // only the F1 window kernel has a bounded dependency footprint; everything else is "all boundary"
    const bool splittable = use_window && window_interior_tiles(h, period, p.lo, p.hi, nullptr, nullptr);
    if (!splittable) {
      if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;  // nothing can start before the exchange
      part = SDPGPU_PART_ALL;
    }
  }
  hipError_t e;
  if (use_window) {
    e = launch_window(h, P, period, v_next, v_cur, pol, pd, pp, p.lo, p.hi, h->stream, part);
    p.kernel_used = SDPGPU_KERNEL_WINDOW;
  } else if (h->d.kernel != SDPGPU_KERNEL_GATHER && h->use_cash_shift && cash_shift_eligible(h, period)) {
    e = flush_pending(h);
    if (e == hipSuccess) e = launch_cash_shift(h, P, period, v_next, v_cur, pol, pd, pp, p.lo, p.hi, h->stream);
    p.kernel_used = SDPGPU_KERNEL_WINDOW;  // reported as a specialised (non-gather) kernel
  } else if (h->d.kernel != SDPGPU_KERNEL_GATHER && cash_row_eligible(h, period)) {
    e = flush_pending(h);
    if (e == hipSuccess) e = launch_cash_row(h, P, period, v_next, v_cur, pol, pd, pp, p.lo, p.hi, h->stream);
    p.kernel_used = SDPGPU_KERNEL_WINDOW;
  } else {
    e = flush_pending(h);  // the gather kernel reads the final V_{t+1} row
    if (e == hipSuccess)
      e = launch_gather<false>(P, v_next, v_cur, pol, pd, pp, p.lo, p.hi, sdp::QueryStates{nullptr, nullptr, nullptr, nullptr}, h->stream);
    p.kernel_used = SDPGPU_KERNEL_GATHER;
  }
  if (e != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "period %d kernel launch: %s", period, hipGetErrorString(e));
  if (h->profiling) {
    HIP_TRY(h, hipEventRecord(p.ev1, h->stream));
    p.timed = true;
  } else {
    p.timed = false;
  }
  if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;  // the period is complete only after its boundary part
  h->period_done[period - 1] = 1;
  h->policy_done[period - 1] = 1;
  if (!h->d.store_all_values && period + 2 <= h->T) h->period_done[period + 1] = 0;  // V_{t+2} was overwritten
  return SDPGPU_OK;
}

template <int FAM>
hipError_t launch_reach_fam(const DevParams& P, const uint8_t* mcur, uint8_t* mnext, const double* pmf_d, int64_t n,
                            sdp::QueryStates q, bool query, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  unsigned blocks = (unsigned)((n + 63) / 64);
  if (query)
    hipLaunchKernelGGL((sdp::reach_kernel<FAM, true>), dim3(blocks), dim3(256), 0, st, P, mcur, mnext, pmf_d, n, q);
  else
    hipLaunchKernelGGL((sdp::reach_kernel<FAM, false>), dim3(blocks), dim3(256), 0, st, P, mcur, mnext, pmf_d, n, q);
  return hipGetLastError();
}

hipError_t launch_reach(const DevParams& P, const uint8_t* mcur, uint8_t* mnext, const double* pmf_d, int64_t n,
                        sdp::QueryStates q, bool query, hipStream_t st) {
  switch (P.family) {
    case sdp::FAM_BACKORDER: return launch_reach_fam<sdp::FAM_BACKORDER>(P, mcur, mnext, pmf_d, n, q, query, st);
    case sdp::FAM_LEADTIME: return launch_reach_fam<sdp::FAM_LEADTIME>(P, mcur, mnext, pmf_d, n, q, query, st);
    case sdp::FAM_CASH: return launch_reach_fam<sdp::FAM_CASH>(P, mcur, mnext, pmf_d, n, q, query, st);
    case sdp::FAM_OVERDRAFT: return launch_reach_fam<sdp::FAM_OVERDRAFT>(P, mcur, mnext, pmf_d, n, q, query, st);
    case sdp::FAM_CASH_LEADTIME: return launch_reach_fam<sdp::FAM_CASH_LEADTIME>(P, mcur, mnext, pmf_d, n, q, query, st);
    case sdp::FAM_SURVIVAL: return launch_reach_fam<sdp::FAM_SURVIVAL>(P, mcur, mnext, pmf_d, n, q, query, st);
  }
  return hipErrorInvalidValue;
}

// Forward propagation from (1, ini_inventory, ini_cash, ini_preq) through every period.
int compute_reachable(sdpgpu_handle* h) {
  if (h->reach_done) return SDPGPU_OK;
  int rc = allocate(h);
  if (rc) return rc;
  rc = ensure_device(h);
  if (rc) return rc;
  h->reach_off.assign((size_t)h->T, 0);
  size_t total = 0;
  for (int t = 0; t < h->T; ++t) {
    h->reach_off[t] = total;
    total += (size_t)h->per[t].S;
  }
  if (!h->d_reach) HIP_TRY(h, hipMalloc((void**)&h->d_reach, std::max<size_t>(total, 1)));
  HIP_TRY(h, hipMemsetAsync(h->d_reach, 0, std::max<size_t>(total, 1), h->stream));
  const sdpgpu_desc& d = h->d;
  double ini[4] = {d.ini_inventory, has_cash(d.family) ? d.ini_cash : 0.0, has_preq(d.family) ? d.ini_preq : 0.0,
                   d.lead_time == 2 ? d.ini_preq2 : 0.0};
  int64_t i0 = sdpgpu_state_index2(h, 1, ini[0], ini[1], ini[2], ini[3]);
  if (i0 >= 0) {
    uint8_t one = 1;
    HIP_TRY(h, hipMemcpyAsync(h->d_reach + i0, &one, 1, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  double* d_ini = nullptr;
  HIP_TRY(h, hipMalloc((void**)&d_ini, sizeof ini));
  hipError_t e = hipMemcpy(d_ini, ini, sizeof ini, hipMemcpyHostToDevice);
  for (int period = 1; period < h->T && e == hipSuccess; ++period) {
    DevParams P = make_params(h, period);
    const PeriodInfo& p = h->per[period - 1];
    const double* pd = h->d_pmf + p.pmf_off;
    uint8_t* mnext = h->d_reach + h->reach_off[period];
    if (h->custom)
      e = period == 1 ? launch_custom_reach(h, 1, nullptr, mnext, 1, d_ini, d_ini + 1, d_ini + 2)
                      : launch_custom_reach(h, period, h->d_reach + h->reach_off[period - 1], mnext, p.S, nullptr, nullptr, nullptr);
    else if (period == 1)
      e = launch_reach(P, nullptr, mnext, pd, 1, sdp::QueryStates{d_ini, d_ini + 1, d_ini + 2, d_ini + 3}, true, h->stream);
    else
      e = launch_reach(P, h->d_reach + h->reach_off[period - 1], mnext, pd, p.S, sdp::QueryStates{nullptr, nullptr, nullptr, nullptr}, false, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d_ini);
  if (e != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "reachable: %s", hipGetErrorString(e));
  rc = custom_check(h);
  if (rc) return rc;
  h->reach_done = true;
  return SDPGPU_OK;
}

// ---- uniform-shift kernel (F3 on dyadic grids) ---------------------------------------------------
bool dyadic(double x, double scale, double max_abs) { return std::fabs(x) <= max_abs && x * scale == std::floor(x * scale); }

// All arithmetic of the F3 lambdas is exact (see sdp_cash.hpp) iff the rates and the penalty are zero, the
// cash quantum is a power of two and every parameter is a multiple of 2^-10 of bounded size.
bool cash_shift_eligible(const sdpgpu_handle* h, int period) {
  const sdpgpu_desc& d = h->d;
  if (h->custom) return false;
  if (d.family != SDPGPU_FAMILY_CASH || !d.clamp_inventory) return false;
  if (d.deposit_rate != 0 || d.overhead_rate != 0 || d.penalty_cost != 0) return false;
  double q;
  if (d.cash_round_int_div) {
    if (d.cash_round_mult != 1.0 || d.cash_round_div != 1.0) return false;
    q = 1.0;
  } else {
    q = d.cash_round_div;  // == mult (validated at create)
  }
  if (!is_pow2_int(q) || q > 1024) return false;
  const PeriodInfo& p = h->per[period - 1];
  const double S = 1024.0, M = 4096.0;
  if (!dyadic(d.price, S, M) || !dyadic(d.fixed_order_cost, S, M) || !dyadic(d.unit_order_cost, S, M) ||
      !dyadic(d.holding_cost, S, M) || !dyadic(d.salvage_value, S, M) || !dyadic(p.overhead, S, 1048576.0))
    return false;
  if (!(d.unit_order_cost != 0)) return false;
  if (!dyadic(d.min_cash, q, 1e9) || !dyadic(d.max_cash, q, 1e9)) return false;
  if (!dyadic(d.discount_factor, 1.0, 1.0) && d.discount_factor != 1.0) {
    // gamma only multiplies p_j (inexact anyway, same product as the general kernel): any value is fine
  }
  double ymax = std::fabs(d.max_inventory) + std::fabs(d.min_inventory) + d.max_order_quantity * d.step;
  double dmax = 0;
  for (double v : h->pmf_d[period - 1]) dmax = std::max(dmax, std::fabs(v));
  if (ymax > 1048576.0 || dmax > 1048576.0) return false;
  double incmax = (std::fabs(d.price) + std::fabs(d.holding_cost) + std::fabs(d.salvage_value)) * (ymax + dmax) +
                  std::fabs(d.fixed_order_cost) + std::fabs(d.unit_order_cost) * d.max_order_quantity * d.step + std::fabs(p.overhead);
  if (incmax * q > 1.0e9) return false;
  if (p.S >= 2147483647LL || p.nD > 2000) return false;
  return true;
}

hipError_t launch_cash_shift(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                             int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st) {
  if (hi <= lo) return hipSuccess;
  const sdpgpu_desc& d = h->d;
  const PeriodInfo& p = h->per[period - 1];
  sdp::CashShiftParams C{};
  C.price = d.price;
  C.K = d.fixed_order_cost;
  C.v = d.unit_order_cost;
  C.h = d.holding_cost;
  C.overhead = p.overhead;
  C.salvage = d.salvage_value;
  C.gamma = P.gamma;
  C.step = d.step;
  C.x_lo = p.g.x_lo;
  C.min_inventory = d.min_inventory;
  C.max_inventory = d.max_inventory;
  C.next_x_lo = period < h->T ? h->per[period].g.x_lo : p.g.x_lo;
  C.q = d.cash_round_int_div ? 1.0 : d.cash_round_div;
  C.k_lo = p.g.k_lo;
  C.nx = (int32_t)p.g.nx;
  C.nc = (int32_t)p.g.nc;
  C.n_demand = p.nD;
  C.max_order_quantity = d.max_order_quantity;
  C.is_last = period == h->T;
  C.tiles_per_row = (int32_t)((p.g.nc + 63) / 64);
  const int64_t row_lo = lo / p.g.nc, row_hi = (hi - 1) / p.g.nc;
  C.row0 = (int32_t)row_lo;
  if (!grid_ok((row_hi - row_lo + 1) * (int64_t)C.tiles_per_row)) return hipErrorInvalidValue;
  dim3 grid((unsigned)((row_hi - row_lo + 1) * C.tiles_per_row));
  size_t smem = (size_t)p.nD * 16 * 5 + 4 * 64 * (sizeof(double) + sizeof(int));
  const bool last = period == h->T;
#define SDP_CS(MX, LS) hipLaunchKernelGGL((sdp::cash_shift_kernel<MX, LS>), grid, dim3(256), smem, st, C, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi)
  if (P.maxdir) {
    if (last) SDP_CS(true, true); else SDP_CS(true, false);
  } else {
    if (last) SDP_CS(false, true); else SDP_CS(false, false);
  }
#undef SDP_CS
  return hipGetLastError();
}

// ---- cash row kernel (F3-F6 on any cash grid) -------------------------------------------------------
bool cash_row_eligible(const sdpgpu_handle* h, int period) {
  const sdpgpu_desc& d = h->d;
  if (h->custom || !has_cash(d.family) || !d.clamp_inventory || !h->use_cash_row) return false;
  if (d.family == SDPGPU_FAMILY_CASH && d.penalty_cost != 0) return false;  // the end-cash penalty branch: generic kernel
  const PeriodInfo& p = h->per[period - 1];
  if (p.g.nc < 32) return false;                       // a wave is 64 consecutive cash points of one row
  if (p.S >= 2147483647LL) return false;               // 32-bit row offsets
  if ((size_t)p.nD * 152 + 4 * 64 * 12 > 64 * 1024) return false;  // per-wave entries of every demand point in LDS
  return true;
}

template <int FAM, bool FORMULA1>
hipError_t launch_cash_row_fam(const DevParams& P, bool last, bool intdiv, const double* v_next, double* v_cur,
                               int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi,
                               int64_t row0, int tiles_per_row, dim3 grid, size_t smem, hipStream_t st) {
#define SDP_CR(LS, ID) hipLaunchKernelGGL((sdp::cash_row_kernel<FAM, LS, FORMULA1, ID>), grid, dim3(256), smem, st, P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, row0, tiles_per_row)
  if (intdiv) {
    if (last) SDP_CR(true, true); else SDP_CR(false, true);
  } else {
    if (last) SDP_CR(true, false); else SDP_CR(false, false);
  }
#undef SDP_CR
  return hipGetLastError();
}

hipError_t launch_cash_row(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                           int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st) {
  if (hi <= lo) return hipSuccess;
  const PeriodInfo& p = h->per[period - 1];
  const int tiles_per_row = (int)((p.g.nc + 63) / 64);
  const int64_t row_lo = lo / p.g.nc, row_hi = (hi - 1) / p.g.nc;
  const int64_t blocks = (row_hi - row_lo + 1) * (int64_t)tiles_per_row;
  if (!grid_ok(blocks)) return hipErrorInvalidValue;
  dim3 grid((unsigned)blocks);
  const size_t smem = (size_t)p.nD * 152 + 4 * 64 * (sizeof(double) + sizeof(int));
  const bool last = period == h->T;
  const bool intdiv = h->d.cash_round_int_div && h->d.cash_round_div != 1.0;
#define SDP_ROWARGS P, last, intdiv, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, row_lo, tiles_per_row, grid, smem, st
  switch (P.family) {
    case sdp::FAM_CASH:
      return P.cash_formula == 0 ? launch_cash_row_fam<sdp::FAM_CASH, false>(SDP_ROWARGS)
                                 : launch_cash_row_fam<sdp::FAM_CASH, true>(SDP_ROWARGS);
    case sdp::FAM_OVERDRAFT: return launch_cash_row_fam<sdp::FAM_OVERDRAFT, false>(SDP_ROWARGS);
    case sdp::FAM_CASH_LEADTIME: return launch_cash_row_fam<sdp::FAM_CASH_LEADTIME, false>(SDP_ROWARGS);
    case sdp::FAM_SURVIVAL: return launch_cash_row_fam<sdp::FAM_SURVIVAL, false>(SDP_ROWARGS);
  }
#undef SDP_ROWARGS
  return hipErrorInvalidValue;
}

// ---- window kernel (F1) -----------------------------------------------------------------------

struct WinPlan {
  int R = 0, S = 1, d_pad = 0, n_chunks = 1, chunk_blocks = 0, n_tiles = 0, n_tasks = 0;
  size_t smem = 0;
  int tile_states() const { return 64 * S; }
};

// F1 / F2 with a unit-stride demand grid: d_j = d_0 + j*step.
bool window_eligible(const sdpgpu_handle* h, int period) {
  if (h->custom) return false;
  if (h->d.family != SDPGPU_FAMILY_BACKORDER && h->d.family != SDPGPU_FAMILY_LEADTIME) return false;
  const std::vector<double>& d = h->pmf_d[period - 1];
  for (size_t j = 1; j < d.size(); ++j)
    if (d[j] - d[j - 1] != h->d.step) return false;
  const PeriodInfo& p = h->per[period - 1];
  if (p.S >= 2147483647LL - 4096) return false;
  if (h->n_actions_full + p.nD > 3500) return false;
  return true;
}

// One task = one wave = (tile of 64*S states, run of R-blocks).  The measured timeline of a SIMD is task
// after task, so a launch costs  rounds x task time  with rounds = ceil(tasks / 1024 SIMDs): pick the
// register block R, the states per lane S and the number of chunks per tile that minimise it (fewest chunks
// on ties: fewer chunk rows, less staging).  More states per lane = fewer fp64 operations per cell
// ((5 + 4(S-1)) / S, see window_f1_kernel) but bigger, fewer tasks: small grids keep S low.
WinPlan plan_window(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi) {
  const PeriodInfo& p = h->per[period - 1];
  const int A = h->n_actions_full, D = p.nD;
  WinPlan best;
  double best_cost = -1;
  // The plan is chosen from the NOMINAL slab S_pad / world_size, which is the same on every rank: ranks
  // must agree on whether a period's row is exchanged as keys or as fp64 values, whatever their own
  // (possibly clipped or empty) slab looks like.
  const int64_t nominal = p.S_pad / std::max(1, h->d.world_size);
  auto rup = [](int v, int r) { return (v + r - 1) / r * r; };
  struct Cand {
    int r, s, occupancy;  // occupancy: waves a SIMD can hold within the register budget
  };
  // (84 / 62 / 54 VGPRs for S = 1, R = 8 / 5 / 4; 134 / 116 / 96 for S = 2; 130 for R = 4, S = 4)
  const Cand cand[] = {{8, 1, 6}, {5, 1, 8}, {4, 1, 9}, {8, 2, 3}, {4, 2, 5}, {4, 4, 3}};
  const bool may_chunk = h->fuse_combine && h->d.store_all_values;
  for (const Cand& c : cand) {
    const int r = c.r, sl = c.s, nw = r + sl - 1, ts = 64 * sl;
    if (h->win_r && r != h->win_r) continue;
    if (h->win_s && sl != h->win_s) continue;
    const int64_t n_tiles = std::max<int64_t>(1, (nominal + ts - 1) / ts);
    const int64_t own_tiles = (hi - lo + ts - 1) / ts;
    const int d_pad = rup(D, nw);
    const int blocks_total = rup(A, r) / r;
    // cost of one R-block on one SIMD, in fp64-instruction units per lane: (5 + 4(S-1)) ops per S cells of an
    // action plus a per-step overhead (LDS read, scalar load, waits) that a bigger register block amortises
    const double block_cost = (double)D * ((5.0 + 4.0 * (sl - 1)) * r + 3.0) + 60.0 + 2.0 * (sl - 1) * r;
    for (int nch = 1; nch <= blocks_total; ++nch) {
      if (h->win_nch && nch != std::min(h->win_nch, blocks_total)) continue;
      const int bpc = (blocks_total + nch - 1) / nch;
      if ((blocks_total + bpc - 1) / bpc != nch) continue;  // same plan as a smaller nch
      if (nch > 1 && !may_chunk) continue;  // chunk rows need the deferred key/finalize scheme
      const int span = ts + bpc * r + d_pad + sl;
      const size_t smem = (size_t)4 * span * 16;
      if (smem > 64 * 1024) continue;
      const int64_t tasks = n_tiles * nch;
      const int64_t rounds = (tasks + 1023) / 1024;  // tasks the busiest SIMD runs, one after the other
      // fp64 issue rate one SIMD sustains with w resident waves (tools/valu_probe): 0.76 / 0.86 / 0.94 / 0.97
      const int64_t w = std::min<int64_t>(rounds, c.occupancy);
      const double eff = w >= 8 ? 0.97 : (w >= 4 ? 0.94 : (w >= 3 ? 0.90 : (w >= 2 ? 0.86 : 0.76)));
      const double staging = 400.0 + 4.0 * span;
      const double cost = (double)rounds * (bpc * block_cost + staging) / eff;
      if (best_cost < 0 || cost < best_cost * 0.999) {
        best_cost = cost;
        best.R = r;
        best.S = sl;
        best.d_pad = d_pad;
        best.n_chunks = nch;
        best.chunk_blocks = bpc;
        best.n_tiles = (int)own_tiles;
        best.n_tasks = (int)(own_tiles * nch);
        best.smem = smem;
      }
    }
  }
  return best;
}

hipError_t ensure_partials(sdpgpu_handle* h, int b, size_t need) {
  if (need <= h->part_elems[b]) return hipSuccess;
  // (re)allocation frees a buffer earlier launches may still read: drain the stream first
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return e;
  if (h->d_part_val[b]) (void)hipFree(h->d_part_val[b]);
  if (h->d_part_idx[b]) (void)hipFree(h->d_part_idx[b]);
  h->d_part_val[b] = nullptr;
  h->d_part_idx[b] = nullptr;
  h->part_elems[b] = 0;
  e = hipMalloc((void**)&h->d_part_val[b], need * sizeof(double));
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&h->d_part_idx[b], need * sizeof(int32_t));
  if (e != hipSuccess) return e;
  h->part_elems[b] = need;
  return hipSuccess;
}

template <bool MAXDIR>
hipError_t launch_combine(const double* pv, const int32_t* pi, int n_chunks, int64_t stride, double* v_cur, int32_t* pol,
                          int64_t lo, int64_t hi, hipStream_t st) {
  unsigned blocks = (unsigned)((hi - lo + 255) / 256);
  hipLaunchKernelGGL((sdp::window_combine_kernel<MAXDIR>), dim3(blocks), dim3(256), 0, st, pv, pi, n_chunks, stride, v_cur, pol, lo, hi);
  return hipGetLastError();
}

// Turn every pending period's keys + chunk rows into its final V_t / policy rows: one launch.
hipError_t flush_pending(sdpgpu_handle* h) {
  if (h->n_pending == 0) return hipSuccess;
  std::vector<sdp::FinalizeJob> jobs;
  int64_t total = 0;
  for (int t = 0; t < h->T; ++t) {
    if (h->pending_chunks[t] <= 0) continue;
    const PeriodInfo& p = h->per[t];
    sdp::FinalizeJob J{};
    J.keys = h->d_keys + (size_t)t * h->key_stride;
    J.part_val = h->d_chunk_val + h->chunk_off[t] - p.lo;
    J.part_idx = h->d_chunk_idx + h->chunk_off[t] - p.lo;
    J.v_out = h->d_values + p.v_off;
    J.pol_out = h->d_policy + p.pol_off - p.lo;
    J.stride = p.hi - p.lo;
    J.lo = p.lo;
    J.hi = p.hi;
    // V_t is decoded over the whole row (after the all-gather every rank holds all keys), the policy
    // only for this rank's slab
    J.vlo = h->d.world_size > 1 ? 0 : p.lo;
    J.vhi = h->d.world_size > 1 ? p.S : p.hi;
    J.first = total;
    J.n_chunks = h->pending_chunks[t];
    total += J.vhi - J.vlo;
    jobs.push_back(J);
    h->pending_chunks[t] = 0;
  }
  h->n_pending = 0;
  if (jobs.empty() || total == 0) return hipSuccess;
  if (!h->d_jobs) {
    hipError_t e = hipMalloc((void**)&h->d_jobs, (size_t)h->T * sizeof(sdp::FinalizeJob));
    if (e != hipSuccess) return e;
  }
  // pageable source: HIP stages the bytes before hipMemcpyAsync returns, so `jobs` may go out of scope
  hipError_t e = hipMemcpyAsync(h->d_jobs, jobs.data(), jobs.size() * sizeof(sdp::FinalizeJob), hipMemcpyHostToDevice, h->stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sdp::finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->d_jobs,
                     (int)jobs.size(), total);
  return hipGetLastError();
}

// ---- row-window kernel (F2) ---------------------------------------------------------------
template <int R, int S, bool MAXDIR>
hipError_t launch_row_r(const sdp::RowParams& W, size_t smem, bool future, const double* v_next, double* out_val,
                        int32_t* out_idx, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st) {
  if (!grid_ok((int64_t)W.n_tiles * W.n_chunks)) return hipErrorInvalidValue;
  dim3 grid((unsigned)((int64_t)W.n_tiles * W.n_chunks));
  if (future)
    hipLaunchKernelGGL((sdp::window_f2_kernel<R, S, MAXDIR, true>), grid, dim3(256), smem, st, W, v_next, out_val, out_idx, pmf_p, lo, hi);
  else
    hipLaunchKernelGGL((sdp::window_f2_kernel<R, S, MAXDIR, false>), grid, dim3(256), smem, st, W, v_next, out_val, out_idx, pmf_p, lo, hi);
  return hipGetLastError();
}

hipError_t launch_row_window(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                             int32_t* pol, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st) {
  {
    hipError_t ef = flush_pending(h);
    if (ef != hipSuccess) return ef;
  }
  const PeriodInfo& p = h->per[period - 1];
  const int A = h->n_actions_full, D = p.nD;
  auto rup = [](int v, int r) { return (v + r - 1) / r * r; };
  int R = 0;
  int64_t best_cost = -1;
  const int cand[3] = {8, 5, 4};
  for (int r : cand) {
    if (h->win_r && r != h->win_r) continue;
    int64_t cost = (int64_t)rup(A, r) * rup(D, r);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      R = r;
    }
  }
  if (!R) R = 8;
  // states per lane: 2 (tiles of 128) unless the inventory axis is short or overridden (SDPGPU_WIN_S)
  int SL = h->win_s ? h->win_s : (p.g.nx >= 96 ? 2 : 1);
  if (SL != 1 && SL != 2 && SL != 4) SL = 2;
  if (SL == 4 && R == 8) SL = 2;  // (no 8 x 4 instantiation: too many registers)
  const int TSZ = 64 * SL;
  const bool future = period < h->T;
  sdp::RowParams W{};
  W.lev0 = p.g.x_lo - h->pmf_d[period - 1][0];
  W.step = h->d.step;
  W.h = h->d.holding_cost;
  W.pi = h->d.penalty_cost;
  W.K = h->d.fixed_order_cost;
  W.v = h->d.unit_order_cost;
  if (future) {
    W.idx_off = (int32_t)((W.lev0 - h->per[period].g.x_lo) / h->d.step);
    W.next_last = (int32_t)(h->per[period].g.nx - 1);
    W.next_nx = (int32_t)h->per[period].g.nx;
  }
  W.cur_nx = (int32_t)p.g.nx;
  W.nq1 = (int32_t)p.g.nq1;
  W.plane_stride = P.lead2 ? (int64_t)W.nq1 * W.next_nx : (int64_t)W.next_nx;
  W.tiles_per_row = (int32_t)((p.g.nx + TSZ - 1) / TSZ);
  W.n_actions = A;
  W.d_pad = rup(D, 4);  // the demand loop is unrolled by S (1, 2 or 4); padded steps carry p = 0
  const int span = TSZ + W.d_pad + 1;
  const int blocks_total = rup(A, R) / R;
  // one R-block per wave: chunks of 4 R-blocks, fewer if the LDS budget (rows of `span` doubles) says so
  int bpc = std::min(4, blocks_total);
  if (h->win_nch) bpc = std::max(1, (blocks_total + h->win_nch - 1) / h->win_nch);
  auto lds = [&](int b) { return (size_t)span * 8 * (1 + (future ? b * R : 0)) + (size_t)4 * TSZ * 12; };
  while (bpc > 1 && lds(bpc) > 60 * 1024) --bpc;
  if (lds(bpc) > 64 * 1024) return hipErrorInvalidValue;
  W.chunk_actions = bpc * R;
  W.n_chunks = (blocks_total + bpc - 1) / bpc;
  // the run of row tiles that covers [lo, hi)
  auto tile_of = [&](int64_t idx) { return (int32_t)((idx / p.g.nx) * W.tiles_per_row + (idx % p.g.nx) / TSZ); };
  W.tile0 = tile_of(lo);
  W.n_tiles = tile_of(hi - 1) - W.tile0 + 1;
  double* out_val = v_cur;
  int32_t* out_idx = pol;
  if (W.n_chunks > 1) {
    int64_t slab = hi - lo;
    const int b = period & 1;
    hipError_t e = ensure_partials(h, b, (size_t)W.n_chunks * (size_t)slab);
    if (e != hipSuccess) return e;
    W.partial_stride = slab;
    out_val = h->d_part_val[b] - lo;
    out_idx = h->d_part_idx[b] - lo;
  }
  hipError_t e = hipErrorInvalidValue;
  size_t smem = lds(bpc);
#define SDP_ROW(RR, SS)                                                                                          \
  if (R == RR && SL == SS)                                                                                       \
    e = P.maxdir ? launch_row_r<RR, SS, true>(W, smem, future, v_next, out_val, out_idx, pmf_p, lo, hi, st)      \
                 : launch_row_r<RR, SS, false>(W, smem, future, v_next, out_val, out_idx, pmf_p, lo, hi, st);
  SDP_ROW(8, 1) SDP_ROW(5, 1) SDP_ROW(4, 1)
  SDP_ROW(8, 2) SDP_ROW(5, 2) SDP_ROW(4, 2)
  SDP_ROW(5, 4) SDP_ROW(4, 4)
#undef SDP_ROW
  if (e != hipSuccess) return e;
  if (W.n_chunks > 1)
    e = P.maxdir ? launch_combine<true>(out_val, out_idx, W.n_chunks, W.partial_stride, v_cur, pol, lo, hi, st)
                 : launch_combine<false>(out_val, out_idx, W.n_chunks, W.partial_stride, v_cur, pol, lo, hi, st);
  return e;
}

// The interior run of slab tiles of a period: tiles whose whole V_{t+1} footprint
// [i0 + idx_off - (D-1), i0 + 63 + idx_off + A - 1] (before the clamp to the grid) lies inside this rank's
// slab of the next period's row, or is clamped at a grid edge this rank owns.
bool window_interior_tiles(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi, int* first, int* count) {
  if (h->d.family != SDPGPU_FAMILY_BACKORDER || period >= h->T || h->d.world_size == 1) return false;
  const PeriodInfo& p = h->per[period - 1];
  const PeriodInfo& pn = h->per[period];
  const int64_t ts = plan_window(h, period, lo, hi).tile_states();
  const int64_t n_tiles = (hi - lo + ts - 1) / ts;
  const double lev0 = p.g.x_lo - h->pmf_d[period - 1][0];
  const int64_t idx_off = (int64_t)((lev0 - pn.g.x_lo) / h->d.step);
  const int64_t A = h->n_actions_full, D = p.nD;
  int64_t f = -1, c = 0;
  for (int64_t u = 0; u < n_tiles; ++u) {
    const int64_t i0 = lo + u * ts;
    int64_t a = i0 + idx_off - (D - 1), b = i0 + ts - 1 + idx_off + A - 1;
    a = std::max<int64_t>(0, std::min<int64_t>(a, pn.g.nx - 1));  // the kernel clamps reads to the grid
    b = std::max<int64_t>(0, std::min<int64_t>(b, pn.g.nx - 1));
    const bool inside = a >= pn.lo && b < pn.hi;
    if (inside) {
      if (f < 0) f = u;
      if (u != f + c) return false;  // not one contiguous run: do not split
      ++c;
    }
  }
  if (c <= 0) return false;
  if (first) *first = (int)f;
  if (count) *count = (int)c;
  return true;
}

hipError_t launch_window(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                         int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st,
                         int part) {
  (void)pmf_d;
  if (h->d.family == SDPGPU_FAMILY_LEADTIME) {
    if (hi <= lo) return hipSuccess;
    return launch_row_window(h, P, period, v_next, v_cur, pol, pmf_p, lo, hi, st);
  }
  // (an empty slab still goes through the bookkeeping below: every rank must treat the row alike)
  PeriodInfo& p = h->per[period - 1];
  WinPlan pl = plan_window(h, period, lo, hi);
  if (!pl.R) return hipErrorInvalidValue;
  if (period == h->T && std::getenv("SDPGPU_DEBUG_PLAN"))
    std::fprintf(stderr, "[sdpgpu] window plan: R=%d S=%d chunks=%d blocks/chunk=%d tiles=%d tasks=%d lds=%zu\n", pl.R, pl.S,
                 pl.n_chunks, pl.chunk_blocks, pl.n_tiles, pl.n_tasks, pl.smem);
  const bool future = period < h->T;
  const bool chunked = pl.n_chunks > 1;
  // a period is never re-run on top of its own pending rows, and a new sweep (period T) first
  // finalizes what the previous one left: the key rows are about to be reset
  // (the BOUNDARY half of a split period continues what its INTERIOR half started: no reset there)
  const bool continuing = part == SDPGPU_PART_BOUNDARY;
  if (!continuing && h->n_pending > 0 && (h->pending_chunks[period - 1] > 0 || period == h->T)) {
    hipError_t e = flush_pending(h);
    if (e != hipSuccess) return e;
  }
  // where V_{t+1} comes from: its key row while that period is still pending, else the final fp64 row
  const bool keyed_in = future && h->pending_chunks[period] > 0;
  if (chunked) {
    if (!h->d_chunk_val) {  // one-time arenas: a key row per period, the chunk rows of every chunked period
      size_t stride = 0;
      for (const PeriodInfo& q : h->per) stride = std::max<size_t>(stride, (size_t)q.S_pad);
      hipError_t e = hipSuccess;
      if (!h->d_keys) e = hipMalloc((void**)&h->d_keys, (size_t)h->T * stride * sizeof(unsigned long long));
      if (e != hipSuccess) return e;
      h->key_stride = stride;
      h->key_row_clean.assign((size_t)h->T, 0);
      h->chunk_off.assign((size_t)h->T, 0);
      size_t total = 0;
      for (int t = 0; t < h->T; ++t) {
        const PeriodInfo& q = h->per[t];
        h->chunk_off[t] = total;
        if (window_eligible(h, t + 1))
          total += (size_t)plan_window(h, t + 1, q.lo, q.hi).n_chunks * (size_t)std::max<int64_t>(q.hi - q.lo, 0);
      }
      e = hipMalloc((void**)&h->d_chunk_val, std::max<size_t>(total, 1) * sizeof(double));
      if (e == hipSuccess) e = hipMalloc((void**)&h->d_chunk_idx, std::max<size_t>(total, 1) * sizeof(int32_t));
      if (e != hipSuccess) return e;
    }
    if (!continuing && !h->key_row_clean[period - 1]) {
      // reset key rows to the reduction identity: all of them when nothing is pending (the usual case:
      // period T of a new sweep), else only this period's row (periods re-run out of order)
      const bool all = h->n_pending == 0;
      const int64_t n = (all ? (int64_t)h->T : 1) * (int64_t)h->key_stride;
      unsigned long long* base = all ? h->d_keys : h->d_keys + (size_t)(period - 1) * h->key_stride;
      hipLaunchKernelGGL(sdp::key_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, base, n, (int)P.maxdir);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      if (all)
        std::fill(h->key_row_clean.begin(), h->key_row_clean.end(), 1);
      else
        h->key_row_clean[period - 1] = 1;
    }
  }
  sdp::WinParams W{};
  const double d0 = h->pmf_d[period - 1][0];
  W.lev0 = p.g.x_lo - d0;
  W.step = h->d.step;
  W.h = h->d.holding_cost;
  W.pi = h->d.penalty_cost;
  W.K = h->d.fixed_order_cost;
  W.v = h->d.unit_order_cost;
  if (future) {
    W.idx_off = (int32_t)((W.lev0 - h->per[period].g.x_lo) / h->d.step);
    W.next_last = (int32_t)(h->per[period].g.nx - 1);
  }
  W.n_actions = h->n_actions_full;
  W.d_pad = pl.d_pad;
  W.d_main = p.nD / (pl.R + pl.S - 1) * (pl.R + pl.S - 1);
  W.maxdir = P.maxdir;
  W.n_demand = p.nD;
  W.n_chunks = pl.n_chunks;
  W.chunk_blocks = pl.chunk_blocks;
  W.n_tiles = pl.n_tiles;
  W.n_tasks = pl.n_tasks;
  W.tile_first = 0;
  W.prio_fair = h->win_prio_fair;
  W.tile_gap_at = pl.n_tiles;  // no gap
  W.tile_gap = 0;
  if (part != SDPGPU_PART_ALL) {
    int first = 0, count = 0;
    if (!window_interior_tiles(h, period, lo, hi, &first, &count)) return hipErrorInvalidValue;
    if (part == SDPGPU_PART_INTERIOR) {
      W.tile_first = first;
      W.n_tiles = count;
    } else {  // the tiles below and above the interior run, in one launch
      // The boundary runs are a handful of tiles on the critical path behind the exchange: cut them finer than
      // the plan does (64-state tiles, 4-action register blocks -- same chunks, so the chunk rows line up) so
      // that the few tasks spread over more SIMDs and each is short.
      const int ratio = pl.S;  // plan tiles are ratio x 64 states
      const int fine_r = (pl.R % 4 == 0) ? 4 : pl.R;
      if (ratio > 1 || fine_r != pl.R) {
        const int chunk_actions = pl.chunk_blocks * pl.R;
        pl.n_tiles = (int)((hi - lo + 63) / 64);
        first *= ratio;
        count = std::min(count * ratio, pl.n_tiles - first);  // (the last plan tile may be a partial one)
        pl.chunk_blocks = chunk_actions / fine_r;
        pl.R = fine_r;
        pl.S = 1;
        pl.d_pad = (p.nD + fine_r - 1) / fine_r * fine_r;
        pl.smem = (size_t)4 * (64 + chunk_actions + pl.d_pad + 1) * 16;
        W.d_pad = pl.d_pad;
        W.d_main = p.nD / fine_r * fine_r;
        W.chunk_blocks = pl.chunk_blocks;
      }
      W.n_tiles = pl.n_tiles - count;
      W.tile_gap_at = first;
      W.tile_gap = count;
    }
    W.n_tasks = W.n_tiles * pl.n_chunks;
    W.tile_gap_at = std::min(W.tile_gap_at, W.n_tiles);
    if (W.n_tiles == 0) return hipSuccess;
  }
  double* out_val = v_cur;
  int32_t* out_idx = pol;
  unsigned long long* k_cur = nullptr;
  const unsigned long long* k_next = keyed_in ? h->d_keys + (size_t)period * h->key_stride : nullptr;
  if (chunked) {
    W.partial_stride = hi - lo;
    out_val = h->d_chunk_val + h->chunk_off[period - 1] - lo;  // the kernel indexes rows by flat state index
    out_idx = h->d_chunk_idx + h->chunk_off[period - 1] - lo;
    k_cur = h->d_keys + (size_t)(period - 1) * h->key_stride;
  }
  if (W.n_tasks > 0 && !grid_ok((W.n_tasks + 3) / 4)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)std::max(1, (W.n_tasks + 3) / 4));
#ifdef SDP_STAMPS
  static unsigned long long* d_stamps = nullptr;
  if (!d_stamps) (void)hipMalloc((void**)&d_stamps, (size_t)1 << 24);
  unsigned long long* stamps = (period == 2) ? d_stamps : nullptr;  // record one mid-sweep launch
#define SDP_STAMP_ARG , stamps
#else
#define SDP_STAMP_ARG
#endif
  if (W.n_tasks > 0) {
#define SDP_WIN_GO(RR, SS, FU, KI)                                                                                      \
  hipLaunchKernelGGL((sdp::window_f1_kernel<RR, SS, FU, KI>), grid, dim3(256), pl.smem, st, W, v_next, k_next, out_val, \
                     out_idx, k_cur, pmf_p, lo, hi SDP_STAMP_ARG)
#define SDP_WIN_R(RR, SS)                      \
  if (pl.R == RR && pl.S == SS) {              \
    if (!future)                               \
      SDP_WIN_GO(RR, SS, false, false);        \
    else if (keyed_in)                         \
      SDP_WIN_GO(RR, SS, true, true);          \
    else                                       \
      SDP_WIN_GO(RR, SS, true, false);         \
    launched = true;                           \
  }
  bool launched = false;
  SDP_WIN_R(8, 1)
  SDP_WIN_R(5, 1)
  SDP_WIN_R(4, 1)
  SDP_WIN_R(8, 2)
  SDP_WIN_R(4, 2)
  SDP_WIN_R(4, 4)
  if (!launched) return hipErrorInvalidValue;
#undef SDP_WIN_R
#undef SDP_WIN_GO
#undef SDP_STAMP_ARG
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
#ifdef SDP_STAMPS
  if (period == 1) {
    std::vector<unsigned long long> hs((size_t)pl.n_tasks * 5);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    if (FILE* f = std::fopen("gpurun_out/stamps.txt", "w")) {
      for (size_t i = 0; i + 4 < hs.size(); i += 5)
        std::fprintf(f, "%zu %llu %llu %llu %llu %llu\n", i / 5, hs[i], hs[i + 1], hs[i + 2], hs[i + 3], hs[i + 4]);
      std::fclose(f);
    }
  }
#endif
  if (chunked && h->pending_chunks[period - 1] == 0) {
    h->pending_chunks[period - 1] = pl.n_chunks;
    h->n_pending++;
    h->key_row_clean[period - 1] = 0;  // holds data now; re-filled when the next sweep starts
  }
  return e;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
namespace {
// does any period run the F1 window kernel with several tasks per tile (=> key rows are used)?
bool keys_needed(sdpgpu_handle* h) {
  if (h->d.family != SDPGPU_FAMILY_BACKORDER || h->d.kernel == SDPGPU_KERNEL_GATHER) return false;
  if (layout(h)) return false;
  for (int t = 1; t <= h->T; ++t) {
    const PeriodInfo& q = h->per[t - 1];
    if (window_eligible(h, t) && plan_window(h, t, q.lo, q.hi).n_chunks > 1) return true;
  }
  return false;
}

int flush_api(sdpgpu_handle* h) {
  if (h->custom) return custom_check(h);
  if (h->n_pending == 0) return SDPGPU_OK;
  int rc = ensure_device(h);
  if (rc) return rc;
  hipError_t e = flush_pending(h);
  if (e != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "combine: %s", hipGetErrorString(e));
  return SDPGPU_OK;
}
}  // namespace

extern "C" {

int sdpgpu_abi_version(void) { return SDPGPU_ABI_VERSION; }

void sdpgpu_desc_init(sdpgpu_desc* d) {
  if (!d) return;
  std::memset(d, 0, sizeof *d);
  d->abi_version = SDPGPU_ABI_VERSION;
  d->family = SDPGPU_FAMILY_BACKORDER;
  d->direction = SDPGPU_MIN;
  d->periods = 1;
  d->step = 1;
  d->clamp_inventory = 1;
  d->discount_factor = 1;
  d->cash_round_mult = 10;
  d->cash_round_div = 10;
  d->kernel = SDPGPU_KERNEL_AUTO;
  d->device = -1;
  d->rank = 0;
  d->world_size = 1;
  d->store_all_values = 1;
}

int sdpgpu_create(const sdpgpu_desc* desc, sdpgpu_handle** out) {
  g_create_error.clear();
  if (!desc || !out) return fail(nullptr, SDPGPU_ERR_ARG, "null argument");
  *out = nullptr;
  int rc = validate(*desc);
  if (rc) return rc;
  sdpgpu_handle* h = new (std::nothrow) sdpgpu_handle();
  if (!h) return fail(nullptr, SDPGPU_ERR_ARG, "out of host memory");
  try {
    h->d = *desc;
    h->T = desc->periods;
    h->device = desc->device;
    h->per.resize((size_t)h->T);
    h->pmf_d.resize((size_t)h->T);
    h->pmf_p.resize((size_t)h->T);
    h->pmf_set.assign((size_t)h->T, 0);
    h->period_done.assign((size_t)h->T, 0);
    h->policy_done.assign((size_t)h->T, 0);
    h->pending_chunks.assign((size_t)h->T + 1, 0);
    if (const char* e = std::getenv("SDPGPU_WIN_R")) h->win_r = std::atoi(e);
    if (const char* e = std::getenv("SDPGPU_WIN_NCH")) h->win_nch = std::atoi(e);
    if (const char* e = std::getenv("SDPGPU_WIN_S")) h->win_s = std::atoi(e);
    if (const char* e = std::getenv("SDPGPU_FUSE_COMBINE")) h->fuse_combine = std::atoi(e) != 0;
    if (const char* e = std::getenv("SDPGPU_CASH_SHIFT")) h->use_cash_shift = std::atoi(e) != 0;
    if (const char* e = std::getenv("SDPGPU_CASH_ROW")) h->use_cash_row = std::atoi(e) != 0;
    if (const char* e = std::getenv("SDPGPU_WIN_PRIO")) h->win_prio_fair = std::atoi(e) != 0;
  } catch (...) {
    delete h;
    return fail(nullptr, SDPGPU_ERR_ARG, "out of host memory");
  }
  *out = h;
  return SDPGPU_OK;
}

int sdpgpu_create_custom(const sdpgpu_desc* desc, const char* functor_source, const double* params, int32_t n_params,
                         sdpgpu_handle** out) {
  g_create_error.clear();
  if (!desc || !out || !functor_source || n_params < 0 || n_params > 256 || (n_params > 0 && !params))
    return fail(nullptr, SDPGPU_ERR_ARG, "null argument, or more than 256 user parameters");
  *out = nullptr;
  if (desc->lead_time == 2) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "a user functor has one pipeline quantity at most (lead_time 2 is a built-in shape)");
  if (desc->kernel != SDPGPU_KERNEL_AUTO && desc->kernel != SDPGPU_KERNEL_GATHER) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "a user functor runs on the generic kernel only");
  // compile: prelude + the user's three device functions + the engine kernels, strict fp64 (no FMA)
  std::string src = std::string(sdp::kCustomPrelude) + functor_source + "\n" + sdp::kCustomEngine;
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "sdp_custom.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return fail(nullptr, SDPGPU_ERR_DEVICE, "hiprtcCreateProgram failed");
  const std::string np_def = "-DSDP_NP=" + std::to_string(std::max(1, (int)n_params));
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", np_def.c_str()};
  hiprtcResult cr = hiprtcCompileProgram(prog, 6, opts);
  if (cr != HIPRTC_SUCCESS) {
    size_t n = 0;
    (void)hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
    (void)hiprtcDestroyProgram(&prog);
    if (log.size() > 400) log.resize(400);
    return fail(nullptr, SDPGPU_ERR_ARG, "user functor does not compile: %s", log.c_str());
  }
  size_t code_size = 0;
  (void)hiprtcGetCodeSize(prog, &code_size);
  std::vector<char> code(code_size);
  hiprtcResult gr = hiprtcGetCode(prog, code.data());
  (void)hiprtcDestroyProgram(&prog);
  if (gr != HIPRTC_SUCCESS || code.empty()) return fail(nullptr, SDPGPU_ERR_DEVICE, "hiprtcGetCode failed");
  int rc = sdpgpu_create(desc, out);
  if (rc) return rc;
  sdpgpu_handle* h = *out;
  h->custom = true;
  h->custom_code.swap(code);
  h->custom_params.assign(params, params + n_params);
  return SDPGPU_OK;
}

void sdpgpu_destroy(sdpgpu_handle* h) {
  if (!h) return;
  if (h->allocated || h->d_policy || h->d_pmf) {
    if (h->device >= 0) (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
  }
  for (auto& p : h->per) {
    if (p.ev0) (void)hipEventDestroy(p.ev0);
    if (p.ev1) (void)hipEventDestroy(p.ev1);
  }
  if (h->ev_solve0) (void)hipEventDestroy(h->ev_solve0);
  if (h->ev_solve1) (void)hipEventDestroy(h->ev_solve1);
  if (h->d_values && !h->values_external) (void)hipFree(h->d_values);
  if (h->d_policy) (void)hipFree(h->d_policy);
  if (h->d_pmf) (void)hipFree(h->d_pmf);
  if (h->d_reach) (void)hipFree(h->d_reach);
  for (int b = 0; b < 2; ++b) {
    if (h->d_part_val[b]) (void)hipFree(h->d_part_val[b]);
    if (h->d_part_idx[b]) (void)hipFree(h->d_part_idx[b]);
  }
  if (h->d_keys && !h->keys_external) (void)hipFree(h->d_keys);
  if (h->d_chunk_val) (void)hipFree(h->d_chunk_val);
  if (h->d_chunk_idx) (void)hipFree(h->d_chunk_idx);
  if (h->d_jobs) (void)hipFree(h->d_jobs);
  if (h->d_custom_params) (void)hipFree(h->d_custom_params);
  if (h->d_custom_cells) (void)hipFree(h->d_custom_cells);
  if (h->d_custom_err) (void)hipFree(h->d_custom_err);
  if (h->custom_mod) (void)hipModuleUnload(h->custom_mod);
  if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* sdpgpu_last_error(const sdpgpu_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int sdpgpu_set_pmf(sdpgpu_handle* h, int32_t t, const double* demand, const double* prob, int32_t n) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (t < 0 || t >= h->T || !demand || !prob || n < 1) return fail(h, SDPGPU_ERR_ARG, "set_pmf: bad argument (t=%d n=%d)", t, n);
  if (h->allocated) return fail(h, SDPGPU_ERR_STATE, "pmf is frozen once the device tables exist");
  for (int32_t j = 0; j < n; ++j) {
    if (std::fmod(demand[j], h->d.step) != 0) return fail(h, SDPGPU_ERR_ARG, "demand %g of period %d is not a multiple of step", demand[j], t + 1);
    if (j && !(demand[j] > demand[j - 1])) return fail(h, SDPGPU_ERR_ARG, "demands of period %d must be strictly ascending", t + 1);
  }
  try {
    h->pmf_d[t].assign(demand, demand + n);
    h->pmf_p[t].assign(prob, prob + n);
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "out of host memory");
  }
  h->pmf_set[t] = 1;
  h->laid_out = false;
  return SDPGPU_OK;
}

int sdpgpu_set_overhead(sdpgpu_handle* h, int32_t t, double overhead_cost) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (t < 0 || t >= h->T) return fail(h, SDPGPU_ERR_ARG, "set_overhead: t=%d", t);
  h->per[t].overhead = overhead_cost;
  h->per[t].overhead_set = true;
  return SDPGPU_OK;
}

int sdpgpu_set_stream(sdpgpu_handle* h, void* hip_stream) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (h->stream && h->own_stream) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamDestroy(h->stream);
  }
  h->stream = (hipStream_t)hip_stream;  // NULL = the legacy default stream, as in any HIP API
  h->own_stream = false;
  h->stream_given = true;
  return SDPGPU_OK;
}

int sdpgpu_set_profiling(sdpgpu_handle* h, int32_t on) {
  if (!h) return SDPGPU_ERR_ARG;
  h->profiling = on != 0;
  return SDPGPU_OK;
}

int64_t sdpgpu_num_states(const sdpgpu_handle* hc, int32_t period) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h || period < 1 || period > h->T) return -1;
  if (layout(h)) return -1;
  return h->per[period - 1].S;
}

int sdpgpu_slab(const sdpgpu_handle* hc, int32_t period, int64_t* padded, int64_t* lo, int64_t* hi) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h) return SDPGPU_ERR_ARG;
  if (period < 1 || period > h->T) return fail(h, SDPGPU_ERR_ARG, "period %d", period);
  int rc = layout(h);
  if (rc) return rc;
  const PeriodInfo& p = h->per[period - 1];
  if (padded) *padded = p.S_pad;
  if (lo) *lo = p.lo;
  if (hi) *hi = p.hi;
  return SDPGPU_OK;
}

int sdpgpu_grid2(const sdpgpu_handle* hc, int32_t period, double* x_lo, int64_t* nx, int64_t* nc, int64_t* nq1,
                 int64_t* nq2) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h) return SDPGPU_ERR_ARG;
  if (period < 1 || period > h->T) return fail(h, SDPGPU_ERR_ARG, "period %d", period);
  int rc = layout(h);
  if (rc) return rc;
  const Grid& g = h->per[period - 1].g;
  if (x_lo) *x_lo = g.x_lo;
  if (nx) *nx = g.nx;
  if (nc) *nc = g.nc;
  if (nq1) *nq1 = g.nq1;
  if (nq2) *nq2 = g.nq / g.nq1;
  return SDPGPU_OK;
}

int sdpgpu_grid(const sdpgpu_handle* hc, int32_t period, double* x_lo, int64_t* nx, int64_t* nc, int64_t* nq) {
  int64_t q1 = 1, q2 = 1;
  int rc = sdpgpu_grid2(hc, period, x_lo, nx, nc, &q1, &q2);
  if (rc == SDPGPU_OK && nq) *nq = q1 * q2;
  return rc;
}

double sdpgpu_cash_value(const sdpgpu_handle* hc, int64_t ic) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h || !has_cash(h->d.family) || layout(h)) return NAN;
  double k = (double)(h->per[0].g.k_lo + ic);
  return h->d.cash_round_int_div ? k : k / h->d.cash_round_div;
}

int64_t sdpgpu_state_index(const sdpgpu_handle* hc, int32_t period, double x, double cash, double preq) {
  return sdpgpu_state_index2(hc, period, x, cash, preq, 0.0);
}

int64_t sdpgpu_state_index2(const sdpgpu_handle* hc, int32_t period, double x, double cash, double preq, double preq2) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h || period < 1 || period > h->T || layout(h)) return -1;
  const sdpgpu_desc& d = h->d;
  const Grid& g = h->per[period - 1].g;
  double qx = (x - g.x_lo) / d.step;
  int64_t ix = (int64_t)qx;
  if ((double)ix != qx || ix < 0 || ix >= g.nx) return -1;
  int64_t ic = 0, iq = 0;
  if (has_cash(d.family)) {
    int64_t k = d.cash_round_int_div ? (int64_t)cash : java_round(cash * d.cash_round_mult);
    double back = d.cash_round_int_div ? (double)k : (double)k / d.cash_round_div;
    if (back != cash) return -1;
    ic = k - g.k_lo;
    if (ic < 0 || ic >= g.nc) return -1;
  }
  if (has_preq(d.family)) {
    double qq = preq / d.step;
    iq = (int64_t)qq;
    if ((double)iq != qq || iq < 0 || iq >= g.nq1) return -1;
  }
  if (d.lead_time == 2) {
    double qq = preq2 / d.step;
    int64_t iq2 = (int64_t)qq;
    if ((double)iq2 != qq || iq2 < 0 || iq2 >= g.nq / g.nq1) return -1;
    iq += iq2 * g.nq1;
  } else if (preq2 != 0.0) {
    return -1;
  }
  return (iq * g.nx + ix) * g.nc + ic;
}

size_t sdpgpu_values_bytes(const sdpgpu_handle* hc) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h || layout(h)) return 0;
  return std::max<size_t>(h->values_elems, 1) * sizeof(double);
}

int sdpgpu_attach_values(sdpgpu_handle* h, void* device_ptr, size_t bytes) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (h->allocated) return fail(h, SDPGPU_ERR_STATE, "attach_values must precede the first run");
  int rc = layout(h);
  if (rc) return rc;
  if (!device_ptr || bytes < std::max<size_t>(h->values_elems, 1) * sizeof(double))
    return fail(h, SDPGPU_ERR_ARG, "attach_values: need %zu bytes", std::max<size_t>(h->values_elems, 1) * sizeof(double));
  h->d_values = (double*)device_ptr;
  h->values_external = true;
  return SDPGPU_OK;
}

void* sdpgpu_values_device_ptr(sdpgpu_handle* h, int32_t period) {
  if (!h || period < 1 || period > h->T) return nullptr;
  if (allocate(h)) return nullptr;
  if (flush_api(h)) return nullptr;
  return h->d_values + h->per[period - 1].v_off;
}

int sdpgpu_run_period(sdpgpu_handle* h, int32_t period) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  try {
    int rc = run_period_impl(h, period);
    if (rc == SDPGPU_OK) count_cells(h, period);
    return rc;
  } catch (const std::exception& e) {
    return fail(h, SDPGPU_ERR_ARG, "exception: %s", e.what());
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "unknown exception");
  }
}

int sdpgpu_run_period_part(sdpgpu_handle* h, int32_t period, int32_t part) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (part < SDPGPU_PART_ALL || part > SDPGPU_PART_BOUNDARY) return fail(h, SDPGPU_ERR_ARG, "run_period_part: part %d", part);
  try {
    int rc = run_period_impl(h, period, part);
    if (rc == SDPGPU_OK && part != SDPGPU_PART_INTERIOR) count_cells(h, period);
    return rc;
  } catch (const std::exception& e) {
    return fail(h, SDPGPU_ERR_ARG, "exception: %s", e.what());
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "unknown exception");
  }
}

int sdpgpu_solve(sdpgpu_handle* h, int32_t sync) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (h->d.world_size != 1) return fail(h, SDPGPU_ERR_STATE, "sdpgpu_solve needs world_size 1; sharded handles run period by period with an all-gather in between");
  try {
    int rc = allocate(h);
    if (rc) return rc;
    std::fill(h->period_done.begin(), h->period_done.end(), 0);
    HIP_TRY(h, hipEventRecord(h->ev_solve0, h->stream));
    for (int period = h->T; period >= 1; --period) {
      rc = run_period_impl(h, period);
      if (rc) return rc;
    }
    rc = flush_api(h);
    if (rc) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_solve1, h->stream));
    h->solve_timed = true;
    for (int period = 1; period <= h->T; ++period)
      if (h->per[period - 1].cells_all == 0) count_cells(h, period);
    if (sync) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SDPGPU_OK;
  } catch (const std::exception& e) {
    return fail(h, SDPGPU_ERR_ARG, "exception: %s", e.what());
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "unknown exception");
  }
}

size_t sdpgpu_keys_bytes(const sdpgpu_handle* hc) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h || !keys_needed(h)) return 0;
  size_t stride = 0;
  for (const PeriodInfo& q : h->per) stride = std::max<size_t>(stride, (size_t)q.S_pad);
  return (size_t)h->T * stride * sizeof(unsigned long long);
}

int sdpgpu_attach_keys(sdpgpu_handle* h, void* device_ptr, size_t bytes) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (h->d_keys) return fail(h, SDPGPU_ERR_STATE, "attach_keys must precede the first run");
  size_t need = sdpgpu_keys_bytes(h);
  if (!device_ptr || bytes < need) return fail(h, SDPGPU_ERR_ARG, "attach_keys: need %zu bytes", need);
  h->d_keys = (unsigned long long*)device_ptr;
  h->keys_external = true;
  return SDPGPU_OK;
}

void* sdpgpu_exchange_ptr(sdpgpu_handle* h, int32_t period) {
  if (!h || period < 1 || period > h->T) return nullptr;
  if (allocate(h)) return nullptr;
  if (h->pending_chunks[period - 1] > 0) return h->d_keys + (size_t)(period - 1) * h->key_stride;
  return h->d_values + h->per[period - 1].v_off;
}

int sdpgpu_finalize(sdpgpu_handle* h) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (!h->allocated) return SDPGPU_OK;
  return flush_api(h);
}

int sdpgpu_synchronize(sdpgpu_handle* h) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (!h->allocated) return SDPGPU_OK;
  int rc = ensure_device(h);
  if (rc) return rc;
  rc = flush_api(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return SDPGPU_OK;
}

int sdpgpu_values(sdpgpu_handle* h, int32_t period, double* out, int64_t n) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (period < 1 || period > h->T || !out) return fail(h, SDPGPU_ERR_ARG, "values: bad argument");
  if (!h->allocated || !h->period_done[period - 1]) return fail(h, SDPGPU_ERR_STATE, "V_%d has not been computed", period);
  const PeriodInfo& p = h->per[period - 1];
  if (n < 0 || n > p.S) return fail(h, SDPGPU_ERR_ARG, "values: n=%lld > %lld states", (long long)n, (long long)p.S);
  int rc = ensure_device(h);
  if (rc) return rc;
  rc = flush_api(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(out, h->d_values + p.v_off, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return SDPGPU_OK;
}

int sdpgpu_policy(sdpgpu_handle* h, int32_t period, int32_t* out, int64_t lo, int64_t n) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (period < 1 || period > h->T || !out) return fail(h, SDPGPU_ERR_ARG, "policy: bad argument");
  if (!h->allocated || !h->policy_done[period - 1]) return fail(h, SDPGPU_ERR_STATE, "period %d has not been computed", period);
  const PeriodInfo& p = h->per[period - 1];
  if (lo < p.lo || n < 0 || lo + n > p.hi) return fail(h, SDPGPU_ERR_ARG, "policy: [%lld, %lld) outside this rank's slab [%lld, %lld)", (long long)lo, (long long)(lo + n), (long long)p.lo, (long long)p.hi);
  int rc = ensure_device(h);
  if (rc) return rc;
  rc = flush_api(h);
  if (rc) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(out, h->d_policy + p.pol_off + (lo - p.lo), (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return SDPGPU_OK;
}

int sdpgpu_eval_states(sdpgpu_handle* h, int32_t period, int64_t n, const double* x, const double* cash,
                       const double* preq, double* out_value, int32_t* out_action_index) {
  return sdpgpu_eval_states2(h, period, n, x, cash, preq, nullptr, out_value, out_action_index);
}

int sdpgpu_eval_states2(sdpgpu_handle* h, int32_t period, int64_t n, const double* x, const double* cash,
                        const double* preq, const double* preq2, double* out_value, int32_t* out_action_index) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (period < 1 || period > h->T || n < 0 || !x || !out_value || !out_action_index) return fail(h, SDPGPU_ERR_ARG, "eval_states: bad argument");
  if (has_cash(h->d.family) && !cash) return fail(h, SDPGPU_ERR_ARG, "eval_states: cash array required");
  if (has_preq(h->d.family) && !preq) return fail(h, SDPGPU_ERR_ARG, "eval_states: preq array required");
  int rc = allocate(h);
  if (rc) return rc;
  if (period < h->T && !h->period_done[period]) return fail(h, SDPGPU_ERR_STATE, "V_%d has not been computed", period + 1);
  if (n == 0) return SDPGPU_OK;
  rc = ensure_device(h);
  if (rc) return rc;
  rc = flush_api(h);
  if (rc) return rc;
  double* d_in = nullptr;
  double* d_val = nullptr;
  int32_t* d_act = nullptr;
  size_t nn = (size_t)n;
  if (h->d.lead_time != 2) preq2 = nullptr;
  HIP_TRY(h, hipMalloc((void**)&d_in, 4 * nn * sizeof(double)));
  hipError_t e = hipMalloc((void**)&d_val, nn * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&d_act, nn * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpy(d_in, x, nn * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess && cash) e = hipMemcpy(d_in + nn, cash, nn * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess && preq) e = hipMemcpy(d_in + 2 * nn, preq, nn * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess && preq2) e = hipMemcpy(d_in + 3 * nn, preq2, nn * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    DevParams P = make_params(h, period);
    const PeriodInfo& p = h->per[period - 1];
    const double* v_next = period < h->T ? h->d_values + h->per[period].v_off : nullptr;
    const double* pd = h->d_pmf + p.pmf_off;
    sdp::QueryStates q{d_in, cash ? d_in + nn : nullptr, preq ? d_in + 2 * nn : nullptr, preq2 ? d_in + 3 * nn : nullptr};
    if (h->custom)
      e = launch_custom_period(h, period, v_next, d_val, d_act, 0, n, q.x, q.cash, q.preq, false);
    else
      e = launch_gather<true>(P, v_next, d_val, d_act, pd, pd + p.nD, 0, n, q, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(out_value, d_val, nn * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_action_index, d_act, nn * sizeof(int32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  (void)hipFree(d_val);
  (void)hipFree(d_act);
  if (e != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "eval_states: %s", hipGetErrorString(e));
  return custom_check(h);
}

int sdpgpu_reachable(sdpgpu_handle* h, int32_t period, uint8_t* out, int64_t n) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (period < 1 || period > h->T || !out) return fail(h, SDPGPU_ERR_ARG, "reachable: bad argument");
  int rc = compute_reachable(h);
  if (rc) return rc;
  const PeriodInfo& p = h->per[period - 1];
  if (n < 0 || n > p.S) return fail(h, SDPGPU_ERR_ARG, "reachable: n=%lld > %lld states", (long long)n, (long long)p.S);
  HIP_TRY(h, hipMemcpy(out, h->d_reach + h->reach_off[period - 1], (size_t)n, hipMemcpyDeviceToHost));
  return SDPGPU_OK;
}

int sdpgpu_simulate(sdpgpu_handle* h, int64_t n_paths, const double* demand, const double* discount, double ini_x,
                    double ini_cash, double ini_preq, double* out_sum, uint8_t* out_valid) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (n_paths < 0 || !demand || !discount || !out_sum || !out_valid) return fail(h, SDPGPU_ERR_ARG, "simulate: bad argument");
  if (h->d.world_size != 1) return fail(h, SDPGPU_ERR_STATE, "simulate needs the whole policy on one GPU (world_size 1)");
  if (h->custom) return fail(h, SDPGPU_ERR_UNSUPPORTED, "simulate: a user functor's lambdas live on the host; roll the policy tables forward there");
  if (!h->allocated) return fail(h, SDPGPU_ERR_STATE, "simulate: nothing has been solved");
  for (int t = 0; t < h->T; ++t)
    if (!h->policy_done[t]) return fail(h, SDPGPU_ERR_STATE, "simulate: period %d has not been computed", t + 1);
  if (n_paths == 0) return SDPGPU_OK;
  int rc = ensure_device(h);
  if (rc) return rc;
  rc = flush_api(h);
  if (rc) return rc;
  const int T = h->T;
  if (!has_cash(h->d.family)) ini_cash = 0;
  if (!has_preq(h->d.family)) ini_preq = 0;
  double ini_preq2 = h->d.lead_time == 2 ? h->d.ini_preq2 : 0.0;
  int64_t idx0 = sdpgpu_state_index2(h, 1, ini_x, ini_cash, ini_preq, ini_preq2);
  int32_t first_k = 0;
  if (idx0 < 0) {
    double v;
    rc = sdpgpu_eval_states2(h, 1, 1, &ini_x, &ini_cash, &ini_preq, &ini_preq2, &v, &first_k);
    if (rc) return rc;
  }
  try {
    std::vector<sdp::SimPeriod> per((size_t)T);
    for (int t = 0; t < T; ++t) {
      per[t].P = make_params(h, t + 1);
      per[t].pol_off = (int64_t)h->per[t].pol_off - h->per[t].lo;
      per[t].n_states = h->per[t].S;
    }
    const size_t nn = (size_t)n_paths;
    sdp::SimPeriod* d_per = nullptr;
    double *d_dem = nullptr, *d_disc = nullptr, *d_sum = nullptr;
    uint8_t* d_valid = nullptr;
    hipError_t e = hipMalloc((void**)&d_per, per.size() * sizeof(sdp::SimPeriod));
    if (e == hipSuccess) e = hipMalloc((void**)&d_dem, nn * T * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_disc, (size_t)T * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_sum, nn * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&d_valid, nn);
    if (e == hipSuccess) e = hipMemcpy(d_per, per.data(), per.size() * sizeof(sdp::SimPeriod), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_dem, demand, nn * T * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_disc, discount, (size_t)T * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      sdp::StateT ini{ini_x, ini_cash, ini_preq, ini_preq2};
      dim3 grid((unsigned)((n_paths + 255) / 256));
#define SDP_SIM(F)                                                                                                   \
  case F:                                                                                                            \
    hipLaunchKernelGGL((sdp::simulate_kernel<F>), grid, dim3(256), 0, h->stream, d_per, T, h->d_policy, d_dem, d_disc, \
                       n_paths, idx0, ini, (int)first_k, d_sum, d_valid);                                             \
    break;
      switch (h->d.family) {
        SDP_SIM(sdp::FAM_BACKORDER)
        SDP_SIM(sdp::FAM_LEADTIME)
        SDP_SIM(sdp::FAM_CASH)
        SDP_SIM(sdp::FAM_OVERDRAFT)
        SDP_SIM(sdp::FAM_CASH_LEADTIME)
        SDP_SIM(sdp::FAM_SURVIVAL)
      }
#undef SDP_SIM
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(out_sum, d_sum, nn * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_valid, d_valid, nn, hipMemcpyDeviceToHost);
    (void)hipFree(d_per);
    (void)hipFree(d_dem);
    (void)hipFree(d_disc);
    (void)hipFree(d_sum);
    (void)hipFree(d_valid);
    if (e != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "simulate: %s", hipGetErrorString(e));
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "simulate: out of host memory");
  }
  return SDPGPU_OK;
}

int sdpgpu_stats_get(sdpgpu_handle* h, sdpgpu_stats* out) {
  if (!h || !out) return SDPGPU_ERR_ARG;
  h->err.clear();
  std::memset(out, 0, sizeof *out);
  if (layout(h)) return SDPGPU_ERR_STATE;
  for (int t = 0; t < h->T; ++t) {
    const PeriodInfo& p = h->per[t];
    out->states_total += p.S;
    out->cells_evaluated += p.cells_rank;
    out->cells_all_ranks += p.cells_all;
    if (h->period_done[t]) out->periods_run++;
  }
  out->kernel_used = h->per[0].kernel_used;
  if (!h->custom && h->d.family == SDPGPU_FAMILY_BACKORDER && h->per[0].kernel_used == SDPGPU_KERNEL_WINDOW &&
      window_eligible(h, 1)) {
    const WinPlan pl = plan_window(h, 1, h->per[0].lo, h->per[0].hi);
    out->window_r = pl.R;
    out->window_s = pl.S;
  }
  if (h->allocated) {
    (void)ensure_device(h);
    if (h->custom && h->d_custom_cells && hipStreamSynchronize(h->stream) == hipSuccess) {
      // the action count of a user functor is only known on the device: the kernel counted its cells
      std::vector<unsigned long long> c((size_t)h->T, 0);
      if (hipMemcpy(c.data(), h->d_custom_cells, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
        out->cells_evaluated = 0;
        for (int t = 0; t < h->T; ++t)
          if (h->period_done[t]) out->cells_evaluated += (int64_t)c[(size_t)t];
        out->cells_all_ranks = h->d.world_size == 1 ? out->cells_evaluated : 0;
      }
    }
    if (h->solve_timed && hipStreamSynchronize(h->stream) == hipSuccess) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, h->ev_solve0, h->ev_solve1) == hipSuccess) out->solve_ms = ms;
    }
    for (int t = 0; t < h->T; ++t) {
      const PeriodInfo& p = h->per[t];
      if (p.timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.ev0, p.ev1) == hipSuccess) out->kernel_ms_sum += ms;
      }
    }
  }
  return SDPGPU_OK;
}

double sdpgpu_period_ms(sdpgpu_handle* h, int32_t period) {
  if (!h || period < 1 || period > h->T) return -1;
  const PeriodInfo& p = h->per[period - 1];
  if (!p.timed) return -1;
  (void)ensure_device(h);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
  float ms = 0;
  if (hipEventElapsedTime(&ms, p.ev0, p.ev1) != hipSuccess) return -1;
  return ms;
}

}  // extern "C"

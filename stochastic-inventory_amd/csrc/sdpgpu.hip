// sdpgpu.hip -- C-ABI implementation (include/sdpgpu.h) of the MI355X SDP engine: descriptor validation,
// per-period grid layout, device tables, period dispatch and the entry points themselves.  The kernels live in
// the other translation units (see sdpgpu_internal.hpp).  gfx950 only; there is NO CPU fallback anywhere in this
// library -- without a HIP device every compute entry point fails with SDPGPU_ERR_DEVICE.
#include <hip/hiprtc.h>

#include "sdpgpu_internal.hpp"
#include "sdp_gather.hpp"      // SimPeriod (the rollout's per-period parameter block)
#include "sdp_custom_src.hpp"  // device text of the user-functor engine

namespace sdpgpu_detail {

thread_local std::string g_create_error;

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

int validate(const sdpgpu_desc& d) {
  if (d.abi_version != SDPGPU_ABI_VERSION) return fail(nullptr, SDPGPU_ERR_ARG, "abi_version %d != %d", d.abi_version, SDPGPU_ABI_VERSION);
  if (d.family < 1 || d.family > 7) return fail(nullptr, SDPGPU_ERR_ARG, "unknown family %d", d.family);
  if (d.direction != SDPGPU_MIN && d.direction != SDPGPU_MAX) return fail(nullptr, SDPGPU_ERR_ARG, "bad direction %d", d.direction);
  if (d.periods < 1 || d.periods > 4096) return fail(nullptr, SDPGPU_ERR_ARG, "periods %d out of range", d.periods);
  if (!is_pow2_int(d.step)) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "step %g: only power-of-two integer step sizes are supported (every in-scope driver uses 1)", d.step);
  if (!(d.max_order_quantity >= 0) || d.max_order_quantity > 1e9) return fail(nullptr, SDPGPU_ERR_ARG, "max_order_quantity %g", d.max_order_quantity);
  if (d.world_size < 1 || d.rank < 0 || d.rank >= d.world_size) return fail(nullptr, SDPGPU_ERR_ARG, "rank %d / world_size %d", d.rank, d.world_size);
  if (d.clamp_inventory) {
    if (!(d.max_inventory >= d.min_inventory)) return fail(nullptr, SDPGPU_ERR_ARG, "max_inventory < min_inventory");
    if (std::fmod(d.min_inventory, d.step) != 0 || std::fmod(d.max_inventory, d.step) != 0) return fail(nullptr, SDPGPU_ERR_ARG, "inventory bounds must be multiples of step");
  } else {
    if (d.family != SDPGPU_FAMILY_BACKORDER && d.family != SDPGPU_FAMILY_LEADTIME && d.family != SDPGPU_FAMILY_STAFF) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "clamp_inventory = 0 only for the backorder / lead-time / staff families");
    if (std::fmod(d.ini_inventory, d.step) != 0) return fail(nullptr, SDPGPU_ERR_ARG, "ini_inventory must be a multiple of step");
  }
  if (has_cash(d.family)) {
    if (!(d.max_cash >= d.min_cash)) return fail(nullptr, SDPGPU_ERR_ARG, "max_cash < min_cash");
    if (!(d.cash_round_mult > 0) || !(d.cash_round_div > 0)) return fail(nullptr, SDPGPU_ERR_ARG, "cash rounding factors must be positive");
    if (!d.cash_round_int_div && d.cash_round_mult != d.cash_round_div) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "Math.round(c*m)/d with m != d does not map onto a uniform grid");
    if (d.cash_round_int_div && d.cash_round_div != std::floor(d.cash_round_div)) return fail(nullptr, SDPGPU_ERR_ARG, "integer cash divisor must be integral");
    if (std::fabs(d.min_cash * d.cash_round_mult) > 2.0e9 || std::fabs(d.max_cash * d.cash_round_mult) > 2.0e9) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "cash keys exceed 32 bits");
  }
  if (d.cash_formula < 0 || d.cash_formula > 2) return fail(nullptr, SDPGPU_ERR_ARG, "cash_formula %d (0 CashConstraint, 1 CashConstraintTesting, 2 CashConstraintXR)", d.cash_formula);
  if (d.cash_formula == 2) {  // sdp.cash.CashRecursionXR over cash.singleItem.CashConstraintXR's lambdas: state (x, R)
    if (d.family != SDPGPU_FAMILY_CASH) return fail(nullptr, SDPGPU_ERR_ARG, "cash_formula 2 (the (x, R) state of CashConstraintXR) belongs to the CASH family");
    if (d.step != 1) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "CashConstraintXR's action list has (int)(maxY - x) + 1 entries whatever the step (CashConstraintXR.java:86-87): step must be 1");
    if (d.penalty_cost != 0) return fail(nullptr, SDPGPU_ERR_ARG, "CashConstraintXR's immediate value has no end-cash penalty: penalty_cost must be 0");
    if (!(d.unit_order_cost > 0)) return fail(nullptr, SDPGPU_ERR_ARG, "CashConstraintXR divides by variCost: unit_order_cost must be positive");
    if (d.min_inventory != std::floor(d.min_inventory)) return fail(nullptr, SDPGPU_ERR_ARG, "inventory bounds must be integers");
  }
  if (d.lead_time < 0 || d.lead_time > 2) return fail(nullptr, SDPGPU_ERR_ARG, "lead_time %d (0/1 = the reference's lead time 1, 2 = two-stage pipeline)", d.lead_time);
  if (d.lead_time == 2) {
    if (d.family != SDPGPU_FAMILY_LEADTIME) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "lead_time 2 exists for the LEADTIME family only");
    if (!d.clamp_inventory) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "lead_time 2 needs clamp_inventory = 1");
  }
  if (d.family == SDPGPU_FAMILY_STAFF) {
    if (d.direction != SDPGPU_MIN) return fail(nullptr, SDPGPU_ERR_ARG, "StaffRecursion is MIN only (StaffRecursion.java:89,110)");
    if (d.step != 1) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "the staff family counts heads: step must be 1");
    if (d.max_order_quantity != std::floor(d.max_order_quantity)) return fail(nullptr, SDPGPU_ERR_ARG, "maxHireNum must be an integer");
    if (d.clamp_inventory ? (d.min_inventory < 0 || d.max_inventory > 1e9) : (d.ini_inventory < 0 || d.ini_inventory > 1e9)) return fail(nullptr, SDPGPU_ERR_ARG, "staff numbers must lie in 0 .. 1e9");
    if (d.clamp_inventory && (d.ini_inventory < d.min_inventory || d.ini_inventory > d.max_inventory || d.ini_inventory != std::floor(d.ini_inventory))) return fail(nullptr, SDPGPU_ERR_ARG, "iniStaffNum must be an integer inside [minX, maxX]");
  }
  if (d.family == SDPGPU_FAMILY_CASH_LEADTIME && d.step != 1)  // (orders k * step, k <= (int)maxQ, would leave the pipeline axis of (int)(maxQ / step) + 1 planes)
    return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "SingleProductLeadtime's action list has (int)maxOrderQuantity + 1 entries whatever the step (SingleProductLeadtime.java:74-78): step must be 1");
  if ((d.family == SDPGPU_FAMILY_LEADTIME) && d.direction != SDPGPU_MIN) return fail(nullptr, SDPGPU_ERR_ARG, "LeadtimeRecursion is MIN only (LeadtimeRecursion.java:52,66)");
  if ((d.family == SDPGPU_FAMILY_SURVIVAL) && d.direction != SDPGPU_MAX) return fail(nullptr, SDPGPU_ERR_ARG, "getSurvProb maximises (RiskRecursion.java:70,101)");
  if ((d.family == SDPGPU_FAMILY_CASH_LEADTIME) && d.direction != SDPGPU_MAX) return fail(nullptr, SDPGPU_ERR_ARG, "CashLeadtimeRecursion is MAX only (CashLeadtimeRecursion.java:53,70)");
  return SDPGPU_OK;
}

int32_t full_action_count(const sdpgpu_desc& d) {
  // `DoubleStream.iterate(0, i -> i + stepSize).limit((int) maxQ + 1)` -- the cash drivers' lists have (int)maxQ + 1 entries whatever the
  // step (CashConstraint.java:99, cashSurvival.java:108, CashOverdraft.java:74, SingleProductLeadtime.java:76); for CASH / SURVIVAL
  // this is the longest list any state has.  F1 / F2: `new double[(int)(maxOrderQuantity / stepSize) + 1]` (CLSPTesting.java:79).
  if (d.family == SDPGPU_FAMILY_OVERDRAFT || d.family == SDPGPU_FAMILY_CASH_LEADTIME || d.family == SDPGPU_FAMILY_CASH || d.family == SDPGPU_FAMILY_SURVIVAL)
    return java_d2i(d.max_order_quantity) + 1;
  return java_d2i(d.max_order_quantity / d.step) + 1;
}

// Per-period grids.  Clamped families: one fixed box.  Unclamped (Leadtime.java:65-66): the box
// of period t+1 is the image of the period-t box under every action and demand.
int layout(sdpgpu_handle* h) {
  if (h->laid_out) return SDPGPU_OK;
  const sdpgpu_desc& d = h->d;
  const bool staff = d.family == SDPGPU_FAMILY_STAFF;
  for (int t = 0; t < h->T; ++t)
    if (!h->pmf_set[t]) return fail(h, SDPGPU_ERR_STATE, staff ? "level pmf of period %d not set (sdpgpu_set_level_pmf)" : "pmf of period %d not set", t + 1);
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
    p.cells_counted = false;
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
    if (!staff) {
      const std::vector<double>& dv = h->pmf_d[t];
      const double span = (dv.back() - dv.front()) / d.step;  // demands are multiples of step (set_pmf)
      const int64_t dense = (int64_t)span + 1;
      p.nD_win = 0;
      p.pmf_win_off = p.pmf_off + (size_t)p.nD;
      if (dense == p.nD) {
        p.nD_win = p.nD;
      } else if (dense <= 3 * (int64_t)p.nD + 16 && dense <= 4000) {  // beyond ~3x padding the generic kernel wins
        p.nD_win = (int32_t)dense;
        p.pmf_win_off = pmf_off;
        pmf_off += (size_t)dense + kPmfPad;
      }
    }
    p.v_off = v_off;
    p.pol_off = pol_off;
    pol_off += (size_t)slab;
    if (d.store_all_values) v_off += (size_t)p.S_pad;
    s_pad_max = std::max(s_pad_max, p.S_pad);
    if (!p.overhead_set) p.overhead = d.overhead_cost;
    if (staff) {
      if (p.overhead != std::floor(p.overhead) || std::fabs(p.overhead) > 1e9) return fail(h, SDPGPU_ERR_ARG, "minStaffNum of period %d (sdpgpu_set_overhead) must be an integer", t + 1);
      if (!d.clamp_inventory) {  // successors x + a - j with 0 <= j <= min(x + a, longest row - 1): never below 0
        lo = std::max(0.0, lo - (double)(h->lvl_maxj[t] - 1));
        hi = hi + (double)(h->n_actions_full - 1);
      }
    } else if (!d.clamp_inventory) {
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
  for (PeriodInfo& q : h->per) q.win_plan.valid = false;  // (slabs and pmf widths may have changed)
  h->laid_out = true;
  return SDPGPU_OK;
}


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
  for (auto& p : h->per) pmf_elems += 2 * (size_t)p.nD + kPmfPad + (p.nD_win > p.nD ? (size_t)p.nD_win + kPmfPad : 0);
  HIP_TRY(h, hipMalloc((void**)&h->d_pmf, std::max<size_t>(pmf_elems, 1) * sizeof(double)));
  std::vector<double> host(pmf_elems, 0.0);
  for (int t = 0; t < h->T; ++t) {
    const PeriodInfo& p = h->per[t];
    std::memcpy(&host[p.pmf_off], h->pmf_d[t].data(), (size_t)p.nD * sizeof(double));
    std::memcpy(&host[p.pmf_off + p.nD], h->pmf_p[t].data(), (size_t)p.nD * sizeof(double));
    if (p.nD_win > p.nD)  // unit-stride layout of a support with gaps
      for (int j = 0; j < p.nD; ++j)
        host[p.pmf_win_off + (size_t)((h->pmf_d[t][(size_t)j] - h->pmf_d[t][0]) / h->d.step)] = h->pmf_p[t][(size_t)j];
  }
  HIP_TRY(h, hipMemcpy(h->d_pmf, host.data(), pmf_elems * sizeof(double), hipMemcpyHostToDevice));
  for (int t = 0; t < h->T; ++t) {
    if (h->counts[(size_t)t].empty()) continue;
    if ((int64_t)h->counts[(size_t)t].size() != h->per[t].S)
      return fail(h, SDPGPU_ERR_ARG, "action counts of period %d: %zu entries for %lld grid states", t + 1, h->counts[(size_t)t].size(), (long long)h->per[t].S);
    HIP_TRY(h, hipMalloc((void**)&h->d_counts[(size_t)t], h->counts[(size_t)t].size() * sizeof(int32_t)));
    HIP_TRY(h, hipMemcpy(h->d_counts[(size_t)t], h->counts[(size_t)t].data(), h->counts[(size_t)t].size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  HIP_TRY(h, hipEventCreate(&h->ev_solve0));
  HIP_TRY(h, hipEventCreate(&h->ev_solve1));
  if (h->d.family == SDPGPU_FAMILY_STAFF) {
    rc = staff_upload(h);
    if (rc) return rc;
  }
  if (h->custom) {
    HIP_TRY(h, hipModuleLoadData(&h->custom_mod, h->custom_code.data()));
    HIP_TRY(h, hipModuleGetFunction(&h->custom_period[0], h->custom_mod, "sdp_custom_period_64"));
    HIP_TRY(h, hipModuleGetFunction(&h->custom_period[1], h->custom_mod, "sdp_custom_period_16"));
    HIP_TRY(h, hipModuleGetFunction(&h->custom_period[2], h->custom_mod, "sdp_custom_period_4"));
    HIP_TRY(h, hipModuleGetFunction(&h->custom_reach, h->custom_mod, "sdp_custom_reach"));
    HIP_TRY(h, hipMalloc((void**)&h->d_custom_params, std::max<size_t>(h->custom_params.size(), 1) * sizeof(double)));
    if (!h->custom_params.empty())
      HIP_TRY(h, hipMemcpy(h->d_custom_params, h->custom_params.data(), h->custom_params.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMalloc((void**)&h->d_custom_cells, (size_t)h->T * sizeof(unsigned long long)));
    HIP_TRY(h, hipMemset(h->d_custom_cells, 0, (size_t)h->T * sizeof(unsigned long long)));
    HIP_TRY(h, hipMalloc((void**)&h->d_custom_err, sizeof(int)));
    HIP_TRY(h, hipMemset(h->d_custom_err, 0, sizeof(int)));
    if (h->level_shape) {
      HIP_TRY(h, hipModuleGetFunction(&h->custom_tabulate, h->custom_mod, "sdp_custom_tabulate"));
      int rc_t = fill_level_tables(h);
      if (rc_t) return rc_t;
    }
  }
  h->allocated = true;
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
  P.counts = h->d_counts[(size_t)period - 1];
  return P;
}

// cells of one period: sum over states of nA(s) * D  (host arithmetic, no device work)
void count_cells(sdpgpu_handle* h, int period) {
  PeriodInfo& p = h->per[period - 1];
  if (h->custom && !h->level_shape) return;  // counted on the device (sdpgpu_stats_get); the level shape offers every order everywhere
  if (p.cells_counted) return;  // once per handle and period: the sums below are O(S) host loops for some families
  p.cells_counted = true;
  const sdpgpu_desc& d = h->d;
  int64_t nD = p.nD;
  auto range_cells = [&](int64_t lo, int64_t hi) -> int64_t {
    if (hi <= lo) return 0;
    if (!h->counts[(size_t)period - 1].empty()) {  // the caller's own list lengths
      int64_t total = 0;
      const std::vector<int32_t>& c = h->counts[(size_t)period - 1];
      for (int64_t i = lo; i < hi; ++i) total += c[(size_t)i];
      return total * nD;
    }
    if (d.family == SDPGPU_FAMILY_STAFF) return staff_cells(h, period, lo, hi);
    if (d.family != SDPGPU_FAMILY_CASH && d.family != SDPGPU_FAMILY_SURVIVAL) {
      int64_t nA = h->n_actions_full;
      if (d.family == SDPGPU_FAMILY_CASH_LEADTIME && d.zero_order_last_period && period == h->T) nA = 1;
      return (hi - lo) * nA * nD;
    }
    int64_t nc = p.g.nc;
    if (d.cash_formula == 2) {  // (x, R): the count depends on both coordinates (CashConstraintXR.java:84-88)
      int64_t total = 0;
      for (int64_t idx = lo; idx < hi; ++idx) {
        const double x = p.g.x_lo + (double)(idx / nc) * d.step;
        const double k = (double)(p.g.k_lo + idx % nc);
        const double cash = d.cash_round_int_div ? k : k / d.cash_round_div;
        const double R = cash + d.unit_order_cost * x;
        const double ry = R / d.unit_order_cost;
        const double maxY = ry < x ? x : ry;
        total += (int64_t)java_d2i(maxY - x) + 1;
      }
      return total * nD;
    }
    // F3: nA depends on the cash index only; flat = ix * nc + ic
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

// part: SDPGPU_PART_ALL, or the two halves a sharded caller overlaps with the all-gather of V_{t+1}:
// INTERIOR = the states whose cells read only THIS rank's slab of V_{t+1}, BOUNDARY = the rest.
// range_lo/range_hi >= 0: sdpgpu_run_period_range -- the states [range_lo, range_hi) instead of this rank's slab
// (F1 window kernel only: the one family whose dependency footprint is bounded).
void graph_drop(sdpgpu_handle* h) {
  if (h->sweep_exec) (void)hipGraphExecDestroy(h->sweep_exec);
  if (h->sweep_graph) (void)hipGraphDestroy(h->sweep_graph);
  h->sweep_exec = nullptr;
  h->sweep_graph = nullptr;
  // back to "one eager sweep first": a capture must START from the state a complete sweep leaves behind (nothing pending,
  // key rows to be reset at period T), which only a whole eager sweep re-establishes
  if (h->graph_state > 0) h->graph_state = 0;
}

int run_period_impl(sdpgpu_handle* h, int period, int part, int64_t range_lo, int64_t range_hi) {
  int rc = allocate(h);
  if (rc) return rc;
  // a period run outside sdpgpu_solve (stepwise callers, sharded sweeps) leaves key rows / pending rows in a state the
  // captured sweep did not start from
  if (!h->in_solve && h->sweep_exec) graph_drop(h);
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
  const bool ranged = range_lo >= 0;
  if (ranged) {
    const bool f1_window = f1_like(h) && h->d.family == SDPGPU_FAMILY_BACKORDER &&
                           (h->d.kernel == SDPGPU_KERNEL_AUTO || h->d.kernel == SDPGPU_KERNEL_WINDOW) &&
                           window_eligible(h, period);
    if (!f1_window) return fail(h, SDPGPU_ERR_UNSUPPORTED, "run_period_range: only the backorder family on the window kernel has a bounded footprint");
    if (range_lo > range_hi || range_hi > p.S || range_lo < p.lo - h->halo || range_hi > p.hi + h->halo)
      return fail(h, SDPGPU_ERR_ARG, "run_period_range: [%lld, %lld) leaves the slab [%lld, %lld) widened by the halo %lld",
                  (long long)range_lo, (long long)range_hi, (long long)p.lo, (long long)p.hi, (long long)h->halo);
  }
  if (h->d.family == SDPGPU_FAMILY_STAFF) {
    if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;  // the footprint spans the whole table (turnover up to all staff)
    hipError_t es = launch_staff(h, period, v_next, v_cur, pol, p.lo, p.hi, h->stream);
    if (es != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "period %d staff kernel: %s", period, hipGetErrorString(es));
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
  // user lambdas of the level shape run on the F1 window kernel wherever the built-in family would (tables in place of the
  // built-in costs); anything else about them -- kernel = GATHER, a period no window plan fits -- is the generic loop below
  const bool level_win = h->level_shape && h->d.kernel != SDPGPU_KERNEL_GATHER && window_eligible(h, period) &&
                         ((h->win_r || h->win_s || h->win_nch) || plan_window(h, period, p.lo, p.hi).R);
  if (h->custom && !level_win) {
    if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;  // no bounded footprint is known for user lambdas
    if ((size_t)p.nD * 16 + 4 * 64 * 12 > kLdsLegacy)  // (a hipRTC module function keeps the 64 KiB launch limit)
      return fail(h, SDPGPU_ERR_UNSUPPORTED, "a user functor takes pmfs of at most 3900 points (period %d has %d)", period, p.nD);
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
  const bool own_counts = !h->counts[(size_t)period - 1].empty();  // the specialised kernels assume the family's action rule
  if (own_counts && (h->d.kernel == SDPGPU_KERNEL_SEPARABLE || h->d.kernel == SDPGPU_KERNEL_WINDOW || ranged))
    return fail(h, SDPGPU_ERR_UNSUPPORTED, "period %d has caller-supplied action counts: only the generic kernel evaluates those", period);
  if (h->d.kernel == SDPGPU_KERNEL_SEPARABLE && h->d.family == SDPGPU_FAMILY_LEADTIME) {
    if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;
    hipError_t es = launch_separable_f2(h, P, period, v_next, v_cur, pol, pd, pp);
    if (es != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "period %d separable kernel (lead-time family): %s", period, hipGetErrorString(es));
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
  if (h->d.kernel == SDPGPU_KERNEL_SEPARABLE && h->d.family == SDPGPU_FAMILY_CASH_LEADTIME) {
    // F5 (SingleProductLeadtime.java:72-119): the lambdas read the inventory and the pipeline quantity through x + preQ only, so
    // the rows of one level hold the same values and the same arg-max: one representative row per level through the row
    // kernel, every cell in the reference's operation order, then one copy per state.  EXACT; opt-in like the F2 mode.
    if (h->d.world_size != 1 || p.lo != 0 || p.hi != p.S) return fail(h, SDPGPU_ERR_UNSUPPORTED, "the separable mode of the cash + lead-time family runs on one rank");
    if (p.g.nq < 2 || !cash_row_eligible(h, period))
      return fail(h, SDPGPU_ERR_UNSUPPORTED, "separable mode (cash + lead-time family): the grid does not fit the cash row kernel");
    if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;
    hipError_t es = flush_pending(h);
    if (es == hipSuccess) es = launch_cash_row(h, P, period, v_next, v_cur, pol, pd, pp, p.lo, p.hi, h->stream, true);
    if (es == hipSuccess) es = launch_level_fill(h, period, v_cur, pol, h->stream);
    if (es != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "period %d separable kernel (cash + lead-time family): %s", period, hipGetErrorString(es));
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
  if (h->d.kernel == SDPGPU_KERNEL_SEPARABLE) {
    if (h->d.family != SDPGPU_FAMILY_BACKORDER) return fail(h, SDPGPU_ERR_UNSUPPORTED, "the separable mode exists for the backorder, lead-time and cash + lead-time families only");
    if (h->n_actions_full > 6000) return fail(h, SDPGPU_ERR_UNSUPPORTED, "separable mode: action range exceeds the LDS tile");
    if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;
    bool too_big = false;
    hipError_t es = launch_separable(h, P, period, v_next, v_cur, pol, pd, pp, &too_big);
    if (too_big) return fail(h, SDPGPU_ERR_UNSUPPORTED, "separable mode: action + demand range exceeds the LDS tile");
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
    use_window = !own_counts && window_eligible(h, period);
  }
  // automatic choice, nothing forced, and no window plan fits (a pmf of thousands of points: the window of even the smallest
  // block would take more than a compute unit's LDS): the generic kernel evaluates any period
  if (use_window && h->d.kernel == SDPGPU_KERNEL_AUTO && h->d.family == SDPGPU_FAMILY_BACKORDER && !ranged &&
      !(h->win_r || h->win_s || h->win_nch) && !plan_window(h, period, p.lo, p.hi).R)
    use_window = false;
  if (part != SDPGPU_PART_ALL) {
    // only the F1 window kernel has a bounded dependency footprint; everything else is "all boundary"
    const bool splittable = use_window && window_interior_tiles(h, period, p.lo, p.hi, nullptr, nullptr);
    if (!splittable) {
      if (part == SDPGPU_PART_INTERIOR) return SDPGPU_OK;  // nothing can start before the exchange
      part = SDPGPU_PART_ALL;
    }
  }
  hipError_t e;
  p.ops_cell = 0;
  p.lds_cell = p.l1_cell = 0;
  h->plan_error.clear();
  if (use_window) {
    e = launch_window(h, P, period, v_next, v_cur, pol, pd, h->d_pmf + p.pmf_win_off, ranged ? range_lo : p.lo,
                      ranged ? range_hi : p.hi, h->stream, part);
    p.kernel_used = SDPGPU_KERNEL_WINDOW;
  } else if (!own_counts && h->d.kernel != SDPGPU_KERNEL_GATHER && h->use_cash_shift && cash_shift_eligible(h, period)) {
    e = flush_pending(h);
    if (e == hipSuccess) e = launch_cash_shift(h, P, period, v_next, v_cur, pol, pd, pp, p.lo, p.hi, h->stream);
    p.kernel_used = SDPGPU_KERNEL_WINDOW;  // reported as a specialised (non-gather) kernel
  } else if (!own_counts && h->d.kernel != SDPGPU_KERNEL_GATHER && cash_row_eligible(h, period)) {
    e = flush_pending(h);
    if (e == hipSuccess) e = launch_cash_row(h, P, period, v_next, v_cur, pol, pd, pp, p.lo, p.hi, h->stream);
    p.kernel_used = SDPGPU_KERNEL_WINDOW;
  } else {
    e = flush_pending(h);  // the gather kernel reads the final V_{t+1} row
    if (e == hipSuccess)
      e = launch_gather_grid(P, v_next, v_cur, pol, pd, pp, p.lo, p.hi, h->stream);
    p.kernel_used = SDPGPU_KERNEL_GATHER;
  }
  if (e != hipSuccess && !h->plan_error.empty()) {  // the planner said no before anything was launched
    const bool forced = h->win_r || h->win_s || h->win_nch;
    if (use_window && h->d.kernel == SDPGPU_KERNEL_AUTO && !forced && part == SDPGPU_PART_ALL && !ranged) {
      // nobody asked for the window kernel and the period does not fit it (a pmf of thousands of points: its window would take
      // more than a compute unit's LDS): the generic kernel evaluates any period
      h->plan_error.clear();
      e = flush_pending(h);
      if (e == hipSuccess) e = launch_gather_grid(P, v_next, v_cur, pol, pd, pp, p.lo, p.hi, h->stream);
      p.kernel_used = SDPGPU_KERNEL_GATHER;
      p.ops_cell = 0;
      p.lds_cell = p.l1_cell = 0;
    } else {
      const std::string why = h->plan_error;
      return fail(h, SDPGPU_ERR_ARG, "period %d: %s", period, why.c_str());
    }
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
  if (h->custom) {
    int rc_c = custom_check(h);
    if (rc_c || !h->level_shape) return rc_c;  // (the level shape's window periods leave pending rows like the built-in family's)
  }
  if (h->n_pending == 0) return SDPGPU_OK;
  int rc = ensure_device(h);
  if (rc) return rc;
  hipError_t e = flush_pending(h);
  if (e != hipSuccess) return fail(h, SDPGPU_ERR_DEVICE, "combine: %s", hipGetErrorString(e));
  return SDPGPU_OK;
}

}  // namespace sdpgpu_detail

using namespace sdpgpu_detail;

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int sdpgpu_abi_version(void) { return SDPGPU_ABI_VERSION; }

#ifndef SDPGPU_BUILD_ID
#define SDPGPU_BUILD_ID "unknown"
#endif
// (the marker in front lets build.py read the identity out of the file without loading it)
static const char g_build_id[] = "sdpgpu-build-id:" SDPGPU_BUILD_ID;
const char* sdpgpu_build_id(void) { return g_build_id + sizeof("sdpgpu-build-id:") - 1; }

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
    h->lvl_p.resize((size_t)h->T);
    h->lvl_len.resize((size_t)h->T);
    h->lvl_rows.assign((size_t)h->T, 0);
    h->lvl_maxj.assign((size_t)h->T, 0);
    h->period_done.assign((size_t)h->T, 0);
    h->policy_done.assign((size_t)h->T, 0);
    h->pending_chunks.assign((size_t)h->T + 1, 0);
    h->counts.resize((size_t)h->T);
    h->d_counts.assign((size_t)h->T, nullptr);
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
  if (desc->family == SDPGPU_FAMILY_STAFF) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "a user functor takes one pmf per period; the staff family's level-dependent pmf is a built-in shape");
  if (desc->lead_time == 2) return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "a user functor has one pipeline quantity at most (lead_time 2 is a built-in shape)");
  // SDP_SHAPE_LEVEL in the text: the lambdas are sdp_action_cost + sdp_level_cost (see include/sdpgpu.h), evaluated from tables by
  // the F1 window kernel; the generic loop around them stays available (kernel = GATHER, off-grid queries, the reachable set)
  const bool level_shape = std::strstr(functor_source, "#define SDP_SHAPE_LEVEL") != nullptr;
  if (level_shape && desc->family != SDPGPU_FAMILY_BACKORDER)
    return fail(nullptr, SDPGPU_ERR_ARG, "SDP_SHAPE_LEVEL is the backorder family's state shape (one inventory level)");
  if (desc->kernel != SDPGPU_KERNEL_AUTO && desc->kernel != SDPGPU_KERNEL_GATHER && !(level_shape && desc->kernel == SDPGPU_KERNEL_WINDOW))
    return fail(nullptr, SDPGPU_ERR_UNSUPPORTED, "a user functor runs on the generic kernel only (or, with SDP_SHAPE_LEVEL, on the window kernel)");
  // compile: prelude + the user's three device functions + the engine kernels, strict fp64 (no FMA)
  std::string src = std::string(sdp::kCustomPrelude) + functor_source + "\n" + sdp::kCustomEngine;
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "sdp_custom.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return fail(nullptr, SDPGPU_ERR_DEVICE, "hiprtcCreateProgram failed");
  const std::string np_def = "-DSDP_NP=" + std::to_string(std::max(1, (int)n_params));
  // the state shape, loop variant and direction of this handle as compile-time constants (sdp_custom_src.hpp)
  const std::string shape_defs[5] = {
      std::string("-DSDP_HAS_CASH=") + (has_cash(desc->family) ? "1" : "0"),
      std::string("-DSDP_HAS_PREQ=") + (has_preq(desc->family) ? "1" : "0"),
      std::string("-DSDP_SURVIVAL=") + (desc->family == SDPGPU_FAMILY_SURVIVAL ? "1" : "0"),
      std::string("-DSDP_MAXDIR=") + (desc->direction == SDPGPU_MAX ? "1" : "0"),
      std::string("-DSDP_CASH_INT_DIV=") + (desc->cash_round_int_div ? "1" : "0")};
  // (the level shape's generic lambdas read the descriptor's action count and clamp as constants; hex floats are exact)
  char hexbuf[2][64];
  std::snprintf(hexbuf[0], sizeof hexbuf[0], "-DSDP_LEVEL_MIN=%a", desc->min_inventory);
  std::snprintf(hexbuf[1], sizeof hexbuf[1], "-DSDP_LEVEL_MAX=%a", desc->max_inventory);
  const std::string level_defs[2] = {
      "-DSDP_LEVEL_NACT=" + std::to_string(desc->step > 0 ? (int)(desc->max_order_quantity / desc->step) + 1 : 1),
      std::string("-DSDP_LEVEL_CLAMP=") + (desc->clamp_inventory ? "1" : "0")};
  std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", np_def.c_str(),
                                   shape_defs[0].c_str(), shape_defs[1].c_str(), shape_defs[2].c_str(), shape_defs[3].c_str(),
                                   shape_defs[4].c_str(), level_defs[0].c_str(), level_defs[1].c_str(), hexbuf[0], hexbuf[1]};
  // Specialise instead of interpret (round 4): on a CLAMPED grid every period has the same box, known from the descriptor alone
  // (layout(): inventory bounds, cash keys of the bounds, pipeline quantities 0 .. maxOrderQuantity) -- the grid, the step and
  // the user's constants go into the generated source as compile-time constants (exact: hex floats).  The successor's index
  // arithmetic folds (one axis: no 64-bit products) and the compiler specialises the user's formulas to their constants:
  // configs[1] through CLSP's three lambdas 0.94e12 -> 1.26e12 cells/s (1.53e12 with the NaN-free compile below).  Constant folding is IEEE round-to-nearest, no
  // contraction: the same doubles.  SDPGPU_CUSTOM_BAKE=0: everything read from the parameter block, as in rounds 1-3.
  std::vector<std::string> bake;
  bool finite = std::isfinite(desc->step) && std::isfinite(desc->min_inventory) && std::isfinite(desc->max_inventory) &&
                std::isfinite(desc->min_cash) && std::isfinite(desc->max_cash) && std::isfinite(desc->max_order_quantity) &&
                std::isfinite(desc->cash_round_mult) && std::isfinite(desc->cash_round_div) && desc->cash_round_div != 0;
  for (int i = 0; i < n_params; ++i) finite = finite && std::isfinite(params[i]);  // (a NaN / infinity has no literal: read at run time then)
  if (!(std::getenv("SDPGPU_CUSTOM_BAKE") && std::atoi(std::getenv("SDPGPU_CUSTOM_BAKE")) == 0) && desc->clamp_inventory && finite &&
      desc->step > 0 && desc->max_inventory >= desc->min_inventory) {
    char b[96];
    long long nc = 1, k_lo = 0, nq = 1;
    if (has_cash(desc->family)) {
      k_lo = cash_key_of_bound(*desc, desc->min_cash);
      nc = cash_key_of_bound(*desc, desc->max_cash) - k_lo + 1;
    }
    if (has_preq(desc->family)) nq = java_d2i(desc->max_order_quantity / desc->step) + 1;
    bake.push_back("-DSDP_BAKE=1");
    if (!(std::getenv("SDPGPU_CUSTOM_NNAN") && std::atoi(std::getenv("SDPGPU_CUSTOM_NNAN")) == 0)) bake.push_back("-fno-honor-nans");
    std::snprintf(b, sizeof b, "-DSDP_B_STEP=%a", desc->step); bake.push_back(b);
    std::snprintf(b, sizeof b, "-DSDP_B_INV_STEP=%a", 1.0 / desc->step); bake.push_back(b);
    std::snprintf(b, sizeof b, "-DSDP_B_XLO=%a", desc->min_inventory); bake.push_back(b);
    std::snprintf(b, sizeof b, "-DSDP_B_NX=%lld", (long long)((desc->max_inventory - desc->min_inventory) / desc->step) + 1); bake.push_back(b);
    std::snprintf(b, sizeof b, "-DSDP_B_NC=%lld", nc); bake.push_back(b);
    std::snprintf(b, sizeof b, "-DSDP_B_NQ=%lld", nq); bake.push_back(b);
    std::snprintf(b, sizeof b, "-DSDP_B_KLO=%lld", k_lo); bake.push_back(b);
    std::string pl = "-DSDP_B_PARAMS={";
    for (int i = 0; i < std::max(1, (int)n_params); ++i) {
      std::snprintf(b, sizeof b, "%s%a", i ? "," : "", i < n_params ? params[i] : 0.0);
      pl += b;
    }
    pl += "}";
    bake.push_back(pl);
    for (const std::string& o : bake) opts.push_back(o.c_str());
  }
  hiprtcResult cr = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
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
  if (const char* dump = std::getenv("SDPGPU_CUSTOM_DUMP")) {  // the code object, for llvm-objdump -d (tools/custom_functor_isa.py)
    if (FILE* f = std::fopen(dump, "wb")) {
      (void)std::fwrite(code.data(), 1, code.size(), f);
      (void)std::fclose(f);
    }
  }
  int rc = sdpgpu_create(desc, out);
  if (rc) return rc;
  sdpgpu_handle* h = *out;
  h->custom = true;
  h->level_shape = level_shape;
  h->custom_code.swap(code);
  h->custom_params.assign(params, params + n_params);
  return SDPGPU_OK;
}

void sdpgpu_destroy(sdpgpu_handle* h) {
  if (!h) return;
  if (h->allocated || h->d_policy || h->d_pmf || !h->staff_owned.empty()) {
    if (h->device >= 0) (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
  }
  graph_drop(h);
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
  if (h->d_diag) (void)hipFree(h->d_diag);
  if (h->d_rowtab) (void)hipFree(h->d_rowtab);
  if (h->d_rowperm) (void)hipFree(h->d_rowperm);
  if (h->d_units) (void)hipFree(h->d_units);
  for (void* q : h->staff_owned) (void)hipFree(q);
  if (h->d_staff_val) (void)hipFree(h->d_staff_val);
  if (h->d_staff_idx) (void)hipFree(h->d_staff_idx);
  for (int32_t* q : h->d_counts)
    if (q) (void)hipFree(q);
  if (h->d_sep_val) (void)hipFree(h->d_sep_val);
  if (h->d_sep_idx) (void)hipFree(h->d_sep_idx);
  if (h->d_custom_params) (void)hipFree(h->d_custom_params);
  if (h->d_custom_cells) (void)hipFree(h->d_custom_cells);
  if (h->d_custom_err) (void)hipFree(h->d_custom_err);
  if (h->d_level_tabs) (void)hipFree(h->d_level_tabs);
  if (h->custom_mod) (void)hipModuleUnload(h->custom_mod);
  comm_release(h);
  if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* sdpgpu_last_error(const sdpgpu_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int sdpgpu_set_pmf(sdpgpu_handle* h, int32_t t, const double* demand, const double* prob, int32_t n) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (t < 0 || t >= h->T || !demand || !prob || n < 1) return fail(h, SDPGPU_ERR_ARG, "set_pmf: bad argument (t=%d n=%d)", t, n);
  if (h->allocated) return fail(h, SDPGPU_ERR_STATE, "pmf is frozen once the device tables exist");
  if (h->d.family == SDPGPU_FAMILY_STAFF) return fail(h, SDPGPU_ERR_ARG, "the staff family takes its pmf per hire-up-to level: sdpgpu_set_level_pmf");
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

int sdpgpu_set_level_pmf(sdpgpu_handle* h, int32_t t, const double* prob, const int32_t* row_len, int32_t n_rows,
                         int32_t row_stride) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (h->d.family != SDPGPU_FAMILY_STAFF) return fail(h, SDPGPU_ERR_ARG, "set_level_pmf: only the staff family has a level-dependent pmf");
  if (t < 0 || t >= h->T || !prob || n_rows < 1 || row_stride < 1) return fail(h, SDPGPU_ERR_ARG, "set_level_pmf: bad argument (t=%d rows=%d stride=%d)", t, n_rows, row_stride);
  if (h->allocated) return fail(h, SDPGPU_ERR_STATE, "pmf is frozen once the device tables exist");
  try {
    std::vector<int32_t> len((size_t)n_rows);
    int32_t maxj = 1;
    for (int32_t y = 0; y < n_rows; ++y) {
      const int32_t n = row_len ? row_len[y] : y + 1;
      if (n < 1 || n > y + 1 || n > row_stride) return fail(h, SDPGPU_ERR_ARG, "set_level_pmf: row %d of period %d has %d entries (1 .. min(y + 1, row_stride) allowed)", y, t + 1, n);
      len[(size_t)y] = n;
      maxj = std::max(maxj, n);
    }
    if ((size_t)n_rows * (size_t)maxj > ((size_t)1 << 31)) return fail(h, SDPGPU_ERR_UNSUPPORTED, "set_level_pmf: table of %d x %d entries is too large", n_rows, maxj);
    std::vector<double> tp((size_t)n_rows * (size_t)maxj, 0.0);
    for (int32_t y = 0; y < n_rows; ++y)
      for (int32_t j = 0; j < len[(size_t)y]; ++j) tp[(size_t)j * (size_t)n_rows + (size_t)y] = prob[(size_t)y * (size_t)row_stride + (size_t)j];
    h->lvl_p[(size_t)t].swap(tp);
    h->lvl_len[(size_t)t].swap(len);
    h->lvl_rows[(size_t)t] = n_rows;
    h->lvl_maxj[(size_t)t] = maxj;
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
  h->per[t].cells_counted = false;
  graph_drop(h);  // (the overhead is a kernel argument of the captured launches)
  return SDPGPU_OK;
}

int sdpgpu_set_action_counts(sdpgpu_handle* h, int32_t t, const int32_t* counts, int64_t n) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (t < 0 || t >= h->T || !counts || n < 1) return fail(h, SDPGPU_ERR_ARG, "set_action_counts: bad argument");
  if (h->allocated) return fail(h, SDPGPU_ERR_STATE, "set_action_counts must precede the first run");
  if (h->custom) return fail(h, SDPGPU_ERR_UNSUPPORTED, "set_action_counts: a user functor has its own sdp_feasible_count");
  if (h->d.family == SDPGPU_FAMILY_STAFF) return fail(h, SDPGPU_ERR_UNSUPPORTED, "set_action_counts: not for the staff family");
  const int32_t cap = full_action_count(h->d);
  try {
    for (int64_t i = 0; i < n; ++i)
      if (counts[i] < 0 || counts[i] > cap)
        return fail(h, SDPGPU_ERR_ARG, "set_action_counts: state %lld has %d actions, the family's action list has at most %d", (long long)i, counts[i], cap);
    h->counts[(size_t)t].assign(counts, counts + n);  // (the length is checked against the grid once it is laid out)
    h->per[(size_t)t].cells_counted = false;
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "out of host memory");
  }
  return SDPGPU_OK;
}

int sdpgpu_set_stream(sdpgpu_handle* h, void* hip_stream) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  graph_drop(h);  // (captured on the old stream's behalf; the next sweep on the new stream is eager, the one after captures)
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
  // range (and NaN) checks come BEFORE the casts: converting a NaN or an out-of-range double to int64 is undefined
  double qx = (x - g.x_lo) / d.step;
  if (!(qx >= 0.0 && qx < (double)g.nx)) return -1;
  int64_t ix = (int64_t)qx;
  if ((double)ix != qx) return -1;
  int64_t ic = 0, iq = 0;
  if (has_cash(d.family)) {
    if (!(std::fabs(cash) < 4.0e15)) return -1;  // NaN, infinities, beyond exact integers: not a grid point
    const bool xr = d.cash_formula == 2;  // the state tuple is (x, R): `cash` is R = gridCash + variCost * x
    const double r_in = cash;
    if (xr) {  // the grid cash nearest to R - variCost * x; the exact R is required below
      const double approx = r_in - d.unit_order_cost * x;
      cash = d.cash_round_int_div ? (double)java_round(approx) : (double)java_round(approx * d.cash_round_mult) / d.cash_round_div;
    }
    int64_t k = d.cash_round_int_div ? (int64_t)cash : java_round(cash * d.cash_round_mult);
    double back = d.cash_round_int_div ? (double)k : (double)k / d.cash_round_div;
    if (back != cash) return -1;
    if (xr && back + d.unit_order_cost * x != r_in) return -1;  // not the R the transition would have formed
    ic = k - g.k_lo;
    if (ic < 0 || ic >= g.nc) return -1;
  }
  if (has_preq(d.family)) {
    double qq = preq / d.step;
    if (!(qq >= 0.0 && qq < (double)g.nq1)) return -1;
    iq = (int64_t)qq;
    if ((double)iq != qq) return -1;
  }
  if (d.lead_time == 2) {
    double qq = preq2 / d.step;
    if (!(qq >= 0.0 && qq < (double)(g.nq / g.nq1))) return -1;
    int64_t iq2 = (int64_t)qq;
    if ((double)iq2 != qq) return -1;
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

int sdpgpu_run_period_range(sdpgpu_handle* h, int32_t period, int64_t lo, int64_t hi) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (lo < 0 || hi < lo) return fail(h, SDPGPU_ERR_ARG, "run_period_range: bad range");
  try {
    int rc = run_period_impl(h, period, SDPGPU_PART_ALL, lo, hi);
    if (rc == SDPGPU_OK) count_cells(h, period);  // (cells of the rank's own slab: the widening is redundant work)
    return rc;
  } catch (const std::exception& e) {
    return fail(h, SDPGPU_ERR_ARG, "exception: %s", e.what());
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "unknown exception");
  }
}

int sdpgpu_set_halo(sdpgpu_handle* h, int64_t halo) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (halo < 0) return fail(h, SDPGPU_ERR_ARG, "set_halo: negative halo");
  if (h->d_chunk_val || h->allocated) return fail(h, SDPGPU_ERR_STATE, "set_halo must precede the first run");
  h->halo = halo;
  return SDPGPU_OK;
}

int sdpgpu_plan_period(const sdpgpu_handle* hc, int32_t period, sdpgpu_plan* out) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h || !out) return SDPGPU_ERR_ARG;
  h->err.clear();
  std::memset(out, 0, sizeof *out);
  if (period < 1 || period > h->T) return fail(h, SDPGPU_ERR_ARG, "plan: period %d", period);
  int rc = layout(h);
  if (rc) return rc;
  out->kernel = SDPGPU_KERNEL_GATHER;
  // (the launcher's own predicate: an AUTO handle with action counts of the caller's for this period runs the generic kernel)
  const bool own_counts = !h->counts[(size_t)period - 1].empty();
  const bool f1_window = f1_like(h) && h->d.kernel != SDPGPU_KERNEL_GATHER && h->d.family == SDPGPU_FAMILY_BACKORDER && window_eligible(h, period) &&
                         (h->d.kernel == SDPGPU_KERNEL_WINDOW || (h->d.kernel == SDPGPU_KERNEL_AUTO && !own_counts));
  if (!f1_window) return SDPGPU_OK;
  const PeriodInfo& p = h->per[period - 1];
  std::string why;
  const WinPlan pl = plan_window(h, period, p.lo, p.hi, &why);
  if (!pl.R) {
    // (automatic kernel choice, nothing forced: such a period runs on the generic kernel, see run_period_impl)
    if (h->d.kernel == SDPGPU_KERNEL_AUTO && !(h->win_r || h->win_s || h->win_nch)) return SDPGPU_OK;
    return fail(h, SDPGPU_ERR_ARG, "period %d: %s", period, why.c_str());
  }
  out->kernel = SDPGPU_KERNEL_WINDOW;
  out->r = pl.R;
  out->s = pl.S;
  out->chunks = pl.n_chunks;
  out->chunk_blocks = pl.chunk_blocks;
  out->tiles = pl.n_tiles;
  out->tasks = pl.n_tasks;
  out->workgroups_per_cu = lds_workgroups(pl.smem);
  out->lds_bytes = (int64_t)pl.smem;
  return SDPGPU_OK;
}

int sdpgpu_footprint(const sdpgpu_handle* hc, int32_t period, int64_t* left, int64_t* right) {
  sdpgpu_handle* h = const_cast<sdpgpu_handle*>(hc);
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (period < 1 || period > h->T) return fail(h, SDPGPU_ERR_ARG, "footprint: period %d", period);
  int rc = layout(h);
  if (rc) return rc;
  // (clamped grids only: with per-period boxes the slabs of consecutive periods are different index ranges and a
  // widened slab is not simply "the same slab plus a margin")
  if (!f1_like(h) || h->d.family != SDPGPU_FAMILY_BACKORDER || !h->d.clamp_inventory || h->d.kernel == SDPGPU_KERNEL_GATHER ||
      h->d.kernel == SDPGPU_KERNEL_SEPARABLE || !window_eligible(h, period))
    return fail(h, SDPGPU_ERR_UNSUPPORTED, "footprint: unbounded (only the backorder family on the window kernel reads a bounded neighbourhood)");
  int64_t l = 0, r = 0;
  if (period < h->T) {
    // state i of period t reads V_{t+1}[i + idx_off - (D - 1) ... i + idx_off + A - 1] (then clamped to the grid)
    const PeriodInfo& p = h->per[period - 1];
    const double lev0 = p.g.x_lo - h->pmf_d[period - 1][0];
    const int64_t idx_off = (int64_t)((lev0 - h->per[period].g.x_lo) / h->d.step);
    l = std::max<int64_t>(0, (int64_t)p.nD_win - 1 - idx_off);
    r = std::max<int64_t>(0, (int64_t)h->n_actions_full - 1 + idx_off);
  }
  if (left) *left = l;
  if (right) *right = r;
  return SDPGPU_OK;
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

// The sweep proper: T period launches + the deferred read-out, enqueued on the handle's stream.
static int enqueue_sweep(sdpgpu_handle* h) {
  std::fill(h->period_done.begin(), h->period_done.end(), 0);
  for (int period = h->T; period >= 1; --period) {
    int rc = run_period_impl(h, period);
    if (rc) return rc;
  }
  return flush_api(h);
}

int sdpgpu_solve(sdpgpu_handle* h, int32_t sync) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (h->d.world_size != 1) return fail(h, SDPGPU_ERR_STATE, "sdpgpu_solve needs world_size 1; sharded handles run period by period with an all-gather in between");
  struct InSolve {
    sdpgpu_handle* h;
    ~InSolve() { h->in_solve = false; }
  } guard{h};
  h->in_solve = true;
  try {
    int rc = allocate(h);
    if (rc) return rc;
    rc = ensure_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_solve0, h->stream));
    // With SDPGPU_GRAPH=1: sweep 1 runs eagerly (lazy allocations, LDS attributes, table uploads), sweep 2 is CAPTURED into a
    // HIP graph while it is enqueued, sweeps 3.. replay it: one hipGraphLaunch instead of T + 2 launches.  Not with per-period profiling events,
    // user functors (hipModule launches), the legacy NULL stream (it cannot be captured) or the tests' guard-word mode.
    const bool graphable = h->graph_state >= 1 && !h->profiling && !h->custom && h->stream != nullptr &&
                           !std::getenv("SDPGPU_CASH_DIAG_CHECK");
    bool done = false;
    if (graphable && h->graph_state == 2 && h->sweep_exec) {
      hipError_t e = hipGraphLaunch(h->sweep_exec, h->stream);
      if (e == hipSuccess) {
        h->graph_replays++;
        done = true;
      } else {
        (void)hipGetLastError();
        graph_drop(h);
        h->graph_state = -1;  // (eager from here on)
      }
    } else if (graphable && h->graph_state == 1) {
      hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        h->flush_uploads = 0;
        rc = enqueue_sweep(h);
        hipGraph_t g = nullptr;
        e = hipStreamEndCapture(h->stream, &g);
        // (the job list of the deferred read-out is uploaded from ONE host buffer: a sweep that flushes twice -- window
        // periods between generic ones -- would replay both uploads with the second list.  Such sweeps stay eager.)
        if (h->flush_uploads > 1) rc = SDPGPU_ERR_UNSUPPORTED;
        if (rc == SDPGPU_OK && e == hipSuccess && g) e = hipGraphInstantiate(&h->sweep_exec, g, nullptr, nullptr, 0);
        if (rc == SDPGPU_OK && e == hipSuccess && g && h->sweep_exec) {
          h->sweep_graph = g;
          e = hipGraphLaunch(h->sweep_exec, h->stream);  // (capturing enqueued nothing: this is sweep 2 itself)
          if (e == hipSuccess) {
            h->graph_state = 2;
            h->graph_replays++;
            done = true;
          } else {
            // the first launch of the new graph failed: nothing of this sweep ran on the device.  Drop the graph and its
            // executable (graph_drop destroys both: a later capture must not overwrite a live exec), rewind the host
            // bookkeeping the un-run sweep advanced, stay eager from here on, and run this sweep eagerly below.
            (void)hipGetLastError();
            graph_drop(h);
            h->graph_state = -1;
            std::fill(h->pending_chunks.begin(), h->pending_chunks.end(), 0);
            h->n_pending = 0;
            std::fill(h->key_row_clean.begin(), h->key_row_clean.end(), 0);
            std::fill(h->period_done.begin(), h->period_done.end(), 0);
            h->err.clear();
          }
        } else {
          // the capture was refused somewhere (a launcher had to allocate or wait): nothing ran.  Forget the graph and the host
          // bookkeeping of the un-run sweep, and run this sweep eagerly.
          (void)hipGetLastError();
          if (g) (void)hipGraphDestroy(g);
          if (h->sweep_exec) (void)hipGraphExecDestroy(h->sweep_exec);
          h->sweep_exec = nullptr;
          h->graph_state = -1;
          std::fill(h->pending_chunks.begin(), h->pending_chunks.end(), 0);
          h->n_pending = 0;
          std::fill(h->key_row_clean.begin(), h->key_row_clean.end(), 0);
          h->err.clear();
        }
      } else {
        (void)hipGetLastError();
        h->graph_state = -1;
      }
    }
    if (!done) {
      rc = enqueue_sweep(h);
      if (rc) return rc;
      if (h->graph_state == 0) {
        // OPT-IN (SDPGPU_GRAPH=1).  Measured on one box, back to back (round 3): the target grid 53.63 ms per sweep replayed
        // against 53.29 ms eager; configs[1] (52 launches of 30 us) 1.525 against 1.501 ms -- the runtime's graph launch
        // costs more between kernel nodes than the eager launches it replaces, and the eager sweep's wall time is already
        // within 0.4 % of its device time.  Kept for hosts whose issuing thread is not that prompt.
        const char* env = std::getenv("SDPGPU_GRAPH");
        h->graph_state = (env && std::atoi(env) == 1) ? 1 : -1;
      }
    }
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
  if (h->d.family == SDPGPU_FAMILY_STAFF) return fail(h, SDPGPU_ERR_UNSUPPORTED, "eval_states: every staff number the recursion can visit lies on the grid; read the tables");
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
    const double* qc = cash ? d_in + nn : nullptr;
    const double* qp = preq ? d_in + 2 * nn : nullptr;
    const double* qp2 = preq2 ? d_in + 3 * nn : nullptr;
    if (h->custom)
      e = launch_custom_period(h, period, v_next, d_val, d_act, 0, n, d_in, qc, qp, false);
    else
      e = launch_gather_query(P, v_next, d_val, d_act, pd, pd + p.nD, n, d_in, qc, qp, qp2, h->stream);
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
  if (h->d.family == SDPGPU_FAMILY_STAFF) {  // an interval per period: host arithmetic (sdpgpu_staff.hip)
    int rl = layout(h);
    if (rl) return rl;
    const PeriodInfo& ps = h->per[period - 1];
    if (n < 0 || n > ps.S) return fail(h, SDPGPU_ERR_ARG, "reachable: n=%lld > %lld states", (long long)n, (long long)ps.S);
    std::vector<int64_t> rlo, rhi;
    staff_reach_intervals(h, &rlo, &rhi);
    const int64_t x0 = (int64_t)ps.g.x_lo;
    for (int64_t i = 0; i < n; ++i) out[i] = (x0 + i >= rlo[(size_t)period - 1] && x0 + i <= rhi[(size_t)period - 1]) ? 1 : 0;
    return SDPGPU_OK;
  }
  int rc = compute_reachable(h);
  if (rc) return rc;
  const PeriodInfo& p = h->per[period - 1];
  if (n < 0 || n > p.S) return fail(h, SDPGPU_ERR_ARG, "reachable: n=%lld > %lld states", (long long)n, (long long)p.S);
  HIP_TRY(h, hipMemcpy(out, h->d_reach + h->reach_off[period - 1], (size_t)n, hipMemcpyDeviceToHost));
  return SDPGPU_OK;
}

int sdpgpu_simulate(sdpgpu_handle* h, int64_t n_paths, const double* demand, const double* discount, double ini_x,
                    double ini_cash, double ini_preq, double* out_sum, uint8_t* out_valid) {
  if (h && h->d.cash_formula == 2) {
    h->err.clear();
    return fail(h, SDPGPU_ERR_UNSUPPORTED, "simulate: not built for the (x, R) state of CashConstraintXR (CashSimulationXR is out of scope)");
  }
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (n_paths < 0 || !demand || !discount || !out_sum || !out_valid) return fail(h, SDPGPU_ERR_ARG, "simulate: bad argument");
  if (h->d.world_size != 1) return fail(h, SDPGPU_ERR_STATE, "simulate needs the whole policy on one GPU (world_size 1)");
  if (h->custom) return fail(h, SDPGPU_ERR_UNSUPPORTED, "simulate: a user functor's lambdas live on the host; roll the policy tables forward there");
  if (h->d.family == SDPGPU_FAMILY_STAFF) return fail(h, SDPGPU_ERR_UNSUPPORTED, "simulate: the workforce drivers simulate an (s, S) rule with binomial draws (SimulatesS.java), not the table policy along demand paths");
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
    if (e == hipSuccess)
      e = launch_simulate(h, d_per, d_dem, d_disc, n_paths, idx0, ini_x, ini_cash, ini_preq, ini_preq2, (int)first_k, d_sum,
                          d_valid);
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
  bool modelled = true;
  for (int t = 0; t < h->T; ++t) {
    const PeriodInfo& p = h->per[t];
    out->states_total += p.S;
    out->cells_evaluated += p.cells_rank;
    out->cells_all_ranks += p.cells_all;
    if (h->period_done[t]) out->periods_run++;
    out->fp64_ops_executed += (double)p.cells_rank * p.ops_cell;
    out->lds_bytes += (double)p.cells_rank * p.lds_cell;
    out->l1_bytes += (double)p.cells_rank * p.l1_cell;
    if (p.ops_cell == 0 && p.cells_rank > 0) modelled = false;
  }
  if (!modelled) out->fp64_ops_executed = 0;
  out->graph_replays = h->graph_replays;
  out->kernel_used = h->per[0].kernel_used;
  if (f1_like(h) && h->d.family == SDPGPU_FAMILY_BACKORDER && h->per[0].kernel_used == SDPGPU_KERNEL_WINDOW &&
      window_eligible(h, 1)) {
    const WinPlan pl = plan_window(h, 1, h->per[0].lo, h->per[0].hi);
    out->window_r = pl.R;
    out->window_s = pl.S;
  }
  if (h->allocated) {
    (void)ensure_device(h);
    if (h->custom && !h->level_shape && h->d_custom_cells && hipStreamSynchronize(h->stream) == hipSuccess) {
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

int64_t sdpgpu_period_cells(sdpgpu_handle* h, int32_t period) {
  if (!h || period < 1 || period > h->T) return -1;
  const PeriodInfo& p = h->per[period - 1];
  if (h->custom && !h->level_shape) {  // counted on the device, one counter per period
    if (!h->d_custom_cells || !h->period_done[period - 1]) return -1;
    (void)ensure_device(h);
    unsigned long long c = 0;
    if (hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemcpy(&c, h->d_custom_cells + (period - 1), sizeof c, hipMemcpyDeviceToHost) != hipSuccess)
      return -1;
    return (int64_t)c;
  }
  return p.cells_counted ? p.cells_rank : -1;
}

}  // extern "C"

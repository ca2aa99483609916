// sdpgpu_generic.hip -- host side of the generic per-cell kernel (every family), the reachable set, the policy
// rollout and the user-defined functor path.
#include "sdpgpu_internal.hpp"
#include "sdp_gather.hpp"
#include "sdp_custom_src.hpp"

namespace sdpgpu_detail {

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
  // enough workgroups to fill 256 CUs several times over: fewer states (more action slots) per workgroup on small grids
  const int64_t n = hi - lo;
  const int variant = n >= 64 * 2048 ? 0 : (n >= 16 * 1024 ? 1 : 2);
  const int sx = variant == 0 ? 64 : (variant == 1 ? 16 : 4);
  const int64_t blocks = (n + sx - 1) / sx;
  if (!grid_ok(blocks)) return hipErrorInvalidValue;
  const size_t smem = (size_t)p.nD * 16 + (size_t)4 * sx * (sizeof(double) + sizeof(int));
  return hipModuleLaunchKernel(h->custom_period[variant], (unsigned)blocks, 1, 1, 256, 1, 1, (unsigned)smem, h->stream, args,
                               nullptr);
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

// User lambdas of the LEVEL SHAPE: per period the table M(m) = sdp_level_cost(lev0 + m step) over every level index the F1
// window kernel may stage (cells: m = state + action - demand step in [-(D - 1), nx + A - 2]; padded steps, padded actions and
// tail lanes reach further and are clamped onto the table by the kernel) and c(a) = sdp_action_cost(a step), a < A -- filled by
// the user's own compiled functions, once (the lambdas' constants are fixed at create; the period is an argument).
int fill_level_tables(sdpgpu_handle* h) {
  const int A = h->n_actions_full;
  h->level_tab_off.assign((size_t)h->T, 0);
  h->level_m_min.assign((size_t)h->T, 0);
  h->level_m_n.assign((size_t)h->T, 0);
  size_t total = 0;
  for (int t = 0; t < h->T; ++t) {
    const PeriodInfo& p = h->per[(size_t)t];
    const int D = std::max<int>(p.nD_win, p.nD);
    h->level_m_min[(size_t)t] = -(D + 64);
    h->level_m_n[(size_t)t] = (int32_t)std::min<int64_t>(p.g.nx + A + 2 * (int64_t)D + 1024, 2000000000LL);
    h->level_tab_off[(size_t)t] = total;
    total += (size_t)h->level_m_n[(size_t)t] + (size_t)A;
  }
  HIP_TRY(h, hipMalloc((void**)&h->d_level_tabs, total * sizeof(double)));
  for (int t = 0; t < h->T; ++t) {
    const PeriodInfo& p = h->per[(size_t)t];
    sdp::CustomParams C = make_custom_params(h, t + 1);
    double lev0 = p.g.x_lo - h->pmf_d[(size_t)t][0];  // level of m = 0: launch_window's W.lev0
    int m_min = h->level_m_min[(size_t)t], n_m = h->level_m_n[(size_t)t], n_a = A;
    double* m_tab = h->d_level_tabs + h->level_tab_off[(size_t)t];
    double* c_tab = m_tab + n_m;
    void* args[] = {&C, &lev0, &m_min, &n_m, &m_tab, &n_a, &c_tab};
    const int64_t blocks = ((int64_t)std::max(n_m, n_a) + 255) / 256;
    if (!grid_ok(blocks)) return fail(h, SDPGPU_ERR_UNSUPPORTED, "level tables: the grid of period %d is too long", t + 1);
    HIP_TRY(h, hipModuleLaunchKernel(h->custom_tabulate, (unsigned)blocks, 1, 1, 256, 1, 1, 0, h->stream, args, nullptr));
  }
  return SDPGPU_OK;
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

// ---- launch helpers --------------------------------------------------------------------------


template <int FAM, bool MAXDIR, int SX, bool QUERY>
hipError_t launch_gather_sx(const DevParams& P, const double* v_next, double* v_cur, int32_t* pol, const double* pmf_d,
                            const double* pmf_p, int64_t lo, int64_t hi, sdp::QueryStates q, hipStream_t st) {
  int64_t n = hi - lo;
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + SX - 1) / SX;
  if (!grid_ok(blocks)) return hipErrorInvalidValue;
  size_t smem = (size_t)P.n_demand * 16 + 4 * 64 * (sizeof(double) + sizeof(int));
  static LdsMark mark;  // (a pmf of 3905 .. 4000 points is 65-67 KiB: above the legacy limit of a launch)
  hipError_t ea = lds_allow(sdp::gather_period_kernel<FAM, MAXDIR, SX, QUERY>, smem, &mark);
  if (ea != hipSuccess) return ea;
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

hipError_t launch_gather_grid(const DevParams& P, const double* v_next, double* v_cur, int32_t* pol, const double* pmf_d,
                              const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st) {
  return launch_gather<false>(P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, sdp::QueryStates{nullptr, nullptr, nullptr, nullptr}, st);
}

// sdpgpu_eval_states: n explicit state tuples instead of grid indices
hipError_t launch_gather_query(const DevParams& P, const double* v_next, double* out_val, int32_t* out_act,
                               const double* pmf_d, const double* pmf_p, int64_t n, const double* x, const double* cash,
                               const double* preq, const double* preq2, hipStream_t st) {
  return launch_gather<true>(P, v_next, out_val, out_act, pmf_d, pmf_p, 0, n, sdp::QueryStates{x, cash, preq, preq2}, st);
}

// sdpgpu_simulate: one path per lane over the policy tables (Simulation.java:59-69; RiskSimulation.java:213-234)
hipError_t launch_simulate(sdpgpu_handle* h, const sdp::SimPeriod* d_per, const double* d_dem, const double* d_disc,
                           int64_t n_paths, int64_t idx0, double ini_x, double ini_cash, double ini_preq,
                           double ini_preq2, int first_k, double* d_sum, uint8_t* d_valid) {
  sdp::StateT ini{ini_x, ini_cash, ini_preq, ini_preq2};
  dim3 grid((unsigned)((n_paths + 255) / 256));
  const int T = h->T;
#define SDP_SIM(F)                                                                                                   \
  case F:                                                                                                            \
    hipLaunchKernelGGL((sdp::simulate_kernel<F>), grid, dim3(256), 0, h->stream, d_per, T, h->d_policy, d_dem, d_disc, \
                       n_paths, idx0, ini, first_k, d_sum, d_valid);                                                  \
    break;
  switch (h->d.family) {
    SDP_SIM(sdp::FAM_BACKORDER)
    SDP_SIM(sdp::FAM_LEADTIME)
    SDP_SIM(sdp::FAM_CASH)
    SDP_SIM(sdp::FAM_OVERDRAFT)
    SDP_SIM(sdp::FAM_CASH_LEADTIME)
    SDP_SIM(sdp::FAM_SURVIVAL)
    default:
      return hipErrorInvalidValue;
  }
#undef SDP_SIM
  return hipGetLastError();
}

}  // namespace sdpgpu_detail

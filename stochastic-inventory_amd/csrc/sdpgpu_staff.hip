// sdpgpu_staff.hip -- host side of the STAFF family (workforce.StaffRecursion): level-pmf tables, the period launch,
// cell counts and the reachable intervals.  Kernels in sdp_staff.hpp.
#include "sdpgpu_internal.hpp"
#include "sdp_staff.hpp"

namespace sdpgpu_detail {

// Transposed device copy of one period's table: pT[j * rows + y], zero beyond a row's length and in kStaffPadJ rows
// before j = 0 and after j = maxj - 1 (the register-blocked kernel steps off both ends); d_lvl_p[t] is the j = 0 row.
int staff_upload(sdpgpu_handle* h) {
  h->d_lvl_p.assign((size_t)h->T, nullptr);
  h->d_lvl_len.assign((size_t)h->T, nullptr);
  for (int t = 0; t < h->T; ++t) {
    // periods that were given the same table share one device copy (the drivers use one turnover rate throughout)
    int same = -1;
    for (int u = 0; u < t && same < 0; ++u)
      if (h->lvl_rows[u] == h->lvl_rows[t] && h->lvl_len[u] == h->lvl_len[t] && h->lvl_p[u] == h->lvl_p[t]) same = u;
    if (same >= 0) {
      h->d_lvl_p[t] = h->d_lvl_p[same];
      h->d_lvl_len[t] = h->d_lvl_len[same];
      continue;
    }
    const size_t rows = (size_t)h->lvl_rows[t], maxj = (size_t)h->lvl_maxj[t];
    const size_t pad = (size_t)sdp::kStaffPadJ * rows;
    double* base = nullptr;
    HIP_TRY(h, hipMalloc((void**)&base, (rows * maxj + 2 * pad) * sizeof(double)));
    h->staff_owned.push_back(base);
    HIP_TRY(h, hipMalloc((void**)&h->d_lvl_len[t], rows * sizeof(int32_t)));
    h->staff_owned.push_back(h->d_lvl_len[t]);
    HIP_TRY(h, hipMemset(base, 0, (rows * maxj + 2 * pad) * sizeof(double)));
    h->d_lvl_p[t] = base + pad;
    HIP_TRY(h, hipMemcpy(h->d_lvl_p[t], h->lvl_p[t].data(), rows * maxj * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_lvl_len[t], h->lvl_len[t].data(), rows * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  return SDPGPU_OK;
}

// action groups per state tile: enough waves to fill 1024 SIMDs a few times over, few enough to keep the partial
// rows small
static int staff_block() {  // register block of staff_block_kernel (SDPGPU_STAFF_R = 4 | 8)
  static const int r = [] {
    const char* e = std::getenv("SDPGPU_STAFF_R");
    return e && std::atoi(e) == 8 ? 8 : 4;
  }();
  return r;
}

static void staff_groups(const sdpgpu_handle* h, int64_t states, int tile_states, int* n_groups, int* group_actions) {
  const int64_t tiles = std::max<int64_t>(1, (states + tile_states - 1) / tile_states);
  const int nA = h->n_actions_full;
  int64_t want = std::max<int64_t>(1, 16384 / tiles);
  want = std::min<int64_t>(want, nA);
  const int R = staff_block();  // groups hold whole register blocks
  const int ga = (int)(((nA + want - 1) / want + R - 1) / R * R);
  *group_actions = ga;
  *n_groups = (nA + ga - 1) / ga;
}

hipError_t launch_staff(sdpgpu_handle* h, int period, const double* v_next, double* v_cur, int32_t* pol, int64_t lo,
                        int64_t hi, hipStream_t st) {
  if (hi <= lo) return hipSuccess;
  const PeriodInfo& p = h->per[period - 1];
  const sdpgpu_desc& d = h->d;
  sdp::StaffParams S{};
  S.K = d.fixed_order_cost;
  S.v = d.unit_order_cost;
  S.salary = d.holding_cost;
  S.pen = d.penalty_cost;
  S.min_staff = (int32_t)p.overhead;
  S.n_actions = h->n_actions_full;
  S.n_rows = h->lvl_rows[period - 1];
  S.uni_rows = !(std::getenv("SDPGPU_STAFF_UNI") && std::atoi(std::getenv("SDPGPU_STAFF_UNI")) == 0);
  S.clamp = d.clamp_inventory;
  S.min_x = (int32_t)d.min_inventory;
  S.max_x = (int32_t)d.max_inventory;
  S.x_lo = (int32_t)p.g.x_lo;
  S.next_x_lo = period < h->T ? (int32_t)h->per[period].g.x_lo : 0;
  if (d.clamp_inventory) {
    S.nn_lo = S.min_x;
    S.nn_hi = S.max_x;
  } else {  // (never binding for a cell inside its row; keeps the reads of zero-probability steps inside the table)
    S.nn_lo = S.next_x_lo;
    S.nn_hi = period < h->T ? S.next_x_lo + (int32_t)h->per[period].g.nx - 1 : 0;
  }
  // two adjacent states per lane (staff_pair_kernel) wherever the table and the next period's box have two entries
  static const bool pair_off = std::getenv("SDPGPU_STAFF_PAIR") && std::atoi(std::getenv("SDPGPU_STAFF_PAIR")) == 0;
  static const bool plain = std::getenv("SDPGPU_STAFF_BLOCK") && std::atoi(std::getenv("SDPGPU_STAFF_BLOCK")) == 0;
  // (only where 128-state tiles x action blocks still make a few waves per SIMD: small staff ranges keep 64-state tiles)
  const bool roomy = ((hi - lo + 127) / 128) * (int64_t)((h->n_actions_full + 3) / 4) >= 4096;
  // (the pair kernel addresses the table and V_{t+1} by 32-bit byte offsets from scalar bases)
  const bool small_tables = ((int64_t)S.n_rows + 2 * sdp::kStaffPadJ) * S.n_rows * 8 < 2147483647LL &&
                            (period == h->T || (int64_t)h->per[period].g.nx * 8 < 2147483647LL);
  const bool pair_ok = !pair_off && !plain && staff_block() == 4 && S.n_rows >= 2 && (period == h->T || S.nn_hi > S.nn_lo) && small_tables;
  // window form (staff_window_kernel: S adjacent states per lane, one probability per LEVEL instead of one per cell).  Which
  // form serves a period is its staff range's size -- measured per period on WorkforceTesting.main's instance (1, 1001, 2001,
  // ... 7001 states from period 1 to 8; tools/staff_variants.py): four states per lane from ~1500 numbers up (2001 states:
  // 0.60 ms against 0.65 with two per lane and 1.10 on the block kernel), two per lane from a few hundred (1001 states: 0.39
  // against 0.42 and 0.59), the block kernel below (one state: 0.22 against 0.30-0.37).  Until late in round 4 the window form
  // waited for 4096 tile-blocks, and the two smallest multi-state periods of that instance ran on the block kernel: 1.7 of the
  // sweep's 7.0 ms.  SDPGPU_STAFF_WIN=0 turns the form off, =2 / =4 forces the states per lane.
  int win_s = 0;
  if (pair_ok && hi - lo >= 1536) win_s = 4;
  else if (pair_ok && hi - lo >= 256) win_s = 2;
  if (const char* e = std::getenv("SDPGPU_STAFF_WIN")) {
    const int v = std::atoi(e);
    win_s = (v == 2 || v == 4) && pair_ok ? v : 0;
  }
  // (without the window form: two adjacent states per lane only where 128-state tiles x action blocks still fill the chip)
  const bool pair = win_s != 0 || (pair_ok && (roomy || std::getenv("SDPGPU_STAFF_PAIR")));
  // a handful of states: lanes = actions (staff_action_kernel); SDPGPU_STAFF_LANES=0 never, =1 for any number of states
  bool by_action = hi - lo <= 16 && !plain;
  if (const char* e = std::getenv("SDPGPU_STAFF_LANES")) by_action = std::atoi(e) != 0;
  const int tile_states = by_action ? 1 : (win_s ? 64 * win_s : (pair ? 128 : 64));
  if (by_action) {
    S.group_actions = 64;
    S.n_groups = (h->n_actions_full + 63) / 64;
  } else {
    staff_groups(h, hi - lo, tile_states, &S.n_groups, &S.group_actions);
  }
  const int64_t tiles = (hi - lo + tile_states - 1) / tile_states;
  const int64_t blocks = tiles * S.n_groups;
  if (blocks * 64 >= 4294967296LL) return hipErrorInvalidValue;
  double* out_val = v_cur;
  int32_t* out_idx = pol;
  const int part_rows = S.n_groups;
  if (part_rows > 1) {
    const size_t need = (size_t)part_rows * (size_t)(hi - lo);
    if (need > h->staff_part_elems) {
      if (h->d_staff_val) (void)hipFree(h->d_staff_val);
      if (h->d_staff_idx) (void)hipFree(h->d_staff_idx);
      h->d_staff_val = nullptr;
      h->d_staff_idx = nullptr;
      h->staff_part_elems = 0;
      hipError_t e = hipStreamSynchronize(st);  // an earlier period may still be reading the old rows
      if (e == hipSuccess) e = hipMalloc((void**)&h->d_staff_val, need * sizeof(double));
      if (e == hipSuccess) e = hipMalloc((void**)&h->d_staff_idx, need * sizeof(int32_t));
      if (e != hipSuccess) return e;
      h->staff_part_elems = need;
    }
    S.part_stride = hi - lo;
    out_val = h->d_staff_val - lo;  // the kernel indexes rows by flat state index
    out_idx = h->d_staff_idx - lo;
  }
  const double* pT = h->d_lvl_p[period - 1];
  const int32_t* len = h->d_lvl_len[period - 1];
  if (by_action) {
    if (period < h->T)
      hipLaunchKernelGGL((sdp::staff_action_kernel<true>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next, out_val,
                         out_idx, lo, hi);
    else
      hipLaunchKernelGGL((sdp::staff_action_kernel<false>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next, out_val,
                         out_idx, lo, hi);
  } else if (win_s) {
#define SDP_SW(SS, FU) \
  hipLaunchKernelGGL((sdp::staff_window_kernel<4, SS, FU>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next, out_val, out_idx, lo, hi)
    if (win_s == 4) { if (period < h->T) SDP_SW(4, true); else SDP_SW(4, false); }
    else { if (period < h->T) SDP_SW(2, true); else SDP_SW(2, false); }
#undef SDP_SW
  } else if (pair) {
    if (period < h->T)
      hipLaunchKernelGGL((sdp::staff_pair_kernel<4, true>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next,
                         out_val, out_idx, lo, hi);
    else
      hipLaunchKernelGGL((sdp::staff_pair_kernel<4, false>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next,
                         out_val, out_idx, lo, hi);
  } else if (plain) {  // one action at a time (kept as the cross-check of the register-blocked kernel)
    if (period < h->T)
      hipLaunchKernelGGL((sdp::staff_period_kernel<true>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next,
                         out_val, out_idx, lo, hi);
    else
      hipLaunchKernelGGL((sdp::staff_period_kernel<false>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next,
                         out_val, out_idx, lo, hi);
  } else if (staff_block() == 8) {
    if (period < h->T)
      hipLaunchKernelGGL((sdp::staff_block_kernel<8, true>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next,
                         out_val, out_idx, lo, hi);
    else
      hipLaunchKernelGGL((sdp::staff_block_kernel<8, false>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next,
                         out_val, out_idx, lo, hi);
  } else if (period < h->T) {
    hipLaunchKernelGGL((sdp::staff_block_kernel<4, true>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next,
                       out_val, out_idx, lo, hi);
  } else {
    hipLaunchKernelGGL((sdp::staff_block_kernel<4, false>), dim3((unsigned)blocks), dim3(64), 0, st, S, pT, len, v_next,
                       out_val, out_idx, lo, hi);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || part_rows == 1) return e;
  hipLaunchKernelGGL(sdp::combine_staff_kernel, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, st, out_val,
                     out_idx, part_rows, S.part_stride, v_cur, pol, lo, hi);
  return hipGetLastError();
}

// sum over states [lo, hi) and actions of pmfs[t][min(x + a, rows - 1)].length
int64_t staff_cells(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi) {
  const PeriodInfo& p = h->per[period - 1];
  const std::vector<int32_t>& len = h->lvl_len[period - 1];
  const int64_t rows = h->lvl_rows[period - 1], nA = h->n_actions_full;
  std::vector<int64_t> pre((size_t)rows + 1, 0);  // pre[y] = len[0] + .. + len[y-1]
  for (int64_t y = 0; y < rows; ++y) pre[(size_t)y + 1] = pre[(size_t)y] + len[(size_t)y];
  auto upto = [&](int64_t y) {  // sum of len(min(k, rows-1)) for k < y
    return y <= rows ? pre[(size_t)y] : pre[(size_t)rows] + (y - rows) * (int64_t)len[(size_t)rows - 1];
  };
  int64_t cells = 0;
  const int64_t x0 = (int64_t)p.g.x_lo;
  for (int64_t i = lo; i < hi; ++i) cells += upto(x0 + i + nA) - upto(x0 + i);
  return cells;
}

// The set StaffRecursion.getExpectedValue visits from (1, iniStaffNum) is an interval per period: from the levels
// y in [L, H + maxHire] the successors are the union of [y - len(y) + 1, y], which contains every y itself and
// hangs contiguous pieces below them; the clamp maps an interval to an interval.
void staff_reach_intervals(const sdpgpu_handle* h, std::vector<int64_t>* lo_out, std::vector<int64_t>* hi_out) {
  const sdpgpu_desc& d = h->d;
  int64_t L = (int64_t)d.ini_inventory, H = L;
  lo_out->assign((size_t)h->T, 0);
  hi_out->assign((size_t)h->T, -1);
  for (int t = 0; t < h->T; ++t) {
    (*lo_out)[(size_t)t] = L;
    (*hi_out)[(size_t)t] = H;
    const std::vector<int32_t>& len = h->lvl_len[(size_t)t];
    const int64_t rows = h->lvl_rows[(size_t)t];
    int64_t low = INT64_MAX;
    const int64_t top = H + (h->n_actions_full - 1);
    for (int64_t y = L; y <= std::min(top, rows - 1); ++y) low = std::min(low, y - len[(size_t)y] + 1);
    if (top > rows - 1) low = std::min(low, std::max(L, rows) - len[(size_t)rows - 1] + 1);
    L = low;
    H = top;
    if (d.clamp_inventory) {
      auto cl = [&](int64_t v) {
        v = v > (int64_t)d.max_inventory ? (int64_t)d.max_inventory : v;
        return v < (int64_t)d.min_inventory ? (int64_t)d.min_inventory : v;
      };
      L = cl(L);
      H = cl(H);
    }
  }
}

}  // namespace sdpgpu_detail

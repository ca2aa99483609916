// sdpgpu_cash.hip -- host side of the two cash-family kernels (sdp_cash.hpp).
#include "sdpgpu_internal.hpp"
#include "sdp_cash.hpp"

namespace sdpgpu_detail {

// ---- uniform-shift kernel (F3 on dyadic grids) ---------------------------------------------------
// LDS of a workgroup of cash_shift_kernel with tiles of tile_pts cash points
static size_t cash_shift_lds(int nD, int tile_pts) {
  const size_t dp8 = ((size_t)nD + 7) & ~(size_t)7;
  return (size_t)nD * 16 + dp8 * 16 * 8 + (size_t)4 * tile_pts * (sizeof(double) + sizeof(int)) + 4 * (dp8 / 2) * sizeof(int);
}
static inline int narrow_pieces(int cap, int S) { return 2 * S + (cap == 64 ? 1 : 2); }  // 64-entry pieces of a staged segment
bool dyadic(double x, double scale, double max_abs) { return std::fabs(x) <= max_abs && x * scale == std::floor(x * scale); }

// All arithmetic of the F3 lambdas is exact (see sdp_cash.hpp) iff the rates and the penalty are zero, the
// cash quantum is a power of two and every parameter is a multiple of 2^-10 of bounded size.
bool cash_shift_eligible(const sdpgpu_handle* h, int period) {
  const sdpgpu_desc& d = h->d;
  if (h->custom) return false;
  if (d.family != SDPGPU_FAMILY_CASH || !d.clamp_inventory) return false;
  if (d.cash_formula == 2) return false;  // (x, R) state: its own feasible-action rule (cash row / generic kernel)
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
  // (byte offsets into V_{t+1} are formed and clamped in signed 32-bit arithmetic)
  if ((p.S + 2 * p.g.nc) * 8 >= 2147483647LL || p.nD > 2000) return false;
  // cash_shift_kernel (period T, short rows, shifts too far apart for the diagonal form) keeps 144 B per demand point in LDS:
  // with its largest tile two workgroups must still fit a compute unit, else the cash row / generic kernels take the period
  if (cash_shift_lds(p.nD, 512) > kLdsPerCU / 2) return false;
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
  // Periods before T on rows of 256 points and more: the diagonal form (cash_diag_kernel) -- DIAG_R consecutive actions
  // per wave read ONE staged row segment per demand step -- when the shifts of neighbouring actions on a diagonal,
  // q (price - v) apart, keep a block's reads within DIAG_CAP points of each other.  SDPGPU_CASH_DIAG=0 turns it off.
  {
    const char* env = std::getenv("SDPGPU_CASH_DIAG");
    const bool diag_on = !env || std::atoi(env) != 0;
    const double slide = std::fabs(C.q * (d.price - d.unit_order_cost) * d.step), brk = std::fabs(C.q * d.fixed_order_cost);
    const int64_t rows = (hi - 1) / p.g.nc - lo / p.g.nc + 1;
    const int n_blocks = ((int)d.max_order_quantity + 1 + sdp::DIAG_R - 1) / sdp::DIAG_R;
    const int n_steps = (p.nD + sdp::DIAG_R - 1 + 1) & ~1;  // (even: the step loop is unrolled by two)
    const size_t head_bytes = (256 + (size_t)rows * n_blocks * 8 + 255) & ~(size_t)255;  // the overflow word, the blocks' bounds
    const size_t table_bytes = (size_t)rows * n_blocks * n_steps * sizeof(sdp::DiagStep) + head_bytes;
    // every shift a whole number of grid steps before Math.round (then the spread bound needs no rounding margin)
    auto whole = [&](double v) { return v * C.q == std::floor(v * C.q); };
    const bool exact = whole(d.price * d.step) && whole(d.unit_order_cost * d.step) && whole(d.holding_cost * d.step) &&
                       whole(d.fixed_order_cost) && whole(p.overhead) && whole(d.price * h->pmf_d[period - 1][0]) &&
                       whole(d.price * p.g.x_lo) && whole(d.holding_cost * p.g.x_lo) && whole(d.holding_cost * h->pmf_d[period - 1][0]);
    const double spread_bound = brk + slide * (sdp::DIAG_R - 1) + (exact ? 0 : 2);
    // The diagonal form pairs action k + i with demand j + i and relies on all of them leaving the SAME inventory behind
    // (one staged row per step): that needs consecutive demand values one `step` apart.  A support with gaps ({0, 2, 3, 7})
    // breaks it -- round 2 took the diagonal kernel there and produced wrong tables (found by round 3's F3 / F4 bridge
    // instance, tests/test_oracle_kat.py); such periods run on the uniform-shift kernel below.
    bool unit_stride = true;
    {
      const std::vector<double>& dv = h->pmf_d[period - 1];
      for (size_t j = 1; j < dv.size(); ++j) unit_stride = unit_stride && dv[j] - dv[j - 1] == d.step;
    }
    bool table_ok = diag_on && unit_stride && period < h->T && p.g.nc >= 256 && spread_bound <= sdp::DIAG_CAP &&
                    table_bytes <= ((size_t)2 << 30);
    if (table_ok && h->diag_bytes < table_bytes) {
      hipError_t e = hipStreamSynchronize(st);  // (an earlier launch may still read the old table)
      if (e != hipSuccess) return e;
      if (h->d_diag) (void)hipFree(h->d_diag);
      h->d_diag = nullptr;
      h->diag_bytes = 0;
      if (hipMalloc(&h->d_diag, table_bytes) == hipSuccess) {
        h->diag_bytes = table_bytes;
      } else {  // no room for the table: the uniform-shift kernel below needs none
        (void)hipGetLastError();
        h->d_diag = nullptr;
        table_ok = false;
      }
    }
    if (table_ok) {
      int S = 1;  // (two tiles per wave, 176 VGPRs: 48.4 against 44.3 ms per sweep on configs[2]; opt-in)
      if (const char* e = std::getenv("SDPGPU_CASH_DIAG_S")) S = std::atoi(e) == 2 ? 2 : 1;
      const int TSZ = 128 * S;
      C.tiles_per_row = (int32_t)((p.g.nc + TSZ - 1) / TSZ);
      const int64_t row_lo = lo / p.g.nc;
      C.row0 = (int32_t)row_lo;
      if (!grid_ok(8 * ((rows + 7) / 8) * (int64_t)C.tiles_per_row)) return hipErrorInvalidValue;
      {
        hipError_t e = hipMemsetAsync(h->d_diag, 0, head_bytes, st);  // (bounds are reduced with atomicMax from zero)
        if (e != hipSuccess) return e;
      }
      int* overflow = reinterpret_cast<int*>(h->d_diag);
      int* bounds = reinterpret_cast<int*>(reinterpret_cast<char*>(h->d_diag) + 256);
      sdp::DiagStep* table = reinterpret_cast<sdp::DiagStep*>(reinterpret_cast<char*>(h->d_diag) + head_bytes);
      sdp::DiagParams Q{};
      Q.C = C;
      Q.n_blocks = n_blocks;
      Q.n_steps = n_steps;
      Q.n_rows = (int32_t)rows;
      Q.cap = spread_bound <= 64 ? 64 : sdp::DIAG_CAP;  // 64-entry pieces staged per step: 2 S + 1 or 2 S + 2
      const int64_t entries = rows * n_blocks * n_steps;
      if (!grid_ok((entries + 255) / 256)) return hipErrorInvalidValue;
      hipLaunchKernelGGL(sdp::cash_diag_table_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, st, Q, (int)rows,
                         table, pmf_d, pmf_p, overflow, bounds);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      // block -> (row, tile): rows i, i + 8, ... per XCD.  OPT-IN SDPGPU_CASH_DIAG_BANDS=1: cash bands per XCD (rows of 16 tiles
      // and more) -- measured on configs[2] (profiles/r03_diag_bands.txt): the fabric traffic falls from 12.6 to 4.4 GB per
      // launch and the L2 hit rate rises from 77 to 95 %, and the sweep takes 39.4 instead of 38.1 ms: the kernel is bound by
      // its LDS reads, not by the fabric, and with bands every XCD streams every row's step records.
      const bool bands_off = !(std::getenv("SDPGPU_CASH_DIAG_BANDS") && std::atoi(std::getenv("SDPGPU_CASH_DIAG_BANDS")) == 1);
      Q.band_tiles = (!bands_off && C.tiles_per_row >= 16) ? (C.tiles_per_row + 7) / 8 : 0;
      if (Q.band_tiles > 0 && !grid_ok(8LL * rows * Q.band_tiles)) return hipErrorInvalidValue;
      const dim3 grid(Q.band_tiles > 0 ? (unsigned)(8LL * rows * Q.band_tiles) : (unsigned)(8 * ((rows + 7) / 8) * C.tiles_per_row));
      const size_t smem = (size_t)4 * 2 * (TSZ + sdp::DIAG_CAP) * 8 + (size_t)4 * TSZ * (sizeof(double) + sizeof(int));
      h->per[period - 1].ops_cell = 3.0;  // acc += T1; acc += (p gamma) * V
      {  // one 8-byte LDS read per cell; per step and wave NU pieces of 64 entries are loaded (L1) and stored (LDS) for DIAG_R * TSZ cells
        const double stage = 8.0 * 64 * (narrow_pieces(Q.cap, S)) / (double)(sdp::DIAG_R * TSZ);
        h->per[period - 1].lds_cell = 8.0 + stage;
        h->per[period - 1].l1_cell = stage;
      }
#define SDP_DIAG(MX, SS, NN) \
  hipLaunchKernelGGL((sdp::cash_diag_kernel<MX, SS, NN>), grid, dim3(256), smem, st, Q, table, bounds, v_next, v_cur, pol, lo, hi)
      const bool narrow = Q.cap == 64;
      if (P.maxdir) {
        if (S == 2) { if (narrow) SDP_DIAG(true, 2, 5); else SDP_DIAG(true, 2, 6); }
        else { if (narrow) SDP_DIAG(true, 1, 3); else SDP_DIAG(true, 1, 4); }
      } else {
        if (S == 2) { if (narrow) SDP_DIAG(false, 2, 5); else SDP_DIAG(false, 2, 6); }
        else { if (narrow) SDP_DIAG(false, 1, 3); else SDP_DIAG(false, 1, 4); }
      }
#undef SDP_DIAG
      e = hipGetLastError();
      if (e == hipSuccess && std::getenv("SDPGPU_CASH_DIAG_CHECK")) {  // tests: the guard word behind the spread bound
        int word = 0;
        e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(&word, overflow, sizeof(int), hipMemcpyDeviceToHost);
        if (e == hipSuccess && word != 0) return hipErrorAssert;
      }
      return e;
    }
  }
  // points per lane (W) and tiles per wave (S), see cash_shift_kernel.  W = 2 whenever the row has two points: the
  // gather unit is the bound and a 16-byte gather serves two cells (configs[2]: 101 -> 64 ms per sweep).  With the
  // gathers halved the LDS broadcasts of the operands show: S tiles share them (64 -> 57 ms at S = 2) where the grid
  // is large enough to keep the chip full.
  int W = p.g.nc >= 2 ? 2 : 1, S = 1;
  const int64_t n_rows_launch = (hi - 1) / p.g.nc - lo / p.g.nc + 1;
  for (int s_try : {4, 2})
    if (S == 1 && n_rows_launch * ((p.g.nc + 128 * s_try - 1) / (128 * s_try)) >= 2048) S = s_try;
  if (const char* e = std::getenv("SDPGPU_CASH_W")) W = std::atoi(e) == 1 ? 1 : W;
  if (const char* e = std::getenv("SDPGPU_CASH_S")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4) S = v;
  }
  const int TSZ = 64 * S * W;
  C.tiles_per_row = (int32_t)((p.g.nc + TSZ - 1) / TSZ);
  const int64_t row_lo = lo / p.g.nc, row_hi = (hi - 1) / p.g.nc;
  C.row0 = (int32_t)row_lo;
  if (!grid_ok((row_hi - row_lo + 1) * (int64_t)C.tiles_per_row)) return hipErrorInvalidValue;
  dim3 grid((unsigned)((row_hi - row_lo + 1) * C.tiles_per_row));
  const size_t smem = cash_shift_lds(p.nD, TSZ);  // (above 64 KiB -- pmfs of ~450 points and more -- the launch raises the kernel's limit)
  const bool last = period == h->T;
  h->per[period - 1].ops_cell = last ? 1.0 : 3.0;  // acc += T1; acc += (p gamma) * V
  h->per[period - 1].l1_cell = last ? 0.0 : 8.0;      // one 8-byte entry per cell through the vector L1 (16-byte gathers of pairs)
#define SDP_CS(MX, LS, SS, WW)                                                                                              \
  do {                                                                                                                     \
    static LdsMark mark;                                                                                                   \
    hipError_t ea = lds_allow(sdp::cash_shift_kernel<MX, LS, SS, WW>, smem, &mark);                                        \
    if (ea != hipSuccess) return ea;                                                                                       \
    hipLaunchKernelGGL((sdp::cash_shift_kernel<MX, LS, SS, WW>), grid, dim3(256), smem, st, C, v_next, v_cur, pol, pmf_d,  \
                       pmf_p, lo, hi);                                                                                     \
  } while (0)
#define SDP_CS_S(MX, LS)      \
  if (S == 4 && W == 2)       \
    SDP_CS(MX, LS, 4, 2);     \
  else if (S == 2 && W == 2)  \
    SDP_CS(MX, LS, 2, 2);     \
  else if (S == 4)            \
    SDP_CS(MX, LS, 4, 1);     \
  else if (S == 2)            \
    SDP_CS(MX, LS, 2, 1);     \
  else if (W == 2)            \
    SDP_CS(MX, LS, 1, 2);     \
  else                        \
    SDP_CS(MX, LS, 1, 1)
  if (P.maxdir) {
    if (last) { SDP_CS_S(true, true); } else { SDP_CS_S(true, false); }
  } else {
    if (last) { SDP_CS_S(false, true); } else { SDP_CS_S(false, false); }
  }
#undef SDP_CS_S
#undef SDP_CS
  return hipGetLastError();
}

// ---- cash row kernel (F3-F6 on any cash grid) -------------------------------------------------------
// LDS of a workgroup of cash_row_kernel / cash_row_pair_kernel: 152 B per demand point (its per-wave entries), the
// read-out scratch of four waves' tiles, and the per-wave trip flags.
size_t cash_row_lds(int nD, int tile_pts) {
  const size_t slots = (size_t)sdp::cash_row_slots(nD);  // (entries and flags of a second action per wave: the pair kernel's setup)
  return (size_t)nD * (24 + 128 * slots) + (size_t)4 * tile_pts * (sizeof(double) + sizeof(int)) +
         4 * slots * (((size_t)nD + 3) / 4 + 3) * sizeof(int);
}

bool cash_row_eligible(const sdpgpu_handle* h, int period) {
  const sdpgpu_desc& d = h->d;
  if (h->custom || !has_cash(d.family) || !d.clamp_inventory || !h->use_cash_row) return false;
  const PeriodInfo& p = h->per[period - 1];
  if (p.g.nc < 32) return false;                       // a wave is 64 consecutive cash points of one row
  if (p.S >= 2147483647LL) return false;               // 32-bit row offsets
  // per-wave entries of every demand point in LDS, plus the read-out scratch of the LARGEST tile a launch may pick (256
  // points, cash_row_pair_kernel with two tiles per wave): two workgroups of that size still share a compute unit
  if (cash_row_lds(p.nD, 256) > kLdsPerCU / 2) return false;
  if (period < h->T) {
    // the kernel addresses V_{t+1} by a 32-bit BYTE offset from a scalar base (per pipeline plane for F5) ...
    const PeriodInfo& n = h->per[period];
    if ((n.g.nx * n.g.nc + 2 * n.g.nc) * 8 >= 2147483647LL) return false;
    // ... and clamps the cash key AFTER the quantiser (cash_key_row): the rounded balance times the quantiser's factor
    // must stay far inside int32.  A generous bound on |cash + increment| from the parameters:
    const double ymax = std::fabs(d.max_inventory) + std::fabs(d.min_inventory) + d.max_order_quantity * d.step +
                        (d.cash_formula == 2 ? std::fabs(d.max_cash) / std::max(d.unit_order_cost, 1e-300) : 0.0);
    const double cmax = std::max(std::fabs(d.min_cash), std::fabs(d.max_cash));
    const double spend = std::fabs(d.fixed_order_cost) + std::fabs(d.unit_order_cost) * ymax + std::fabs(p.overhead);
    const double rate = std::max({std::fabs(d.r0), std::fabs(d.r2), std::fabs(d.r3), std::fabs(d.deposit_rate)});
    double bound = (cmax + spend) * (1.0 + rate) + std::fabs(d.overdraft_limit) * rate +
                   (std::fabs(d.price) + std::fabs(d.holding_cost) + std::fabs(d.salvage_value)) * ymax;
    bound *= 1.0 + std::fabs(d.penalty_cost);
    if (!(bound * d.cash_round_mult < 5.0e8)) return false;  // (NaN fails too) -> generic kernel
  }
  return true;
}

}  // namespace sdpgpu_detail
namespace sdpgpu_detail {

template <int FAM, bool FORMULA1, bool PEN = false, bool LEAN = false, bool UNI = false>
hipError_t launch_cash_row_fam(const DevParams& P, bool last, bool intdiv, const double* v_next, double* v_cur,
                               int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi,
                               int64_t row0, sdp::RowTiling G, dim3 grid, size_t smem, hipStream_t st) {
#define SDP_CR(LS, ID)                                                                                                    \
  do {                                                                                                                    \
    static LdsMark mark;                                                                                                  \
    hipError_t ea = lds_allow(sdp::cash_row_kernel<FAM, LS, FORMULA1, ID, PEN, LEAN, UNI && !(ID)>, smem, &mark);         \
    if (ea != hipSuccess) return ea;                                                                                      \
    hipLaunchKernelGGL((sdp::cash_row_kernel<FAM, LS, FORMULA1, ID, PEN, LEAN, UNI && !(ID)>), grid, dim3(256), smem, st, \
                       P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, row0, G);                                             \
  } while (0)
  if (intdiv) {
    if (last) SDP_CR(true, true); else SDP_CR(false, true);
  } else {
    if (last) SDP_CR(true, false); else SDP_CR(false, false);
  }
#undef SDP_CR
  return hipGetLastError();
}

hipError_t launch_cash_row(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                           int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st,
                           bool levels_only) {
  if (hi <= lo) return hipSuccess;
  const PeriodInfo& p = h->per[period - 1];
  const int64_t row_lo = lo / p.g.nc, row_hi = (hi - 1) / p.g.nc;
  const bool last = period == h->T;
  const bool intdiv = h->d.cash_round_int_div && h->d.cash_round_div != 1.0;
  // uniform-key trips (see cash_row_kernel): the cash balance cancels out of the increment only without a deposit rate
  const bool uni_off = std::getenv("SDPGPU_CASH_UNI") && std::atoi(std::getenv("SDPGPU_CASH_UNI")) == 0;
  const bool pair_off = std::getenv("SDPGPU_CASH_PAIR") && std::atoi(std::getenv("SDPGPU_CASH_PAIR")) == 0;
  const bool uni = P.family == sdp::FAM_CASH && !uni_off && P.pi == 0.0 && !intdiv &&
                   (P.cash_formula == 1 || h->d.deposit_rate == 0.0);
  // two adjacent cash points per lane (cash_row_pair_kernel): rows of two 128-point tiles and more
  const bool pair_f3 = uni && !pair_off && p.g.nc >= 256 && (last || h->per[period].g.nc >= 2);
  // round 4: the overdraft families F4 / F5 on the same kernel (uniform-key trips wherever a tile's balances pay no interest
  // under the order; quantiser and one gather per point elsewhere) -- not with the `/ 10` long division of CashOverdraft.main's
  // quantiser, whose key is not the rounded balance.  SDPGPU_CASH_OD_PAIR=0: the one-point row kernel of rounds 1-3.
  const bool od_off = std::getenv("SDPGPU_CASH_OD_PAIR") && std::atoi(std::getenv("SDPGPU_CASH_OD_PAIR")) == 0;
  const bool od_pair = (P.family == sdp::FAM_OVERDRAFT || P.family == sdp::FAM_CASH_LEADTIME) && !intdiv && !od_off && !pair_off &&
                       p.g.nc >= 256 && (last || h->per[period].g.nc >= 2);
  const bool pair = pair_f3 || od_pair;
  // ... and S such tiles per wave (per-action setup and entry reads shared) where that still leaves a few thousand workgroups.
  // CashConstraint.main's grid, after the clamp-free trips became straight-line code: S = 1 / 2 -> 46.9 / 39.7 ms per sweep
  // (before: 61.7 / 63.7; S = 4 needs 260 VGPRs: 96 ms, not instantiated).  SDPGPU_CASH_PAIR_S=1|2 overrides.
  int pair_s = 1;
  if (pair && p.g.nc >= 512 && (row_hi - row_lo + 1) * ((p.g.nc + 255) / 256) >= 2048) pair_s = 2;
  if (const char* e = std::getenv("SDPGPU_CASH_PAIR_S")) {
    pair_s = std::atoi(e) == 2 ? 2 : 1;
  }
  const int tile_pts = pair ? 128 * pair_s : 64;
  // OPT-IN (SDPGPU_CASH_SHARE=1): SHARED blocks (cash_row_pair_kernel, SRC 2) -- a workgroup is four tiles of one row, one per
  // wave, and forms every action's block of wave-uniform operands once for the four (rows of four tiles and more).  Built to
  // take the per-action setup (12 % of the kernel's instructions) out of the hot loop; measured SLOWER on CashConstraint.main's
  // grid, 49-50 against 39-40 ms per sweep: with one tile per workgroup the four waves run CONSECUTIVE actions on the same
  // cash window, whose gathers land on the same next-inventory rows 90 keys apart and share the compute unit's vector-L1 lines;
  // four different tiles share nothing (profiles/r03_cash_share_l1.txt).  The default stays one tile per workgroup.
  const bool share_off = !(std::getenv("SDPGPU_CASH_SHARE") && std::atoi(std::getenv("SDPGPU_CASH_SHARE")) == 1);
  const bool tab_on = std::getenv("SDPGPU_CASH_TAB") && std::atoi(std::getenv("SDPGPU_CASH_TAB")) == 1;
  const size_t blk_bytes = (size_t)sdp::row_tab_block(p.nD);
  const size_t smem_share = (size_t)p.nD * 16 + 4 * blk_bytes + (size_t)p.nD * 8 + 16;
  const bool share = pair_f3 && !share_off && !tab_on && p.g.nc > 3 * (int64_t)tile_pts && smem_share <= kLdsPerCU / 2;
  const int wg_pts = share ? 4 * tile_pts : tile_pts;  // cash points per workgroup
  // F5: rows in order of the level x + preQ (RowTiling::perm); SDPGPU_CASH_ROWPERM=0 keeps the (preQ, x) order
  const bool level_order = P.family == sdp::FAM_CASH_LEADTIME && p.g.nq > 1 &&
                           !(std::getenv("SDPGPU_CASH_ROWPERM") && std::atoi(std::getenv("SDPGPU_CASH_ROWPERM")) == 0);
  sdp::RowTiling G{};
  // two actions per setup pass of the pair kernel when a pmf fits half a wave (SDPGPU_CASH_SLOTS=1: one, as in round 2)
  const char* slots_env = std::getenv("SDPGPU_CASH_SLOTS");
  G.slots = (slots_env && std::atoi(slots_env) == 1) ? 1 : sdp::cash_row_slots(p.nD);
  G.tiles_per_row = (int32_t)((p.g.nc + wg_pts - 1) / wg_pts);
  G.n_rows = (int32_t)(row_hi - row_lo + 1);
  // F5 on the pair kernel: a workgroup is one tile of FOUR rows consecutive in the level order, one row per wave (RW = 4, see
  // cash_row_pair_kernel): rows of one level gather the same entries, and on one compute unit they share its vector L1.
  // SDPGPU_CASH_RW=1: one row per workgroup.
  const bool rw4 = od_pair && level_order && !(std::getenv("SDPGPU_CASH_RW") && std::atoi(std::getenv("SDPGPU_CASH_RW")) == 1);
  if (levels_only) {  // (the caller has checked: F5, the whole grid in one slab) one representative row per level x + preQ
    if (!level_order || row_lo != 0 || row_hi != p.g.nx * p.g.nq - 1) return hipErrorInvalidValue;
    G.n_rows = (int32_t)(p.g.nx + p.g.nq - 1);
  }
  G.rows_real = G.n_rows;
  if (rw4) G.n_rows = (G.n_rows + 3) / 4;
  int64_t blocks = (int64_t)G.n_rows * G.tiles_per_row;
  // cash bands per XCD (see RowTiling): rows of 16 tiles and more; ~1280 cash points per band (CashConstraint.main, 313
  // 64-point tiles per row: 1 / 2 / 3 / 6 bands per XCD = 70.0 / 69.0 / 73.4 / 72.8 ms per sweep, row-major 119.8).
  // SDPGPU_CASH_BANDS=0 keeps the plain row-major numbering, =n forces n bands per XCD.
  int nsub = -1;
  if (const char* e = std::getenv("SDPGPU_CASH_BANDS")) nsub = std::atoi(e);
  if (share && nsub != 0 && (int64_t)G.n_rows * G.tiles_per_row >= 64) {
    // few, wide workgroup tiles: equal runs of the column-major (tile, row) order per XCD (RowTiling::colmajor)
    const int64_t units = (int64_t)G.n_rows * G.tiles_per_row;
    G.colmajor = (int32_t)((units + 7) / 8);
    blocks = 8LL * G.colmajor;
  } else if (nsub != 0 && G.tiles_per_row >= 16 && row_hi - row_lo + 1 < (1LL << 24)) {
    const int tpb = (G.tiles_per_row + 7) / 8;  // tiles per XCD and row
    // (F5 walks its rows in order of the level x + preQ, below: the rows of a level gather the same entries, and with bands of
    // two tiles ~130 rows are in flight on an XCD -- four levels, whose windows fit its L2.  SingleProductLeadtime's size:
    // 20 / 5 / 2 / 1 tiles per band = 293 / 220 / 185 / 185 ms per sweep; without the level order 291 / 416 / 374 / 375.)
    const int per_band = level_order ? std::max(1, 128 / wg_pts) : std::max(1, 1280 / wg_pts);  // tiles per band (F5: ~128 cash points)
    G.nsub = nsub > 0 ? std::min(nsub, tpb) : std::max(1, (tpb + per_band / 2) / per_band);
    G.tps = (tpb + G.nsub - 1) / G.nsub;
    // bands i, i + 8, ... per XCD for the families whose cells below a zero balance cost twice the others' (F4 / F5 / F6:
    // SingleProductLeadtime's size on the band numbering 177.7 -> 131.3 ms per sweep); F3's tiles cost about the same along the
    // axis and neighbouring bands share window lines in the XCD's L2 (CashConstraint.main: 36.4 contiguous, 37.1 interleaved)
    G.band_interleave = P.family != sdp::FAM_CASH;
    if (const char* e = std::getenv("SDPGPU_CASH_BAND_INTERLEAVE")) G.band_interleave = std::atoi(e) != 0;
    blocks = 8LL * G.nsub * G.tps * G.n_rows;
  }
  if (level_order) {
    const int64_t key[4] = {row_lo, levels_only ? -2 - row_hi : row_hi, p.g.nx, p.g.nq};
    if (std::memcmp(key, h->rowperm_key, sizeof key) != 0) {
      const int64_t n = levels_only ? p.g.nx + p.g.nq - 1 : row_hi - row_lo + 1;
      std::vector<int32_t>& perm = h->rowperm_host;
      hipError_t e = hipStreamSynchronize(st);  // (an earlier launch may still read the old order)
      if (e != hipSuccess) return e;
      perm.resize((size_t)n);
      if (levels_only) {  // level y's representative: the row (preQ index, inventory index) = (max(0, y - (nx - 1)), y - that)
        for (int64_t y = 0; y < n; ++y) {
          const int64_t iq = std::max<int64_t>(0, y - (p.g.nx - 1));
          perm[(size_t)y] = (int32_t)(iq * p.g.nx + (y - iq));
        }
      } else {
        for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = (int32_t)i;
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) {
          const int64_t ra = row_lo + a, rb = row_lo + b;
          return ra % p.g.nx + ra / p.g.nx < rb % p.g.nx + rb / p.g.nx;
        });
      }
      h->units_key[0] = -1;  // (the diagonal unit order is built from this row order)
      if (h->d_rowperm) (void)hipFree(h->d_rowperm);
      h->d_rowperm = nullptr;
      e = hipMalloc((void**)&h->d_rowperm, (size_t)n * sizeof(int32_t));
      if (e == hipSuccess) e = hipMemcpy(h->d_rowperm, perm.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice);
      if (e != hipSuccess) return e;
      std::memcpy(h->rowperm_key, key, sizeof key);
    }
    G.perm = h->d_rowperm;
  }
  // F5 on the pair kernel: the (row group, tile) units in DIAGONAL order (RowTiling::units).  SDPGPU_CASH_DIAG_ORDER=0: bands.
  const bool diag_order = od_pair && level_order && !last && G.tiles_per_row >= 8 &&
                          !(std::getenv("SDPGPU_CASH_DIAG_ORDER") && std::atoi(std::getenv("SDPGPU_CASH_DIAG_ORDER")) == 0);
  if (diag_order) {
    const int64_t n_units = (int64_t)G.n_rows * G.tiles_per_row;
    double sigma = std::fabs(h->d.price) * h->d.cash_round_mult * h->d.step;  // keys per unit of level along a diagonal
    if (const char* e = std::getenv("SDPGPU_CASH_DIAG_SIGMA")) sigma = std::atof(e);  // (experiments: 0 = tile-major, levels in order)
    int64_t sig_bits;
    std::memcpy(&sig_bits, &sigma, sizeof sig_bits);
    const int64_t key[8] = {row_lo, row_hi, p.g.nx, p.g.nq, (int64_t)G.tiles_per_row, (int64_t)wg_pts, rw4 ? 4 : 1, sig_bits};
    if (std::memcmp(key, h->units_key, sizeof key) != 0) {
      hipError_t e = hipStreamSynchronize(st);  // (an earlier launch may still read the old order)
      if (e != hipSuccess) return e;
      const std::vector<int32_t>& perm = h->rowperm_host;  // rows in level order (built above)
      const int rw = rw4 ? 4 : 1;
      struct Unit {
        double diag;
        int32_t level, grp, tile;
      };
      std::vector<Unit> us((size_t)n_units);
      size_t at = 0;
      for (int32_t g = 0; g < G.n_rows; ++g) {
        const int64_t r = row_lo + perm[(size_t)g * rw];  // the group's first row: its level stands for the group
        const int32_t level = (int32_t)(r % p.g.nx + r / p.g.nx);
        for (int32_t t = 0; t < G.tiles_per_row; ++t) us[at++] = Unit{(double)t * wg_pts + sigma * level, level, g, t};
      }
      std::stable_sort(us.begin(), us.end(), [](const Unit& a, const Unit& b) {
        return a.diag != b.diag ? a.diag < b.diag : (a.level != b.level ? a.level < b.level : a.grp < b.grp);
      });
      std::vector<int2> host((size_t)n_units);
      for (size_t i = 0; i < host.size(); ++i) host[i] = make_int2(us[i].grp, us[i].tile);
      if (h->d_units) (void)hipFree(h->d_units);
      h->d_units = nullptr;
      e = hipMalloc((void**)&h->d_units, host.size() * sizeof(int2));
      if (e == hipSuccess) e = hipMemcpy(h->d_units, host.data(), host.size() * sizeof(int2), hipMemcpyHostToDevice);
      if (e != hipSuccess) return e;
      std::memcpy(h->units_key, key, sizeof key);
    }
    G.units = h->d_units;
    G.units_total = (int32_t)n_units;
    int seg = 64;  // (SingleProductLeadtime's size: 8 / 64 / 256 / 1024 units per segment = 124.7 / 123.9 / 126.5 / 133.9 ms per sweep; one contiguous eighth per XCD: 160)
    if (const char* e = std::getenv("SDPGPU_CASH_DIAG_SEG")) seg = std::max(1, std::atoi(e));
    G.units_seg = seg;
    const int64_t n_seg = (n_units + seg - 1) / seg;
    blocks = 8LL * ((n_seg + 7) / 8) * seg;
  }
  if (!grid_ok(blocks)) return hipErrorInvalidValue;
  dim3 grid((unsigned)blocks);
  // (the scratch grows with the tile: above 64 KiB -- pmfs of 340 points and more on 128- or 256-point tiles -- the launch
  // raises the kernel's dynamic-LDS limit; cash_row_eligible has bounded it by half a compute unit's LDS)
  const size_t smem = cash_row_lds(p.nD, tile_pts);
  // LEAN: `- holdCosts - overheadCost` subtract +0.0 in every cell (holdingCost and the period's overhead are +0.0)
  const bool lean = P.family == sdp::FAM_CASH && P.cash_formula != 1 && P.pi == 0.0 && h->d.holding_cost == 0.0 &&
                    !std::signbit(h->d.holding_cost) && P.overhead == 0.0 && !std::signbit(P.overhead);
#define SDP_ROWARGS P, last, intdiv, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, row_lo, G, grid, smem, st
  switch (P.family) {
    case sdp::FAM_CASH: {
      // (cash_formula 2, the (x, R) state of CashConstraintXR: formula 0's increment on initCash = R - variCost * x)
      if (pair) {
#define SDP_PAIR_GO(LS, F1, LN, SS, SRC)                                                                                        \
  do {                                                                                                                         \
    static LdsMark mark;                                                                                                       \
    hipError_t ea = lds_allow(sdp::cash_row_pair_kernel<LS, F1, LN, SS, SRC>, smem_pair, &mark);                               \
    if (ea != hipSuccess) return ea;                                                                                           \
    hipLaunchKernelGGL((sdp::cash_row_pair_kernel<LS, F1, LN, SS, SRC>), grid, dim3(256), smem_pair, st, P, v_next, v_cur, pol, \
                       pmf_d, pmf_p, lo, hi, row_lo, G, tab_ptr, tab_actions);                                                 \
  } while (0)
#define SDP_PAIR(LS, F1, LN)                                                                                                  \
  do {                                                                                                                        \
    if (use_tab) {                                                                                                            \
      const int64_t waves = (int64_t)G.n_rows * tab_actions;                                                                  \
      hipLaunchKernelGGL((sdp::cash_row_table_kernel<LS, F1, LN>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, P,    \
                         pmf_d, h->d_rowtab, row_lo, G.n_rows, tab_actions);                                                  \
      hipError_t et = hipGetLastError();                                                                                      \
      if (et != hipSuccess) return et;                                                                                        \
      if (pair_s == 2) SDP_PAIR_GO(LS, F1, LN, 2, 1); else SDP_PAIR_GO(LS, F1, LN, 1, 1);                                     \
    } else if (share) {                                                                                                       \
      if (pair_s == 2) SDP_PAIR_GO(LS, F1, LN, 2, 2); else SDP_PAIR_GO(LS, F1, LN, 1, 2);                                     \
    } else {                                                                                                                  \
      if (pair_s == 2) SDP_PAIR_GO(LS, F1, LN, 2, 0); else SDP_PAIR_GO(LS, F1, LN, 1, 0);                                     \
    }                                                                                                                         \
  } while (0)
        // OPT-IN (SDPGPU_CASH_TAB=1; measured slower, see cash_row_pair_kernel): the (row, action) blocks from a table formed
        // once per row by cash_row_table_kernel -- rows of two tiles and more, formulas 0 and 1 (the (x, R) state has its own
        // action range), while the two block buffers per wave still leave two workgroups per compute unit and the table
        // stays under 1 GiB.
        const int tab_actions = h->n_actions_full;
        const size_t blk = blk_bytes;
        const size_t smem_tab = (size_t)p.nD * 16 + 8 * blk + (size_t)4 * tile_pts * (sizeof(double) + sizeof(int));
        const size_t tab_bytes = (size_t)G.n_rows * (size_t)tab_actions * blk;
        bool use_tab = tab_on && P.cash_formula != 2 && G.tiles_per_row >= 2 && smem_tab <= kLdsPerCU / 2 &&
                       tab_bytes <= ((size_t)1 << 30) && grid_ok(((int64_t)G.n_rows * tab_actions + 3) / 4);
        if (use_tab && h->rowtab_bytes < tab_bytes) {
          hipError_t e = hipStreamSynchronize(st);  // (an earlier launch may still read the old table)
          if (e != hipSuccess) return e;
          if (h->d_rowtab) (void)hipFree(h->d_rowtab);
          h->d_rowtab = nullptr;
          h->rowtab_bytes = 0;
          if (hipMalloc((void**)&h->d_rowtab, tab_bytes) == hipSuccess) {
            h->rowtab_bytes = tab_bytes;
          } else {  // no room: the in-kernel setup needs no table
            (void)hipGetLastError();
            use_tab = false;
          }
        }
        const char* tab_ptr = use_tab ? h->d_rowtab : nullptr;
        const size_t smem_pair = use_tab ? smem_tab : (share ? smem_share : smem);
        if (P.cash_formula == 1) {
          if (last) SDP_PAIR(true, true, false); else SDP_PAIR(false, true, false);
        } else if (lean) {
          if (last) SDP_PAIR(true, false, true); else SDP_PAIR(false, false, true);
        } else {
          if (last) SDP_PAIR(true, false, false); else SDP_PAIR(false, false, false);
        }
#undef SDP_PAIR
#undef SDP_PAIR_GO
        return hipGetLastError();
      }
      if (uni) {
        if (lean) return launch_cash_row_fam<sdp::FAM_CASH, false, false, true, true>(SDP_ROWARGS);
        return P.cash_formula != 1 ? launch_cash_row_fam<sdp::FAM_CASH, false, false, false, true>(SDP_ROWARGS)
                                   : launch_cash_row_fam<sdp::FAM_CASH, true, false, false, true>(SDP_ROWARGS);
      }
      if (lean) return launch_cash_row_fam<sdp::FAM_CASH, false, false, true>(SDP_ROWARGS);
      if (P.pi != 0.0)
        return P.cash_formula != 1 ? launch_cash_row_fam<sdp::FAM_CASH, false, true>(SDP_ROWARGS)
                                   : launch_cash_row_fam<sdp::FAM_CASH, true, true>(SDP_ROWARGS);
      return P.cash_formula != 1 ? launch_cash_row_fam<sdp::FAM_CASH, false>(SDP_ROWARGS)
                                 : launch_cash_row_fam<sdp::FAM_CASH, true>(SDP_ROWARGS);
    }
#define SDP_OD_GO(FM, LS, SS, RR)                                                                                                \
  do {                                                                                                                          \
    static LdsMark mark;                                                                                                        \
    hipError_t ea = lds_allow(sdp::cash_row_pair_kernel<LS, false, false, SS, 0, FM, RR>, smem, &mark);                         \
    if (ea != hipSuccess) return ea;                                                                                            \
    hipLaunchKernelGGL((sdp::cash_row_pair_kernel<LS, false, false, SS, 0, FM, RR>), grid, dim3(256), smem, st, P, v_next,     \
                       v_cur, pol, pmf_d, pmf_p, lo, hi, row_lo, G, (const char*)nullptr, 0);                                  \
  } while (0)
#define SDP_OD(FM, RR)                                                                  \
  do {                                                                                  \
    if (last) {                                                                         \
      if (pair_s == 2) SDP_OD_GO(FM, true, 2, RR); else SDP_OD_GO(FM, true, 1, RR);     \
    } else {                                                                            \
      if (pair_s == 2) SDP_OD_GO(FM, false, 2, RR); else SDP_OD_GO(FM, false, 1, RR);   \
    }                                                                                   \
    return hipGetLastError();                                                           \
  } while (0)
    case sdp::FAM_OVERDRAFT:
      if (od_pair) SDP_OD(sdp::FAM_OVERDRAFT, 1);
      return launch_cash_row_fam<sdp::FAM_OVERDRAFT, false>(SDP_ROWARGS);
    case sdp::FAM_CASH_LEADTIME:
      if (od_pair && rw4) SDP_OD(sdp::FAM_CASH_LEADTIME, 4);
      if (od_pair) SDP_OD(sdp::FAM_CASH_LEADTIME, 1);
      return launch_cash_row_fam<sdp::FAM_CASH_LEADTIME, false>(SDP_ROWARGS);
#undef SDP_OD
#undef SDP_OD_GO
    case sdp::FAM_SURVIVAL: return launch_cash_row_fam<sdp::FAM_SURVIVAL, false>(SDP_ROWARGS);
  }
#undef SDP_ROWARGS
  return hipErrorInvalidValue;
}

namespace {
// every row (iq, ix) of the F5 grid takes the tables of its level's representative row (launch_cash_row, levels_only)
__global__ __launch_bounds__(256) void level_fill_kernel(double* __restrict__ v, int32_t* __restrict__ pol, int64_t nx, int64_t nq,
                                                         int64_t nc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nx * nq * nc) return;
  const int64_t row = i / nc, ic = i - row * nc;
  const int64_t iq = row / nx, ix = row - iq * nx, y = ix + iq;
  const int64_t rq = y - (nx - 1) > 0 ? y - (nx - 1) : 0;
  const int64_t rep = rq * nx + (y - rq);
  if (rep == row) return;
  v[i] = v[rep * nc + ic];
  pol[i] = pol[rep * nc + ic];
}
}  // namespace

hipError_t launch_level_fill(sdpgpu_handle* h, int period, double* v_cur, int32_t* pol, hipStream_t st) {
  const PeriodInfo& p = h->per[period - 1];
  const int64_t n = p.g.nx * p.g.nq * p.g.nc;
  if (!grid_ok((n + 255) / 256)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(level_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v_cur, pol, (int64_t)p.g.nx,
                     (int64_t)p.g.nq, (int64_t)p.g.nc);
  return hipGetLastError();
}

}  // namespace sdpgpu_detail

// sdp_cash.hpp -- "uniform shift" period kernel for the cash-constrained family F3
// (CashConstraint.java:95-133, CashConstraintTesting.java:110-148) on dyadic grids.
//
// When the deposit rate, the overhead rate and the end-cash penalty are zero and every parameter
// and the cash quantum are dyadic rationals of bounded size (checked on the host, see
// cash_shift_eligible), EVERY fp64 operation of the reference's immediateValue lambda is exact.
// Its value is then the real number
//     inc(y, a, d) = price*min(y,d) - fixed(a) - v*a - h*max(y-d,0) - overhead [+ salvage*max(y-d,0) at T],
// y = x + a, which does not depend on the cash balance at all, and the transition moves every
// cash point by the same number of grid steps:
//     next inventory index = clamp(max(0, y - d)),     next cash index = clamp(ic + delta),
//     delta = Math.round(inc * q)      (q = cash points per unit; Math.round(k + z) = k + Math.round(z)).
// So for one inventory row x and 64 consecutive cash points (one wave, lane = cash index) the triple
// {p_j*inc, row offset, delta} is WAVE-UNIFORM per (action, demand): the wave computes it once per
// action (lanes = demand indices), parks it in LDS, and the per-cell work drops from ~25 fp64 + ~20
// integer operations to
//     acc += T1_j;  idx = med3(ic + delta_j, 0, nc-1);  acc += (p_j*gamma) * V[rowoff_j + idx]
// i.e. 3 fp64 + 3 integer operations and one coalesced 512-B gather per wave.  The values are
// bit-identical to the general kernel's because nothing was rounded in the first place; the
// accumulation order is untouched.  Bound: L1/TA gather rate (8 B per cell out of L2), then VALU.
#pragma once
#include <type_traits>

#include "sdp_device.hpp"

namespace sdp {

// A demand trip of the cash row kernels has two phases: (1) offsets formed, gathers issued, increments multiplied; (2) the
// gathered values consumed.  A scheduling barrier between them keeps the compiler from sinking phase-1 work under the waits of
// phase 2 (+1 % on CashConstraint.main's grid; s_setprio at the same two places -- any level, 0 included -- measured the same:
// it is the instruction order that matters, not the priority).
#define SDP_TRIP_PHASE(hi) __builtin_amdgcn_sched_barrier(0)

// The same state from its coordinates: a cash family's row r = iq * nx + ix (inventory level, pipeline plane) and cash index ic.
// The row kernels know both; decode_state's four 64-bit divisions (emulated: ~130 instructions each) cost the pair kernel a
// thousand instructions per tile.  Every value is formed by the statements of decode_state above.
template <int FAM>
__device__ __forceinline__ void decode_state_row(const DevParams& P, int64_t row, int ic, StateT& s) {
  static_assert(FAM != FAM_BACKORDER && FAM != FAM_LEADTIME, "cash families only");
#ifdef SDP_DECODE_DIV  // (A/B builds only, tools/build_variant.sh: the round-2 form)
  decode_state<FAM>(P, row * P.cur.nc + ic, s);
  return;
#endif
  s.cash = 0;
  s.preq = 0;
  int64_t ix = row, iq = 0;
  if constexpr (FAM == FAM_CASH_LEADTIME) {  // (the one cash family with planes: one division per wave, not four per point)
    iq = row / P.cur.nx;
    ix = row - iq * P.cur.nx;
  }
  s.x = P.cur.x_lo + (double)ix * P.step;
  double k = (double)(P.cur.k_lo + (int64_t)ic);
  s.cash = P.cash_round_int_div ? k : k / P.round_div;
  if constexpr (FAM == FAM_CASH_LEADTIME) s.preq = (double)iq * P.step;
  xr_state<FAM>(P, s, true);
}


struct CashShiftParams {
  double price, K, v, h, overhead, salvage, gamma, step;
  double x_lo;          // inventory value of ix = 0 (same grid every period: clamped family)
  double min_inventory, max_inventory;
  double next_x_lo;
  double q;             // cash points per unit (cash = key / q)
  int64_t k_lo;         // cash key of ic = 0
  int32_t nx, nc;
  int32_t n_demand;
  int32_t n_actions_cap;  // (int) maxOrderQuantity + 1
  double max_order_quantity;
  int32_t is_last;
  int32_t tiles_per_row;  // ceil(nc / (64 S))
  int32_t row0;           // first inventory row launched
};

// Per (action, demand) operands of the demand loop, in BYTES of V_{t+1} so that a cell's gather address is
// `v_next + clamp(ic8 + off8, lo8, hi8)`: one integer add, one v_med3_i32 and a load with a scalar base and a
// 32-bit offset (the launcher refuses tables of 4 GB and more).
struct ShiftEntry {
  double t1;     // p_j * inc
  int32_t off8;  // 8 * (next inventory index * nc + cash index shift)
  int32_t lo8;   // 8 * next inventory index * nc: byte offset of the row's first cash point
};
struct ShiftEntry2 {
  double pg;     // p_j * gamma
  int32_t hi8;   // lo8 + 8 * (nc - 1): the row's last cash point
  int32_t pad;
};

__device__ __forceinline__ int med3_i32(int x, int lo, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
}

// S = cash tiles per wave: lane l owns the points ic0 + 64 s + l, s < S (each gather stays one contiguous 512 B).
// The per-(action, demand) operands are wave-uniform, read from LDS as broadcasts; an LDS broadcast still costs the
// full 64-lane data path (32 B per step here), and at S = 1 that, not the gathers, is what bounds the loop
// (four SIMDs x 16 clk per step = 4 cells/clk/CU).  S tiles share one read.
// W = adjacent cash points per lane.  The gather unit (TA) is what bounds this loop: it is busy ~14 cycles per 64-lane
// 8-byte gather (rocprofv3 TA_BUSY: 85 % of the kernel's cycles at W = 1).  With W = 2 a lane reads its two
// neighbouring points with ONE 16-byte load -- the wave's 1 KB is contiguous -- and picks them apart again at the
// clamped ends of the row: half the gather instructions per cell.
typedef double dpair_u __attribute__((ext_vector_type(2), aligned(8)));

template <bool MAXDIR, bool LAST, int S, int W>
__global__ __launch_bounds__(256) void cash_shift_kernel(CashShiftParams P, const double* __restrict__ v_next,
                                                         double* __restrict__ v_cur, int32_t* __restrict__ pol,
                                                         const double* __restrict__ pmf_d,
                                                         const double* __restrict__ pmf_p, int64_t lo, int64_t hi) {
  constexpr int TS = 64 * S * W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int D = P.n_demand;
  double2* s_pmf = reinterpret_cast<double2*>(smem);                     // {d_j, p_j * gamma}
  const int DP = (D + 7) & ~7;  // the demand loop runs in blocks of eight; the padding entries add exact zeros
  ShiftEntry* s_ent = reinterpret_cast<ShiftEntry*>(smem + (size_t)D * 16);  // [4 waves][DP]
  ShiftEntry2* s_ent2 = reinterpret_cast<ShiftEntry2*>(smem + (size_t)D * 16 + (size_t)DP * 16 * 4);  // [4 waves][DP]
  double* s_val = reinterpret_cast<double*>(smem + (size_t)D * 16 + (size_t)DP * 16 * 8);
  int* s_k = reinterpret_cast<int*>(s_val + 4 * TS);
  int* s_flag = s_k + 4 * TS;  // [4 waves][DP / 2]: per trip of the demand loop, "no point of this wave's tiles clamps"

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int j = tid; j < D; j += 256) s_pmf[j] = make_double2(pmf_d[j], pmf_p[j] * P.gamma);
  __syncthreads();

  const int row = P.row0 + blockIdx.x / P.tiles_per_row;
  const int ic0 = (blockIdx.x % P.tiles_per_row) * TS;
  const double x = P.x_lo + (double)row * P.step;

  // feasible action count per point (CashConstraint.java:96-99) and the wave's maximum
  int ic8[S], nA[S][W];  // lane l of tile s owns the points ic0 + 64 W s + W l + w, w < W
  int nA_max = 0;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int ic_first = ic0 + 64 * W * s + W * lane;
    ic8[s] = (ic_first < P.nc ? ic_first : P.nc - 1) * 8;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int ic = ic_first + w;
      const int ic_c = ic < P.nc ? ic : P.nc - 1;
      const double cash = (double)(P.k_lo + ic_c) / P.q;  // exact: q is a power of two
      double m = jmin(P.max_order_quantity, jmax(0.0, (cash - P.overhead - P.K) / P.v));
      nA[s][w] = ((m != m) ? 0 : (int)m) + 1;
      nA_max = nA[s][w] > nA_max ? nA[s][w] : nA_max;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_xor(nA_max, off, 64);
    nA_max = o > nA_max ? o : nA_max;
  }
  nA_max = __builtin_amdgcn_readfirstlane(nA_max);

  ShiftEntry* ent = s_ent + (size_t)wave * DP;
  ShiftEntry2* ent2 = s_ent2 + (size_t)wave * DP;
  int* flag = s_flag + (size_t)wave * (DP / 2);
  if (lane < DP - D) {  // t1 = 0, p = 0, a valid address: acc += 0.0; acc += 0.0 * V[0]
    ent[D + lane] = ShiftEntry{0.0, 0, 0};
    ent2[D + lane] = ShiftEntry2{0.0, 0, 0};
  }
  constexpr int U = S == 1 ? 8 : (S == 2 ? 4 : 2);  // steps per trip: eight gathers in flight per wave
  const bool tile_whole = ic0 + TS <= P.nc;  // every lane's points exist (no lane was folded onto the row's last point)
  const int nc18 = (P.nc - 1) * 8;
  const char* vbase = reinterpret_cast<const char*>(v_next);
  double best[S][W];
  int bestk[S][W];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int w = 0; w < W; ++w) {
      best[s][w] = MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;
      bestk[s][w] = 0;
    }
  for (int k = wave; k < nA_max; k += 4) {
    // ---- per-action setup: lanes walk the demand index, every operation below is exact ----
    const double a = (double)k * P.step;
    const double y = x + a;
    const double fixed = a > 0 ? P.K : 0.0;
    const double var = P.v * a;
    for (int j = lane; j < D; j += 64) {
      const double2 dp = s_pmf[j];
      const double d = dp.x;
      const double level = y - d;
      const double pos = jmax(level, 0.0);
      double inc = P.price * jmin(y, d) - fixed - var - P.h * pos - P.overhead;
      if constexpr (LAST) inc += P.salvage * pos;
      ShiftEntry e;
      e.t1 = pmf_p[j] * inc;
      e.off8 = 0;
      e.lo8 = 0;
      if constexpr (!LAST) {
        double ninv = jmax(0.0, level);
        ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
        ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
        const int rowoff = (int)((ninv - P.next_x_lo) / P.step) * P.nc;
        // the shift is held to +-nc: anything beyond already clamps every cash point of the row to an end
        int delta = (int)jmax(jmin(jround_d(inc * P.q), (double)P.nc), -(double)P.nc);
        e.lo8 = rowoff * 8;
        e.off8 = (rowoff + delta) * 8;
        ShiftEntry2 e2;
        e2.pg = dp.y;
        e2.hi8 = e.lo8 + nc18;
        e2.pad = 0;
        ent2[j] = e2;
        if constexpr (W == 2) {
          // Does any point of this wave's tiles leave the row under this step's shift?  If none does for the U steps of
          // a trip, the trip needs neither the clamp nor the selects that pick a clamped pair apart: the wave decides
          // once per trip (a scalar branch) instead of every lane in every cell.
          const bool fast = tile_whole && ic0 + delta >= 0 && ic0 + TS - 1 + delta <= P.nc - 1;
          const unsigned long long m = __ballot(fast);
          if ((lane % U) == 0) flag[j / U] = ((m >> lane) & ((1ull << U) - 1)) == ((1ull << U) - 1);
        }
      }
      ent[j] = e;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
    if constexpr (W == 2 && !LAST) {
      // trips that hold padding entries (shift 0, row 0, probability 0): clamp-free iff the tile is whole and their real
      // steps are -- the ballot above saw only the lanes with a real step
      if ((D % U) != 0 || DP > D) {
        if (lane == 0) {
          for (int t = D / U; t < DP / U; ++t) {
            bool ok = tile_whole;
            for (int j = t * U; j < D && ok; ++j)
              ok = ent[j].off8 - ent[j].lo8 + ic0 * 8 >= 0 && ent[j].off8 - ent[j].lo8 + (ic0 + TS - 1) * 8 <= nc18;
            flag[t] = ok;
          }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);
      }
    }
    // ---- the demand loop: serial in j, reference order (CashRecursion.java:113-122) ----
    double acc[S][W];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int w = 0; w < W; ++w) acc[s][w] = 0.0;
    if constexpr (LAST) {
      for (int j = 0; j < D; ++j) {
        const double t1 = ent[j].t1;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int w = 0; w < W; ++w) acc[s][w] += t1;
      }
    } else {
      for (int jb = 0; jb < DP; jb += U) {
        if constexpr (W == 2) {
          if (__builtin_amdgcn_readfirstlane(flag[jb / U])) {  // clamp-free trip: address = lane's base + shift
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const ShiftEntry e = ent[jb + u];
              const ShiftEntry2 e2 = ent2[jb + u];
#pragma unroll
              for (int s = 0; s < S; ++s) {
                const dpair_u v = *reinterpret_cast<const dpair_u*>(vbase + (uint32_t)(ic8[s] + e.off8));
                acc[s][0] += e.t1;
                acc[s][0] += e2.pg * v.x;
                acc[s][1] += e.t1;
                acc[s][1] += e2.pg * v.y;
              }
            }
            continue;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const ShiftEntry e = ent[jb + u];
          const ShiftEntry2 e2 = ent2[jb + u];
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const int i8 = ic8[s] + e.off8;  // 8 * (rowoff + ic + delta), before the clamp to the row
            if constexpr (W == 1) {
              const int t8 = med3_i32(i8, e.lo8, e2.hi8);  // 8 * (rowoff + clamp(ic + delta, 0, nc - 1))
              acc[s][0] += e.t1;
              acc[s][0] += e2.pg * *reinterpret_cast<const double*>(vbase + (uint32_t)t8);
            } else {
              // the pair {V[c], V[c + 1]}, c = clamp(ic + delta, 0, nc - 2); at the ends of the row both points of
              // the lane may clamp to the same entry
              const int c8 = med3_i32(i8, e.lo8, e2.hi8 - 8);
              const dpair_u v = *reinterpret_cast<const dpair_u*>(vbase + (uint32_t)c8);
              const double v0 = i8 > e2.hi8 - 8 ? v.y : v.x;  // ic + delta     >= nc - 1: the last entry
              const double v1 = i8 < e.lo8 ? v.x : v.y;       // ic + delta + 1 <= 0:      the first entry
              acc[s][0] += e.t1;
              acc[s][0] += e2.pg * v0;
              acc[s][1] += e.t1;
              acc[s][1] += e2.pg * v1;
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int w = 0; w < W; ++w)
        if (k < nA[s][w] && (MAXDIR ? (acc[s][w] > best[s][w]) : (acc[s][w] < best[s][w]))) {
          best[s][w] = acc[s][w];
          bestk[s][w] = k;
        }
  }

#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int w = 0; w < W; ++w) {
      s_val[wave * TS + 64 * W * s + W * lane + w] = best[s][w];
      s_k[wave * TS + 64 * W * s + W * lane + w] = bestk[s][w];
    }
  __syncthreads();
  for (int q = tid; q < TS; q += 256) {
    const int ic = ic0 + q;
    const int64_t idx = (int64_t)row * P.nc + ic;
    if (ic < P.nc && idx >= lo && idx < hi) {
      double bv = s_val[q];
      int bk = s_k[q];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        double ov = s_val[w * TS + q];
        int ok = s_k[w * TS + q];
        if (better<MAXDIR>(ov, ok, bv, bk)) {
          bv = ov;
          bk = ok;
        }
      }
      v_cur[idx] = bv;
      pol[idx] = bk;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// "Diagonal" form of the uniform-shift kernel (same family, same exactness argument; periods before T).
//
// What bounds cash_shift_kernel is the vector L1: every cell reads its 8 bytes of V_{t+1} through it (64 B/clk/CU,
// 94 % of that measured).  But the cells (action a, demand d) and (a + 1, d + 1) of one inventory row leave the SAME
// inventory y - d behind -- the same row of V_{t+1} -- and their cash shifts differ by the constant
// q (price - v): along a diagonal of the (action, demand) plane the reads of a tile slide over one row.  So a wave
// takes DR CONSECUTIVE ACTIONS at once, skewed by one demand step each (action k0 + i is at demand tau + i in step
// tau): the DR reads of a step fall into one row segment of 128 S + (shift spread) entries, which the wave stages in
// its own LDS region ONCE (clamped to the row's ends while staging, so that the cells need neither clamp nor select)
// and reads DR times: L1 traffic per cell falls by DR x 128 S / (128 S + spread), the per-cell read moves to the LDS
// (128 B/clk/CU, conflict-free: lane l owns the points c0 + 64 w + l).  The wave-uniform operands of a step --
// t1 = p_j * inc per action, the shifts, the row -- come from a table (DiagStep, 128 B per step) that a pre-pass
// builds per (row, action block) with the very operations of cash_shift_kernel's per-action setup, and reach the
// wave through the scalar cache: no LDS broadcasts, no per-action setup in the kernel.  Accumulation per cell is
// unchanged (acc += t1; acc += (p gamma) V, demand ascending); steps outside an action's demand range carry
// t1 = 0 and p = 0 and add exact zeros.  While the demand exceeds y (stock-out) neither row nor shifts change from
// step to step and the staged segment is kept.  The launcher takes this kernel only when the shifts of a block's cells
// in one step provably stay within DIAG_CAP points of each other -- q K + (DIAG_R - 1) q |price - v| + 2 <= DIAG_CAP: the
// slide along the diagonal, the action-0 break of a fixed cost, the rounding -- and cash_shift_kernel otherwise.
// ---------------------------------------------------------------------------------------------
constexpr int DIAG_R = 8;      // actions per block
constexpr int DIAG_CAP = 128;  // largest shift spread a staged segment covers
// One demand step of one action block: 160 bytes, read by the wave through the scalar cache one step ahead of its use.
struct alignas(16) DiagHalf {  // four actions of a step
  double t1[4];  // p_j * inc of (action k0 + i, demand tau + i); 0 outside the demand range
  double pg[4];  // p_j * gamma of the same cells; 0 outside the demand range
};
struct alignas(16) DiagHead {
  uint32_t rel[2];  // shift_i - dmin, one byte per action: where action i reads inside the staged segment
  int32_t nbase8;   // the segment of the step AFTER NEXT (segments are requested two steps ahead): 8 * next inventory index * nc,
  int32_t ndmin;    // and its smallest shift (cash grid points)
  int32_t base8;    // the same two of this step (the prologue stages steps 0 and 1 from them)
  int32_t dmin;
  int32_t pad[2];
};
struct alignas(16) DiagStep {
  DiagHalf h[2];  // actions 0..3, 4..7
  DiagHead hd;
};
static_assert(sizeof(DiagStep) == 160 && sizeof(DiagHalf) == 64 && sizeof(DiagHead) == 32, "DiagStep is one 160-byte record");

struct DiagParams {
  CashShiftParams C;
  int32_t n_blocks;  // action blocks per row: ceil(n_actions_cap / DIAG_R)
  int32_t n_steps;   // D + DIAG_R - 1 rounded up to an even number (the kernel's step loop is unrolled by two)
  int32_t n_rows;    // inventory rows launched
  int32_t cap;       // largest shift spread of a step the launch stages for (64 or DIAG_CAP: one 64-entry piece less)
  int32_t band_tiles;  // > 0: XCD i owns the cash band of tiles [i * band_tiles, (i + 1) * band_tiles) of EVERY row (see the kernel)
  int32_t pad0;
};

// shift, row and t1 of one (action, demand): the operations of cash_shift_kernel's per-action setup, in its order
__device__ __forceinline__ void diag_cell(const CashShiftParams& P, double x, int k, double d, double p, double& t1,
                                          int& delta, int& base8) {
  const double a = (double)k * P.step;
  const double y = x + a;
  const double fixed = a > 0 ? P.K : 0.0;
  const double var = P.v * a;
  const double level = y - d;
  const double pos = jmax(level, 0.0);
  const double inc = P.price * jmin(y, d) - fixed - var - P.h * pos - P.overhead;
  t1 = p * inc;
  double ninv = jmax(0.0, level);
  ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
  ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
  base8 = (int)((ninv - P.next_x_lo) / P.step) * P.nc * 8;
  delta = (int)jmax(jmin(jround_d(inc * P.q), (double)P.nc), -(double)P.nc);
}

// geometry of one step: row, smallest shift, spread (and, when `e` is given, the record but its look-ahead fields).  Steps
// behind the last real one (the even padding, the look-ahead of the last steps) have no valid action: row 0, shift 0.
__device__ __forceinline__ void diag_step(const DiagParams& Q, double x, int k0, int t, const double* __restrict__ pmf_d,
                                          const double* __restrict__ pmf_p, int& base8, int& dmin, int& spread, DiagStep* e) {
  const int D = Q.C.n_demand;
  int delta[DIAG_R];
  bool valid[DIAG_R];
  int lo = 0x7fffffff, hi = -0x7fffffff;
  base8 = 0;
#pragma unroll
  for (int i = 0; i < DIAG_R; ++i) {
    const int j = t - (DIAG_R - 1) + i;
    valid[i] = j >= 0 && j < D;
    double t1 = 0.0;
    delta[i] = 0;
    if (valid[i]) {
      int b8;
      diag_cell(Q.C, x, k0 + i, pmf_d[j], pmf_p[j], t1, delta[i], b8);
      lo = delta[i] < lo ? delta[i] : lo;
      hi = delta[i] > hi ? delta[i] : hi;
      base8 = b8;  // (every valid action of a step leaves the same inventory behind: same row)
    }
    if (e) {
      e->h[i >> 2].t1[i & 3] = t1;
      e->h[i >> 2].pg[i & 3] = valid[i] ? pmf_p[j] * Q.C.gamma : 0.0;
    }
  }
  if (hi < lo) lo = hi = 0;
  dmin = lo;
  spread = hi - lo;
  if (e) {
    e->hd.rel[0] = e->hd.rel[1] = 0;
    if (spread <= DIAG_CAP) {
#pragma unroll
      for (int i = 0; i < DIAG_R; ++i)
        if (valid[i]) e->hd.rel[i >> 2] |= (uint32_t)(delta[i] - lo) << (8 * (i & 3));
    }
    e->hd.base8 = base8;
    e->hd.dmin = dmin;
  }
}

// pre-pass: one thread per (row, action block, step)
// bounds[2 b], bounds[2 b + 1]: DIAG_BIAS + largest dmin and DIAG_BIAS - smallest dmin over the steps of block b (atomicMax
// on zero-filled words): is every segment of the block inside the row?
constexpr int DIAG_BIAS = 1 << 30;
__global__ __launch_bounds__(256) void cash_diag_table_kernel(DiagParams Q, int n_rows, DiagStep* __restrict__ table,
                                                              const double* __restrict__ pmf_d,
                                                              const double* __restrict__ pmf_p, int* __restrict__ overflow,
                                                              int* __restrict__ bounds) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)n_rows * Q.n_blocks * Q.n_steps;
  if (g >= total) return;
  const int t = (int)(g % Q.n_steps);
  const int64_t rb = g / Q.n_steps;
  const int kb = (int)(rb % Q.n_blocks);
  const int row = Q.C.row0 + (int)(rb / Q.n_blocks);
  const double x = Q.C.x_lo + (double)row * Q.C.step;
  const int k0 = kb * DIAG_R;
  DiagStep e;
  int base8, dmin, spread;
  diag_step(Q, x, k0, t, pmf_d, pmf_p, base8, dmin, spread, &e);
  // (the launcher only takes this kernel when q K + (DIAG_R - 1) q |price - v| (+ 2 unless every shift is a whole number of
  // grid steps before rounding) <= Q.cap, which bounds every spread:
  // shifts held at +-nc only move closer together.  The word is the guard behind that argument, read by the launcher's
  // SDPGPU_CASH_DIAG_CHECK mode and by the tests.)
  if (spread > Q.cap) atomicOr(overflow, 1);
  int b2, s2;
  diag_step(Q, x, k0, t + 2, pmf_d, pmf_p, b2, e.hd.ndmin, s2, nullptr);
  e.hd.nbase8 = b2;
  e.hd.pad[0] = e.hd.pad[1] = 0;
  // (the look-ahead steps behind the last one have dmin 0)
  const int dhi = t + 2 >= Q.n_steps ? (dmin > 0 ? dmin : 0) : dmin, dlo = t + 2 >= Q.n_steps ? (dmin < 0 ? dmin : 0) : dmin;
  atomicMax(bounds + 2 * rb, DIAG_BIAS + dhi);
  atomicMax(bounds + 2 * rb + 1, DIAG_BIAS - dlo);
  table[g] = e;
}

// The step loop of one action block.  INTERIOR: every segment of the block lies inside the row (decided per wave and block
// from the block's smallest and largest shift): a piece is read at scalar base + lane, without the clamp.
// Software pipeline, one basic block per two steps.  LDS reads and scalar loads share one counter (lgkmcnt) and scalar loads
// return out of order, so every wait for LDS data also drains the scalar loads in flight: the waits sit at the TOP of a
// phase, ahead of the new requests, and every request (the other half's LDS reads, the next record's scalar loads) has one
// half step of arithmetic to arrive.  The segment of step t + 2 is requested (vector loads, their own counter) during step
// t into one of two register sets and stored behind the first half of step t + 1.
template <int S, bool INTERIOR, int NU>
__device__ __forceinline__ void diag_block(const DiagStep* __restrict__ tab, int n_steps, const char* vbase, char* my_lds,
                                           int lane8, int lane, int ic0, int nc1, double (&acc)[DIAG_R][2 * S]) {
  constexpr int NP = 2 * S;
  constexpr int TS = 64 * NP;
  constexpr int SEG = TS + DIAG_CAP;  // (buffer stride; NU pieces of 64 entries are staged: spreads up to 64 NU - TS)
  auto seg_load = [&](int base8, int dmin, double (&tmp)[NU]) {
    const int first = ic0 + dmin;
    if constexpr (INTERIOR) {
      const char* src = vbase + ((int64_t)base8 + (int64_t)first * 8);
#pragma unroll
      for (int u = 0; u < NU; ++u) tmp[u] = *reinterpret_cast<const double*>(src + (uint32_t)(lane8 + u * 512));
    } else {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int c = med3_i32(first + u * 64 + lane, 0, nc1);
        tmp[u] = *reinterpret_cast<const double*>(vbase + (uint32_t)(base8 + c * 8));
      }
    }
  };
  auto seg_store = [&](const double (&tmp)[NU], int buf) {
#pragma unroll
    for (int u = 0; u < NU; ++u) *reinterpret_cast<double*>(my_lds + (uint32_t)(buf + u * 512 + lane8)) = tmp[u];
  };
  // (the compiler pairs the two points of an action into one ds_read2st64_b64, 8 LDS-array cycles per wave; two ds_read_b64
  // written out in assembly -- 2 + 2 cycles by MI355X_MICROARCH.md's table -- measured SLOWER here, 41.8 against 38.9 ms per
  // sweep: twice the DS instructions to issue and address)
  auto lds_reads = [&](double (&v)[4][NP], uint32_t rel, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rel8 = (int)((rel >> (8 * i)) & 0xffu) * 8 + buf;
      const double* src = reinterpret_cast<const double*>(my_lds + (uint32_t)(lane8 + rel8));
#pragma unroll
      for (int w = 0; w < NP; ++w) v[i][w] = src[64 * w];
    }
  };
  auto half_step = [&](const DiagHalf& hh, const double (&v)[4][NP], int i0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double t1 = hh.t1[i];
      const double pgi = hh.pg[i];
#pragma unroll
      for (int w = 0; w < NP; ++w) {
        acc[i0 + i][w] += t1;
        acc[i0 + i][w] += pgi * v[i][w];
      }
    }
  };
  // prologue: segments of steps 0 and 1 (step 0 into buffer 0 at once), records of step 0, first reads
  double tmp_a[NU], tmp_b[NU];
  uint32_t rel0, rel1;
  int nb8, ndm;
  {
    const DiagHead h0 = tab[0].hd;
    const DiagHead h1 = tab[1].hd;
    seg_load(h0.base8, h0.dmin, tmp_b);
    seg_load(h1.base8, h1.dmin, tmp_a);
    seg_store(tmp_b, 0);
    rel0 = h0.rel[0];
    rel1 = h0.rel[1];
    nb8 = h0.nbase8;
    ndm = h0.ndmin;
  }
  DiagHalf ha = tab[0].h[0];
  DiagHalf hb = tab[0].h[1];
  double va[4][NP], vb[4][NP];
  __builtin_amdgcn_wave_barrier();
  lds_reads(va, rel0, 0);
  // one step: reads buffer `cur`, stores `tst` (the segment of step t + 1) into the other buffer, loads the segment of step
  // t + 2 into `tld`
  auto step = [&](int t, int cur, double (&tst)[NU], double (&tld)[NU]) {
    const int tn = t + 1 < n_steps ? t + 1 : t;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): va (requested before the previous half's arithmetic)
    __builtin_amdgcn_sched_barrier(0);
    lds_reads(vb, rel1, cur);
    const uint32_t nrel0 = tab[tn].hd.rel[0], nrel1 = tab[tn].hd.rel[1];
    const int nnb8 = tab[tn].hd.nbase8, nndm = tab[tn].hd.ndmin;
    const DiagHalf han = tab[tn].h[0];
    seg_load(nb8, ndm, tld);
    __builtin_amdgcn_sched_barrier(0);
    half_step(ha, va, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // vb, the next head and first half
    __builtin_amdgcn_sched_barrier(0);
    // (pin: what is derived from the record just waited for is formed here, not ahead of the wait)
    uint32_t nrel0p = nrel0;
    asm volatile("" : "+s"(nrel0p));
    seg_store(tst, cur ^ (SEG * 8));
    lds_reads(va, nrel0p, cur ^ (SEG * 8));
    const DiagHalf hbn = tab[tn].h[1];
    __builtin_amdgcn_sched_barrier(0);
    half_step(hb, vb, 4);
    __builtin_amdgcn_sched_barrier(0);
    rel0 = nrel0;
    rel1 = nrel1;
    nb8 = nnb8;
    ndm = nndm;
    ha = han;
    hb = hbn;
  };
  for (int t = 0; t < n_steps; t += 2) {
    step(t, 0, tmp_a, tmp_b);
    step(t + 1, SEG * 8, tmp_b, tmp_a);
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
}

template <bool MAXDIR, int S, int NU>
__global__ __launch_bounds__(256) void cash_diag_kernel(DiagParams Q, const DiagStep* __restrict__ table,
                                                        const int* __restrict__ bounds,
                                                        const double* __restrict__ v_next, double* __restrict__ v_cur,
                                                        int32_t* __restrict__ pol, int64_t lo, int64_t hi) {
  constexpr int R = DIAG_R;
  constexpr int NP = 2 * S;              // points per lane: c0 + 64 w + lane
  constexpr int TS = 64 * NP;            // points per tile
  constexpr int SEG = TS + DIAG_CAP;     // entries of one staged segment
  const CashShiftParams& P = Q.C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [4 waves][2 buffers][SEG] doubles, then the read-out scratch
  double* s_val = reinterpret_cast<double*>(smem + (size_t)4 * 2 * SEG * 8);
  int* s_k = reinterpret_cast<int*>(s_val + 4 * TS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Block -> (row, tile): XCD i (blocks b with b % 8 == i, the dispatcher's round-robin) takes the rows i, i + 8, ... one after
  // the other, all tiles of a row in a run.  The workgroups of one row stream the same DiagStep records: kept on one XCD
  // and close in time they find them in its L2 (numbered row-major, the tiles of a row sat on eight XCDs and ~25 rows were in
  // flight: 9 GB per launch came through the fabric and every scalar load waited for it).  Placement only.
  // Q.band_tiles > 0 (round 3): XCD i owns a BAND of the cash axis for every row and walks the rows in order.  Its working set
  // of V_{t+1} is then one band (plus the shifts' reach) of every row -- configs[2]: 200 rows x ~1000 points = 1.6 MB, inside
  // its 4 MB L2 -- where "rows i, i + 8, ..." makes every XCD read the whole 8 MB table per row it walks; the price is that
  // every XCD streams all rows' DiagStep records instead of an eighth of them.
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  int rowi, tile;
  if (Q.band_tiles > 0) {
    rowi = seq / Q.band_tiles;
    tile = xcd * Q.band_tiles + (seq - rowi * Q.band_tiles);
    if (tile >= P.tiles_per_row) return;
  } else {
    rowi = (seq / P.tiles_per_row) * 8 + xcd;
    tile = seq % P.tiles_per_row;
  }
  if (rowi >= Q.n_rows) return;
  const int row = P.row0 + rowi;
  const int ic0 = tile * TS;
  const int nc1 = P.nc - 1;

  // feasible action count per point (CashConstraint.java:96-99) and the wave's maximum
  int nA[NP];
  int nA_max = 0;
#pragma unroll
  for (int w = 0; w < NP; ++w) {
    const int ic = ic0 + 64 * w + lane;
    const int ic_c = ic < P.nc ? ic : nc1;
    const double cash = (double)(P.k_lo + ic_c) / P.q;  // exact: q is a power of two
    double m = jmin(P.max_order_quantity, jmax(0.0, (cash - P.overhead - P.K) / P.v));
    nA[w] = ((m != m) ? 0 : (int)m) + 1;
    nA_max = nA[w] > nA_max ? nA[w] : nA_max;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_xor(nA_max, off, 64);
    nA_max = o > nA_max ? o : nA_max;
  }
  nA_max = __builtin_amdgcn_readfirstlane(nA_max);

  double best[NP];
  int bestk[NP];
#pragma unroll
  for (int w = 0; w < NP; ++w) {
    best[w] = MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;
    bestk[w] = 0;
  }
  const char* vbase = reinterpret_cast<const char*>(v_next);
  char* my_lds = smem + (size_t)wave * 2 * SEG * 8;
  const int lane8 = lane * 8;

  for (int kb = wave; kb * R < nA_max; kb += 4) {
    const DiagStep* tab = table + ((size_t)rowi * Q.n_blocks + kb) * Q.n_steps;
    double acc[R][NP];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int w = 0; w < NP; ++w) acc[i][w] = 0.0;
    const int bmax = bounds[2 * ((size_t)rowi * Q.n_blocks + kb)] - DIAG_BIAS;
    const int bmin = DIAG_BIAS - bounds[2 * ((size_t)rowi * Q.n_blocks + kb) + 1];
    if (ic0 + bmin >= 0 && ic0 + bmax + SEG - 1 <= nc1)
      diag_block<S, true, NU>(tab, Q.n_steps, vbase, my_lds, lane8, lane, ic0, nc1, acc);
    else
      diag_block<S, false, NU>(tab, Q.n_steps, vbase, my_lds, lane8, lane, ic0, nc1, acc);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int k = kb * R + i;
#pragma unroll
      for (int w = 0; w < NP; ++w)
        if (k < nA[w] && (MAXDIR ? (acc[i][w] > best[w]) : (acc[i][w] < best[w]))) {
          best[w] = acc[i][w];
          bestk[w] = k;
        }
    }
  }

#pragma unroll
  for (int w = 0; w < NP; ++w) {
    s_val[wave * TS + 64 * w + lane] = best[w];
    s_k[wave * TS + 64 * w + lane] = bestk[w];
  }
  __syncthreads();
  for (int q = tid; q < TS; q += 256) {
    const int ic = ic0 + q;
    const int64_t idx = (int64_t)row * P.nc + ic;
    if (ic < P.nc && idx >= lo && idx < hi) {
      double bv = s_val[q];
      int bk = s_k[q];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        double ov = s_val[w * TS + q];
        int ok = s_k[w * TS + q];
        if (better<MAXDIR>(ov, ok, bv, bk)) {
          bv = ov;
          bk = ok;
        }
      }
      v_cur[idx] = bv;
      pol[idx] = bk;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// "Cash row" period kernel: the cash families F3 (both formulas), F4, F5, F6 on ANY cash grid
// (tenths, hundredths, non-dyadic prices: the cases the uniform-shift kernel must refuse).
//
// The cash axis is the fastest one, so the 64 lanes of a wave are 64 consecutive cash points of ONE
// (inventory[, preQ]) row.  For such a wave everything in the reference's lambdas that does not read
// the cash balance -- revenue, level, holding cost, salvage, the next inventory row -- is WAVE-UNIFORM per
// (action, demand): lanes = demand indices compute it once per action into LDS, and the per-cell work is
// only the cash-dependent tail (4-5 additions, the clamp, Math.round, one gather).  Every operation, its
// operands and its order are those of cell<FAM>() in sdp_device.hpp -- hoisting an operation out of a loop
// does not change its result -- so the tables are bit-identical to the generic kernel's
// (tests run both).  Two operations are elided where they provably change nothing: `inc += 0.0` (salvage
// outside period T) and `inc += 0 * endCash` (penalty rate zero): either can only turn a -0.0 into +0.0,
// which no later operation (p * inc added to an accumulator that is never -0.0; cash + inc) can see.
// ---------------------------------------------------------------------------------------------
// Cash index of a next-period balance: CashConstraint.java:126-131 (clamp, Math.round quantiser).  Same result as
// cash_index() in sdp_device.hpp with fewer instructions: the two clamp ternaries become v_min/v_max (they differ
// from the ternaries only in the sign of a zero, which Math.round maps to the same key) and the tie rule of
// Math.round is folded into the integer add.
// The clamp moves behind the quantiser: q(x) = Math.round(x * mult) [/ div] is monotone non-decreasing and the grid's end
// keys are q(minCash), q(maxCash) (cash_key_of_bound on the host), so q(clamp(x, minCash, maxCash)) ==
// clamp(q(x), q(minCash), q(maxCash)) for every x -- one v_med3_i32 instead of v_min_f64 + v_max_f64.  The launcher
// only takes this kernel when |x * mult| stays far below 2^31 (cash_row_eligible), so the conversion never saturates.
// Math.round of the quantiser (CashConstraint.java:131 `Math.round(cash * 10)`), N values at once, as ints -- TWO additions per
// value instead of floor / subtract / compare / convert / add-with-carry.  Math.round(x) = floor(x + 1/2) in exact arithmetic
// (ties toward +infinity).  With the fp64 rounding mode set to ROUND TOWARD MINUS INFINITY for these additions only:
//   t  = fl_down(x + 0.5)    lies in [floor(x + 0.5), x + 0.5]: floor(x + 0.5) is representable and not above the exact sum, and
//                            fl_down returns the largest double not above it -- so floor(t) == floor(x + 0.5), exactly, for every x
//                            (round-to-nearest would turn 0.49999999999999994 + 0.5 into 1.0: the classic wrong floor(x + 0.5));
//   t2 = fl_down(t + 1.5 * 2^52) = 1.5 * 2^52 + floor(t), whose low mantissa word IS the integer (two's complement), |x| < 2^31
//                            (the launchers admit |x| < 5e8: cash_row_eligible).
// The mode is switched and restored INSIDE one asm block (s_setreg_imm32_b32 on the double-precision round field of MODE, the
// instruction the compiler's own mode-register pass emits), so no other fp64 operation of the wave can be scheduled into the
// window: every accumulation keeps round-to-nearest-even.  Same result as cash_key_row's floor form on every input, bit for bit.
#define SDP_RTN_ON "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
#define SDP_RTN_OFF "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
constexpr double kRoundMagic = 6755399441055744.0;  // 1.5 * 2^52
__device__ __forceinline__ int jround_rtn(double x) {
  double t;
  asm volatile(SDP_RTN_ON "v_add_f64 %0, %1, 0.5\n\tv_add_f64 %0, %0, %2\n\t" SDP_RTN_OFF : "=&v"(t) : "v"(x), "s"(kRoundMagic));
  return __double2loint(t);
}
__device__ __forceinline__ void jround_rtn2(const double (&x)[2], int (&k)[2]) {
  double t0, t1;
  asm volatile(SDP_RTN_ON
               "v_add_f64 %0, %2, 0.5\n\tv_add_f64 %1, %3, 0.5\n\t"
               "v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\t" SDP_RTN_OFF
               : "=&v"(t0), "=&v"(t1)
               : "v"(x[0]), "v"(x[1]), "s"(kRoundMagic));
  k[0] = __double2loint(t0);
  k[1] = __double2loint(t1);
}
__device__ __forceinline__ void jround_rtn4(const double (&x)[4], int (&k)[4]) {
  double t0, t1, t2, t3;
  asm volatile(SDP_RTN_ON
               "v_add_f64 %0, %4, 0.5\n\tv_add_f64 %1, %5, 0.5\n\tv_add_f64 %2, %6, 0.5\n\tv_add_f64 %3, %7, 0.5\n\t"
               "v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t" SDP_RTN_OFF
               : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "s"(kRoundMagic));
  k[0] = __double2loint(t0);
  k[1] = __double2loint(t1);
  k[2] = __double2loint(t2);
  k[3] = __double2loint(t3);
}
__device__ __forceinline__ void jround_rtn8(const double (&x)[8], int (&k)[8]) {
  double t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(SDP_RTN_ON
               "v_add_f64 %0, %8, 0.5\n\tv_add_f64 %1, %9, 0.5\n\tv_add_f64 %2, %10, 0.5\n\tv_add_f64 %3, %11, 0.5\n\t"
               "v_add_f64 %4, %12, 0.5\n\tv_add_f64 %5, %13, 0.5\n\tv_add_f64 %6, %14, 0.5\n\tv_add_f64 %7, %15, 0.5\n\t"
               "v_add_f64 %0, %0, %16\n\tv_add_f64 %1, %1, %16\n\tv_add_f64 %2, %2, %16\n\tv_add_f64 %3, %3, %16\n\t"
               "v_add_f64 %4, %4, %16\n\tv_add_f64 %5, %5, %16\n\tv_add_f64 %6, %6, %16\n\tv_add_f64 %7, %7, %16\n\t" SDP_RTN_OFF
               : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "s"(kRoundMagic));
  k[0] = __double2loint(t0);
  k[1] = __double2loint(t1);
  k[2] = __double2loint(t2);
  k[3] = __double2loint(t3);
  k[4] = __double2loint(t4);
  k[5] = __double2loint(t5);
  k[6] = __double2loint(t6);
  k[7] = __double2loint(t7);
}

template <bool INTDIV>
__device__ __forceinline__ int cash_key_row(double next_cash, int key_lo, int key_hi, double round_mult, double round_div) {
  const double xm = next_cash * round_mult;
#ifdef SDP_ROUND_FLOOR  // (A/B builds only: the floor form of rounds 1-3)
  const double f = floor(xm);
  int k = (int)f + ((xm - f) >= 0.5 ? 1 : 0);
#else
  int k = jround_rtn(xm);
#endif
  if constexpr (INTDIV) k = (int)trunc((double)k / round_div);  // `/ 10`: long division (CashOverdraft.java:116)
  return med3_i32(k, key_lo, key_hi);
}

struct alignas(16) RowEnt {
  double u;       // F3 formula 0: (1 - overheadRate) * revenue; formula 1: the whole increment; F4/F5/F6: revenue
  double hold;    // holdingCost * max(level, 0)
  double sal;     // salvageValue * max(level, 0) (period T)
  int32_t rowoff8; // BYTE offset of the next state's inventory row less the grid's first key: 8 * (inv_index * nc - k_lo),
                   // so that a cell's gather address is v_next + (rowoff8 + 8 * key): one v_lshl_add_u32, scalar base
  int32_t dkey;    // uniform key shift of this (action, demand), see "uniform-key trips" below; kNoShift: none
};
constexpr int32_t kNoShift = INT32_MIN;

// FORMULA1: F3 with CashConstraintTesting.java's increment (no cash term); INTDIV: the quantiser divides by an
// integer other than 1; PEN: F3 with a non-zero end-cash penalty rate (CashConstraint.java:115-118); LEAN: F3 formula
// 0 / 2 with holdingCost == +0.0 and overheadCost == +0.0 (CashConstraint.main, CashConstraintXR.main): the two
// subtractions `- holdCosts - overheadCost` of the increment subtract +0.0, the identity on every double, and are skipped.
//
// Block -> tile map (RowTiling): the V_{t+1} table of these grids (1e7 states = 80 MB) is far larger than an XCD's
// 4 MB L2, and a tile's cells read, per next-inventory row, a window of a few thousand cash points around its own
// position.  With blocks numbered row-major the ~600 workgroups in flight span the whole cash axis of two rows, their
// union of windows is the whole of ~125 rows (20 MB), and three L2 requests in four miss (rocprofv3: TCC_MISS 1.35e9 of
// 1.79e9, 172 GB per launch through the fabric at 8 TB/s = the kernel's time).  So the cash axis is cut into 8 * nsub
// bands of `tps` tiles: XCD i (blocks b with b % 8 == i, the dispatcher's round-robin) owns bands i*nsub .. , and walks
// band by band, inside a band row by row: what is in flight on one XCD is a few rows of ONE narrow band, whose
// windows (~125 rows x ~20 KB) fit its L2.  Placement only: any map is correct.
struct RowTiling {
  int32_t tiles_per_row;
  int32_t n_rows;  // inventory(/preQ) rows launched
  int32_t tps;     // tiles per band; 0 = plain row-major numbering (short rows)
  int32_t nsub;    // bands per XCD
  // 1: XCD i walks the bands i, i + 8, i + 16, ... instead of the contiguous run i * nsub ...: the cost of a tile varies ALONG the
  // cash axis (action counts grow with the balance, trips clamp near the ends, balances below zero pay interest), a contiguous
  // eighth per XCD hands the dear end to one XCD and the launch lasts as long as its slowest XCD
  int32_t band_interleave;
  // colmajor > 0 (the shared-block form, whose workgroup tiles are few and wide -- 20 per row on CashConstraint.main's grid, which
  // whole-column bands would deal 3, 3, 3, 3, 3, 3, 2, 0 to the eight XCDs): the (tile, row) units in column-major order,
  // unit u = tile * n_rows + row, are cut into eight equal runs of `colmajor` units, XCD i walks run i.  Still one narrow band
  // of consecutive rows in flight per XCD, and every XCD gets the same number of workgroups.
  int32_t colmajor;
  int32_t slots;  // cash_row_pair_kernel, in-kernel setup: entry slots per wave (cash_row_slots(D); 1 = one action per setup pass)
  int32_t rows_real;  // cash_row_pair_kernel with four rows per workgroup (RW = 4): n_rows counts groups of four, this the rows
  // F5, DIAGONAL ORDER (round 4; nullptr: the band numbering above).  A cell of level y = x + preQ at demand d reads row y - d of
  // plane a at the keys  c + mult * (price * min(y, d) - v a)  (balances that pay no interest): a (level, tile) unit touches
  // 31 planes x 40 rows x its window = 3 MB of V_{t+1} that NO other level's unit of the same tile touches -- rocprofv3 on
  // SingleProductLeadtime's size: 34 GB per launch from beyond the L2, every trip waiting for one of those misses (VALU 58 %,
  // TA 62 % busy).  Units on one DIAGONAL  tile_cash + price * mult * level = const  read, per (plane, row), the SAME window
  // whatever their level: the (row group, tile) units are walked in order of that diagonal, an eighth of the order per XCD.
  // units[u] = {row group (or row), tile}; the order is dealt to the XCDs in segments of units_seg units.  Placement only.
  const int2* units;
  int32_t units_seg, units_total;
  // Row order inside a band (nullptr: as numbered).  F5's state is (x, preQ) but its cells read V_{t+1} through the level
  // y = x + preQ only (SingleProductLeadtime.java:82-119): rows with equal y gather the very same entries.  Walked in order of y
  // they are in flight together and find each other's lines in L2; in (preQ, x) order the 31 rows of a level are 61 rows apart
  // and every gather went to the fabric (rocprofv3 on SingleProductLeadtime's size: 59 % L2 misses, 547 GB per launch = 5.7 TB/s,
  // which was the kernel's time).  Placement only.
  const int32_t* perm;
};

// UNI: uniform-key trips are compiled in (F3 without deposit rate, penalty and integer division -- CashConstraint.main,
// CashConstraintTesting, CashConstraintXR.main).  With a zero deposit rate the cash balance cancels out of the increment in
// REAL arithmetic: nextCash = cash + (u - fixed - var - hold - overhead) for every cash point.  If that real number times
// the quantiser's factor is an integer delta (prices and costs on the cash grid, as in those drivers), then
// Math.round(nextCash_fp * mult) -- the reference's rounded floating-point chain -- can only be key(cash) + delta: the
// chain's error is below 2^-16 for the magnitudes the launcher admits (cash_row_eligible), and Math.round maps anything
// within +-0.5 of an integer to it.  The per-action setup checks it per (action, demand) (|mult * inc - delta| < 2^-20,
// formed from the same doubles), the wave ANDs the four steps of a trip, and such a trip replaces the eight instructions
// of the quantiser (add, mul, floor, sub, cmp, cvt, addc, clamp) by an integer add and the clamp.  The increment itself
// -- what enters the accumulator -- is still formed per lane exactly as the reference forms it.
template <int FAM, bool LAST, bool FORMULA1, bool INTDIV, bool PEN = false, bool LEAN = false, bool UNI = false>
__global__ __launch_bounds__(256) void cash_row_kernel(DevParams P, const double* __restrict__ v_next,
                                                       double* __restrict__ v_cur, int32_t* __restrict__ pol,
                                                       const double* __restrict__ pmf_d,
                                                       const double* __restrict__ pmf_p, int64_t lo, int64_t hi,
                                                       int64_t row0, RowTiling G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int D = P.n_demand;
  double2* s_p = reinterpret_cast<double2*>(smem);                  // {p_j, p_j * gamma}
  RowEnt* s_ent = reinterpret_cast<RowEnt*>(s_p + D);               // [4 waves][D]
  double* s_d = reinterpret_cast<double*>(s_ent + (size_t)4 * D);   // d_j
  double* s_val = s_d + D;
  int* s_k = reinterpret_cast<int*>(s_val + 4 * 64);
  int* s_uni = s_k + 4 * 64;  // [4 waves][(D + 3) / 4]: per trip of four steps, "all four have a uniform key shift"

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int row_i, tile;
  if (G.tps == 0) {
    row_i = blockIdx.x / G.tiles_per_row;
    tile = blockIdx.x % G.tiles_per_row;
  } else {
    const int xcd = blockIdx.x & 7, n = blockIdx.x >> 3;
    const int per_band = G.n_rows * G.tps;
    const int sub = n / per_band, rem = n - sub * per_band;
    row_i = rem / G.tps;
    tile = (G.band_interleave ? sub * 8 + xcd : xcd * G.nsub + sub) * G.tps + (rem - row_i * G.tps);
    if (tile >= G.tiles_per_row) return;  // the bands cover a little more than the row (whole workgroup leaves)
  }
  for (int j = tid; j < D; j += 256) {
    s_d[j] = pmf_d[j];
    s_p[j] = make_double2(pmf_p[j], pmf_p[j] * P.gamma);
  }
  __syncthreads();

  const int64_t row = row0 + (G.perm ? G.perm[row_i] : row_i);  // (iq * nx + ix)
  const int ic0 = tile * 64;
  const int nc = (int)P.cur.nc;
  const int ic = ic0 + lane;
  const int ic_c = ic < nc ? ic : nc - 1;
  const int64_t idx = row * nc + ic;
  const bool live = ic < nc && idx >= lo && idx < hi;

  StateT s;
  decode_state_row<FAM>(P, row, ic_c, s);  // x (and preQ) are the same in every lane
  const double base = (FAM == FAM_CASH_LEADTIME) ? s.x + s.preq : s.x;  // + action below for F3/F4/F6

  // feasible actions: per lane, and the tile's maximum (non-decreasing in cash where it varies at all)
  const int nA = n_actions<FAM>(P, s);
  int nA_max = nA;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_xor(nA_max, off, 64);
    nA_max = o > nA_max ? o : nA_max;
  }
  nA_max = __builtin_amdgcn_readfirstlane(nA_max);

  RowEnt* ent = s_ent + (size_t)wave * D;
  const bool MAXDIR = P.maxdir != 0;
  double best = MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;
  int bestk = 0;
  // loop invariants of the cash quantiser, in registers
  const double round_mult = P.round_mult, round_div = P.round_div;
  const double overhead = P.overhead;
  const int k_lo_next = (int)P.next.k_lo;
  // keys of minCash / maxCash, parked in VGPRs once (v_med3_i32 below takes vector operands; left to itself the
  // compiler keeps the two wave-uniform values in SGPRs and copies them over in every cell)
  int key_lo_v, key_hi_v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(key_lo_v) : "s"(k_lo_next));
  asm volatile("v_mov_b32 %0, %1" : "=v"(key_hi_v) : "s"(k_lo_next + (int)P.next.nc - 1));
  for (int k = wave; k < nA_max; k += 4) {
    // ---- wave-uniform part, lanes = demand indices --------------------------------------------
    const double a = (double)k * P.step;
    const double y = (FAM == FAM_CASH_LEADTIME) ? base : base + a;  // level before demand
    const double fixed = a > 0 ? P.K : 0.0;
    const double var = P.v * a;
    for (int j = lane; j < D; j += 64) {
      const double d = s_d[j];
      const double revenue = P.price * jmin(y, d);
      const double level = y - d;
      const double pos = jmax(level, 0.0);
      RowEnt e;
      e.hold = P.h * pos;
      e.sal = LAST ? P.salvage * pos : 0.0;
      if constexpr (FAM == FAM_CASH) {
        if constexpr (!FORMULA1) {
          e.u = P.one_minus_overhead_rate * revenue;
        } else {  // CashConstraintTesting.java:117-132: nothing in the increment reads the cash balance
          double inc = revenue - fixed - var - e.hold - P.overhead;
          inc += e.sal;  // (0.0 outside period T, exactly what the reference adds)
          e.u = inc;
        }
      } else {
        e.u = revenue;
      }
      e.rowoff8 = 0;
      e.dkey = kNoShift;
      if constexpr (!LAST) {
        double ninv = jmax(0.0, level);
        ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
        ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
        e.rowoff8 = (inv_index(P, ninv) * (int)P.next.nc - k_lo_next) * 8;  // + 8 * cash key = byte offset of the state
        if constexpr (UNI) {
          // the increment with the cash balance cancelled (real arithmetic), times the quantiser's factor
          const double inc_u = FORMULA1 ? e.u : (LEAN ? e.u - fixed - var : e.u - fixed - var - e.hold - overhead);
          const double dm = inc_u * round_mult;
          const double dn = rint(dm);
          const bool uni = fabs(dm - dn) < 9.5367431640625e-07 && fabs(dn) < 1.0e9;  // 2^-20
          e.dkey = uni ? (int)dn : kNoShift;
          const unsigned long long m4 = __ballot(uni);
          if ((lane & 3) == 0) s_uni[wave * ((D + 3) / 4) + j / 4] = ((m4 >> lane) & 15ull) == 15ull;
        }
      }
      ent[j] = e;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed

    // ---- per-lane (cash) invariants of this action ------------------------------------------------
    double dep_or_bi = 0.0;  // F3/F6: deposit; F4/F5: cashBalanceBefore - interest
    if constexpr (FAM == FAM_CASH || FAM == FAM_SURVIVAL) {
      dep_or_bi = (s.cash - fixed - var) * P.one_plus_deposit;
    } else if constexpr (FAM == FAM_OVERDRAFT) {
      const double before = s.cash - fixed - var - P.overhead;
      dep_or_bi = before - overdraft_interest(P, before);
    } else {  // FAM_CASH_LEADTIME
      const double before = s.cash - var - P.overhead;
      dep_or_bi = before - overdraft_interest(P, before);
    }
    // the plane of the next pipeline quantity (F5: next preQ = action) goes into the scalar base address
    const char* vbase = reinterpret_cast<const char*>(v_next + ((FAM == FAM_CASH_LEADTIME && !LAST) ? (int64_t)k * P.next.nx * P.next.nc : 0));
    // ---- the demand loop: serial in j, reference order -------------------------------------------
    // U demand steps per trip: their cash-dependent tails are formed first and the U gathers issued back to back (a
    // wave then has U loads in flight instead of one), and the accumulator takes the 2U addends afterwards in the
    // reference's order -- the same operations on the same operands as the one-step loop.
    constexpr int U = 4;
    double acc = 0.0;
    // one step's tail: add1 = the first addend (p * imm; survival, period T: p * [final cash >= 0]), off = byte offset of
    // the successor (or dead: a bankrupt successor of the survival family, worth 0)
    auto tail = [&](const RowEnt& e, const double2 pp, double& add1, uint32_t& off, bool& dead) {
      double inc;
      dead = false;
      off = 0;
      if constexpr (FAM == FAM_CASH) {
        if constexpr (!FORMULA1) {
          if constexpr (LEAN)
            inc = e.u + dep_or_bi - s.cash;  // (- holdCosts - overheadCost: both +0.0)
          else
            inc = e.u + dep_or_bi - e.hold - overhead - s.cash;
          if constexpr (LAST) inc += e.sal;
        } else {
          inc = e.u;
        }
        if constexpr (PEN) {
          const double end_cash = s.cash + inc;
          if (end_cash < 0) inc += P.pi * end_cash;
        }  // (with a zero penalty rate `inc += 0 * endCash` changes nothing and is skipped)
        add1 = pp.x * inc;
      } else if constexpr (FAM == FAM_SURVIVAL) {
        inc = e.u + dep_or_bi - e.hold - overhead - s.cash;
        if constexpr (LAST) {
          inc += e.sal;
          add1 = pp.x * ((s.cash + inc) >= 0 ? 1.0 : 0.0);
        } else {
          add1 = 0.0;  // (not added: the survival recursion has no immediate term before period T)
        }
      } else {  // F4 / F5: CashOverdraft.java:99-104
        const double after = dep_or_bi + e.u;
        inc = after - s.cash;
        if constexpr (LAST) inc += e.sal;
        add1 = pp.x * inc;
      }
      if constexpr (!LAST) {
        const int key = cash_key_row<INTDIV>(s.cash + inc, key_lo_v, key_hi_v, round_mult, round_div);  // CashConstraint.java:125-131
        if constexpr (FAM == FAM_SURVIVAL) dead = key < 0;
        off = (uint32_t)(e.rowoff8 + (key << 3));
      }
    };
    // the lane's own cash key, as a byte offset (uniform-key trips add the step's shift to it)
    [[maybe_unused]] const int my_key = (int)P.cur.k_lo + ic_c;
    [[maybe_unused]] const int* uni = s_uni + wave * ((D + 3) / 4);
    int j = 0;
    for (; j + U <= D; j += U) {
      double add1[U], pg[U], v[U];
      uint32_t off[U];
      bool dead[U];
      if constexpr (UNI && !LAST) {
        if (__builtin_amdgcn_readfirstlane(uni[j / U])) {  // all four steps shift every cash point by a fixed number of keys
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const RowEnt e = ent[j + u];
            const double2 pp = s_p[j + u];
            pg[u] = pp.y;
            double inc;
            if constexpr (FORMULA1)
              inc = e.u;
            else if constexpr (LEAN)
              inc = e.u + dep_or_bi - s.cash;
            else
              inc = e.u + dep_or_bi - e.hold - overhead - s.cash;
            add1[u] = pp.x * inc;
            off[u] = (uint32_t)(e.rowoff8 + (med3_i32(my_key + e.dkey, key_lo_v, key_hi_v) << 3));
          }
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const double*>(vbase + off[u]);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            acc += add1[u];
            acc += pg[u] * v[u];
          }
          continue;
        }
      }
      if constexpr (!LAST) SDP_TRIP_PHASE(1);  // (two phases per trip, see cash_row_pair_kernel)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double2 pp = s_p[j + u];
        pg[u] = pp.y;
        tail(ent[j + u], pp, add1[u], off[u], dead[u]);
      }
      if constexpr (!LAST) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const double*>(vbase + off[u]);
        SDP_TRIP_PHASE(0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (FAM != FAM_SURVIVAL || LAST) acc += add1[u];
        if constexpr (!LAST) acc += pg[u] * (dead[u] ? 0.0 : v[u]);
      }
    }
    for (; j < D; ++j) {
      const double2 pp = s_p[j];
      double add1;
      uint32_t off;
      bool dead;
      tail(ent[j], pp, add1, off, dead);
      if constexpr (FAM != FAM_SURVIVAL || LAST) acc += add1;
      if constexpr (!LAST) acc += pp.y * (dead ? 0.0 : *reinterpret_cast<const double*>(vbase + off));
    }
    __builtin_amdgcn_wave_barrier();
    if (k < nA && (MAXDIR ? (acc > best) : (acc < best))) {
      best = acc;
      bestk = k;
    }
  }

  s_val[wave * 64 + lane] = best;
  s_k[wave * 64 + lane] = bestk;
  __syncthreads();
  if (tid < 64 && live) {
    double bv = s_val[tid];
    int bk = s_k[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      double ov = s_val[w * 64 + tid];
      int ok = s_k[w * 64 + tid];
      if (MAXDIR ? better<true>(ov, ok, bv, bk) : better<false>(ov, ok, bv, bk)) {
        bv = ov;
        bk = ok;
      }
    }
    v_cur[idx] = bv;
    pol[idx] = bk;
  }
}

// ---------------------------------------------------------------------------------------------
// The wave-uniform part of a (row, action) of the F3 pair kernel -- one RowEnt per demand point, the uniform-key flag of every
// trip of four steps, the smallest and largest uniform shift -- as a BLOCK of bytes, laid out the way the kernel's demand loop
// reads it from LDS:  [RowEnt x D] [int flag x row_tab_flags(D)] (pad to 16) [RowTabHead].
// cash_row_pair_kernel forms it itself per (tile, action), lanes = demand indices (25 of 64 lanes busy on CashConstraint.main);
// with TAB it is formed ONCE per (row, action) by cash_row_table_kernel and every tile of the row copies the finished block
// into its LDS (one 16-byte load and store per lane, requested an action ahead): the ~100 setup instructions per (wave,
// action) leave the hot kernel.  row_entry() is the one place the entry's arithmetic is written: both paths call it.
// ---------------------------------------------------------------------------------------------
// Entry slots per wave of cash_row_pair_kernel's in-kernel setup: two when a pmf fits half a wave.
__host__ __device__ inline int cash_row_slots(int D) { return D <= 32 ? 2 : 1; }

struct RowTabHead {
  int32_t dmin, dmax;  // smallest / largest key shift over the uniform steps of the action (0, 0: none)
  int32_t pad0, pad1;
};
__host__ __device__ inline int row_tab_flags(int D) { return (D + 3) / 4 + 3; }  // whole trips, then the D mod 4 single steps
__host__ __device__ inline int row_tab_head_off(int D) { return (D * 32 + row_tab_flags(D) * 4 + 15) & ~15; }
__host__ __device__ inline int row_tab_block(int D) { return row_tab_head_off(D) + 16; }

// off8 = rowoff8 + 8 * dkey (uniform steps; rowoff8 otherwise): what a clamp-free trip adds to the lane's own byte offset.
// It travels in the `sal` slot of the entry, which only period T reads (and period T gathers nothing).
__device__ __forceinline__ int row_ent_off8(const RowEnt& e) { return __double2loint(e.sal); }

template <bool LAST, bool FORMULA1, bool LEAN, int FAM = FAM_CASH>
__device__ __forceinline__ RowEnt row_entry(const DevParams& P, double y, double fixed, double var, double d, int k_lo_next,
                                            bool& is_uni) {
  const double revenue = P.price * jmin(y, d);
  const double level = y - d;
  const double pos = jmax(level, 0.0);
  RowEnt e;
  e.hold = P.h * pos;
  e.sal = LAST ? P.salvage * pos : 0.0;
  if constexpr (FAM != FAM_CASH) {
    e.u = revenue;  // F4 / F5: after = (before - interest) + revenue (CashOverdraft.java:99-104, SingleProductLeadtime.java:98-104)
  } else if constexpr (!FORMULA1) {
    e.u = P.one_minus_overhead_rate * revenue;
  } else {
    double inc = revenue - fixed - var - e.hold - P.overhead;
    inc += e.sal;
    e.u = inc;
  }
  e.rowoff8 = 0;
  e.dkey = kNoShift;
  is_uni = false;
  if constexpr (!LAST) {
    double ninv = jmax(0.0, level);
    ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
    ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
    e.rowoff8 = (inv_index(P, ninv) * (int)P.next.nc - k_lo_next) * 8;
    // the increment with the cash balance cancelled (real arithmetic).  F4 / F5: only where the balance pays no interest
    // (the kernel ANDs the flag with that, per tile and action): before = cash - fixed - var - overhead, after = before + revenue
    // (`fixed` is 0 for F5, whose lambdas charge none: SingleProductLeadtime.java:88-93)
    const double inc_u = FAM != FAM_CASH ? e.u - fixed - var - P.overhead
                                         : (FORMULA1 ? e.u : (LEAN ? e.u - fixed - var : e.u - fixed - var - e.hold - P.overhead));
    const double dm = inc_u * P.round_mult;
    const double dn = rint(dm);
    is_uni = fabs(dm - dn) < 9.5367431640625e-07 && fabs(dn) < 1.0e9;  // 2^-20, see cash_row_kernel
    e.dkey = is_uni ? (int)dn : kNoShift;
    e.sal = __hiloint2double(0, e.rowoff8 + (is_uni ? e.dkey * 8 : 0));
  }
  return e;
}

// One wave per (row, action): the block of that action of that row, written to `tab` (see above).
template <bool LAST, bool FORMULA1, bool LEAN>
__global__ __launch_bounds__(256) void cash_row_table_kernel(DevParams P, const double* __restrict__ pmf_d, char* __restrict__ tab,
                                                             int64_t row0, int n_rows, int n_actions) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t g = (int64_t)blockIdx.x * 4 + wave;
  if (g >= (int64_t)n_rows * n_actions) return;
  const int row_i = (int)(g / n_actions), k = (int)(g - (int64_t)row_i * n_actions);
  const int D = P.n_demand;
  StateT s0;
  decode_state_row<FAM_CASH>(P, row0 + row_i, 0, s0);  // (the row's inventory level: the same in every cash point)
  const double a = (double)k * P.step;
  const double y = s0.x + a;
  const double fixed = a > 0 ? P.K : 0.0;
  const double var = P.v * a;
  const int k_lo_next = (int)P.next.k_lo;
  char* blk = tab + (size_t)g * row_tab_block(D);
  RowEnt* ent = reinterpret_cast<RowEnt*>(blk);
  int* uni = reinterpret_cast<int*>(blk + (size_t)D * 32);
  int dmin = 0x7fffffff, dmax = -0x7fffffff;
  for (int j0 = 0; j0 < D; j0 += 64) {
    const int j = j0 + lane;
    bool is_uni = false;
    if (j < D) {
      const RowEnt e = row_entry<LAST, FORMULA1, LEAN>(P, y, fixed, var, pmf_d[j], k_lo_next, is_uni);
      ent[j] = e;
      if (is_uni) {
        dmin = e.dkey < dmin ? e.dkey : dmin;
        dmax = e.dkey > dmax ? e.dkey : dmax;
      }
    }
    if constexpr (!LAST) {
      const unsigned long long mu = __ballot(is_uni);
      if (j < D && (lane & 3) == 0) uni[j / 4] = (((mu >> lane) & 15ull) == 15ull) ? 1 : 0;
      if (j < D && j >= (D & ~3)) uni[(D + 3) / 4 + (j & 3)] = is_uni ? 1 : 0;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int a0 = __shfl_xor(dmin, off, 64), b0 = __shfl_xor(dmax, off, 64);
    dmin = a0 < dmin ? a0 : dmin;
    dmax = b0 > dmax ? b0 : dmax;
  }
  if (lane == 0) {
    RowTabHead hd;
    hd.dmin = dmax < dmin ? 0 : dmin;
    hd.dmax = dmax < dmin ? 0 : dmax;
    hd.pad0 = hd.pad1 = 0;
    *reinterpret_cast<RowTabHead*>(blk + row_tab_head_off(D)) = hd;
  }
}

// ---------------------------------------------------------------------------------------------
// Cash row kernel, TWO ADJACENT CASH POINTS PER LANE (F3: formulas 0, 1 and 2 without deposit rate, penalty and
// integer division -- the CashConstraint / CashConstraintTesting / CashConstraintXR drivers).
//
// With the cash bands keeping the gathers in L2 and the uniform-key trips taking the quantiser out, what binds
// cash_row_kernel on CashConstraint.main's grid is the gather unit: one 64-lane 8-byte gather per cell-step costs the TA
// ~15 cycles whatever its width (2.1e12 cells/s = 82 % TA busy), exactly where the uniform-shift kernel stood before
// it paired its points.  In a uniform-key trip the two points c, c + 1 of a lane go to the keys k + delta, k + delta + 1
// of one row: ONE 16-byte gather serves both (picked apart where the grid's ends clamp them onto one entry; a trip in
// which no point of the wave's 128 clamps needs neither the clamp nor the selects).  Trips that are not uniform fall
// back to the quantiser and one 8-byte gather per point.  Everything that enters an accumulator is formed per point
// exactly as in cash_row_kernel.
// Trip flag (s_uni): 0 = not uniform, 1 = uniform, 2 = uniform and clamp-free for this wave's tile.
// ---------------------------------------------------------------------------------------------
// S = 128-point tiles per wave: lane l owns the pairs at ic0 + 128 s + 2 l, s < S.  The per-action setup (lanes = demand
// indices: with 25 demand points only 25 of 64 lanes work) and the LDS reads of a step's entries are shared by the S pairs.
// SRC: where a (row, action)'s block of wave-uniform operands comes from.
//  0  every wave forms the blocks of its own actions (k = wave, wave + 4, ...) for its workgroup's ONE tile; the four waves'
//     results are merged through LDS.  Round 2's form.
//  2  SHARED (round 3; opt-in SDPGPU_CASH_SHARE=1, measured slower, see the launcher): a workgroup is FOUR tiles of one row, one per wave, and
//     walks the actions in groups of four: wave w forms the block of action k0 + w, a barrier, then every wave runs the four
//     actions on its own tile.  The ~100 setup instructions per block are spent once per workgroup instead of once per
//     wave-action -- a quarter of the setup per cell -- with no memory traffic, and a wave owns its tile's arg-max (no merge).
//     The blocks carry tile-free flags (uniform trip or not) and the action's smallest / largest shift; a wave decides
//     "clamp-free" per action from those.
//  1  TABLE: the blocks come from cash_row_table_kernel's table (`tab`, `tab_actions` blocks per row), copied into two LDS
//     buffers per wave an action ahead.  Built and measured SLOWER than 0 (42.1 against 38.2 ms per sweep on
//     CashConstraint.main's grid): vector loads return in order, so every first gather of an action waits behind the block
//     load, which misses the XCD's L2 on a row's first tile.  Kept behind SDPGPU_CASH_TAB=1.
// FAM (round 4): FAM_CASH as above; FAM_OVERDRAFT / FAM_CASH_LEADTIME (F4 / F5: CashOverdraft, SingleProductLeadtime; SRC 0 only,
// FORMULA1 = LEAN = false, no integer division in the quantiser).  Their increment is (before - interest(before) + revenue) - cash
// with before = cash - fixed - var - overhead: where the piecewise interest of EVERY point of the wave's tile is zero for the
// action (r0 == 0 and before >= -interestFreeAmount: decided per tile and action from the points' own `before`, one wave vote)
// the balance cancels in real arithmetic exactly as in F3 and the uniform-key trips apply -- one 16-byte gather per two cells
// on a kernel the gather unit binds (rocprofv3, SingleProductLeadtime's size: TA 84 % busy at one 8-byte gather per cell, VALU
// 75 %); elsewhere (balances that pay interest: the shift depends on the balance) the quantiser and one gather per point.
// F5's entries do not depend on the action (its level is x + preQ) except for the shift; the next pipeline plane (preQ' = action)
// goes into the scalar base address.
// RW = rows per workgroup (round 4).  1: a workgroup is ONE tile of one row, its four waves take the actions k = wave, wave + 4, ...
// and merge through LDS.  4 (F5): a workgroup is one tile of FOUR rows consecutive in the launcher's row order, one row per wave, and
// every wave walks ALL its row's actions and owns its arg-opt.  F5's rows are walked in order of the level x + preQ, and rows of one
// level gather the very same entries of V_{t+1} for the same tile, action and demand: with one row per workgroup those requests come
// from different compute units and every one is served by the L2 -- rocprofv3 on SingleProductLeadtime's size: 6.3e9 L2 requests per
// launch = 13.7 TB/s of 128-byte lines, three quarters of what the XCDs' L2s deliver to gathers, while TA (65 %) and VALU (59 %) idle;
// neither fewer instructions nor fewer gathers moved the time.  Four rows of a level on ONE compute unit, in step, find each other's
// lines in its vector L1.  Placement only: every cell still issues its own gather and performs its own arithmetic.
template <bool LAST, bool FORMULA1, bool LEAN, int S, int SRC = 0, int FAM = FAM_CASH, int RW = 1>
__global__ __launch_bounds__(256) void cash_row_pair_kernel(DevParams P, const double* __restrict__ v_next,
                                                            double* __restrict__ v_cur, int32_t* __restrict__ pol,
                                                            const double* __restrict__ pmf_d,
                                                            const double* __restrict__ pmf_p, int64_t lo, int64_t hi,
                                                            int64_t row0, RowTiling G, const char* __restrict__ tab,
                                                            int tab_actions) {
  constexpr bool OD = FAM != FAM_CASH;  // F4 / F5
  static_assert(FAM == FAM_CASH || ((FAM == FAM_OVERDRAFT || FAM == FAM_CASH_LEADTIME) && SRC == 0 && !FORMULA1 && !LEAN),
                "the overdraft families run the in-kernel setup only");
  static_assert(RW == 1 || (RW == 4 && SRC == 0), "four rows per workgroup: in-kernel setup only");
  constexpr int KS = RW == 4 ? 1 : 4;  // stride of a wave's actions
  constexpr int TS = 128 * S;
  constexpr int NP = 2 * S;  // cash points per lane: point p = 2 s + w is ic0 + 128 s + 2 lane + w
  constexpr bool TAB = SRC == 1, SHARE = SRC == 2, BLK = SRC != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int D = P.n_demand;
  const int BS = row_tab_block(D);
  double2* s_p = reinterpret_cast<double2*>(smem);                  // {p_j, p_j * gamma}
  // SRC 0: [4 waves][D] entries, d_j, scratch, [4 waves] flags.  TAB: [4 waves][2 buffers][BS] bytes, scratch.
  // SHARE: [4 actions][BS] bytes, d_j, the four waves' action counts.
  RowEnt* s_ent = reinterpret_cast<RowEnt*>(s_p + D);
  char* s_blocks = reinterpret_cast<char*>(s_p + D);
  // SRC 0, pmfs of at most 32 points: TWO entry slots per wave -- one setup pass forms the entries of two of the wave's actions,
  // lanes 0-31 those of action k, lanes 32-63 those of action k + 4 (cash_row_slots; the launcher sizes the LDS with it)
  const int SLOTS = (SRC == 0 && G.slots == 2) ? 2 : 1;
  double* s_d = SHARE ? reinterpret_cast<double*>(s_blocks + (size_t)4 * BS) : reinterpret_cast<double*>(s_ent + (size_t)4 * SLOTS * D);  // d_j
  int* s_na = reinterpret_cast<int*>(s_d + D);
  double* s_val = TAB ? reinterpret_cast<double*>(s_blocks + (size_t)8 * BS) : s_d + D;
  int* s_k = reinterpret_cast<int*>(s_val + 4 * TS);
  int* s_uni = s_k + 4 * TS;  // (SRC 0) [4 waves][(D + 3) / 4 trips + 3 single steps]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int row_i, tile;
  if (G.units) {
    // segments of `units_seg` consecutive units of the order go round the XCDs: segment s to XCD s % 8 (a contiguous eighth per
    // XCD would hand the low-cash end -- balances that pay interest: the quantiser, one gather per cell -- to XCDs 0-2 and the
    // cheap uniform-key cells to the rest, and the launch lasts as long as its slowest XCD)
    const int n = blockIdx.x >> 3;
    const int64_t u = ((int64_t)(n / G.units_seg) * 8 + (blockIdx.x & 7)) * G.units_seg + n % G.units_seg;
    if (u >= G.units_total) return;
    const int2 ut = G.units[u];
    row_i = ut.x;
    tile = ut.y;
  } else if (G.colmajor > 0) {
    const int64_t u = (int64_t)(blockIdx.x & 7) * G.colmajor + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= (unsigned)G.colmajor || u >= (int64_t)G.n_rows * G.tiles_per_row) return;
    tile = (int)(u / G.n_rows);
    row_i = (int)(u - (int64_t)tile * G.n_rows);
  } else if (G.tps == 0) {
    row_i = blockIdx.x / G.tiles_per_row;
    tile = blockIdx.x % G.tiles_per_row;
  } else {
    const int xcd = blockIdx.x & 7, n = blockIdx.x >> 3;
    const int per_band = G.n_rows * G.tps;
    const int sub = n / per_band, rem = n - sub * per_band;
    row_i = rem / G.tps;
    tile = (G.band_interleave ? sub * 8 + xcd : xcd * G.nsub + sub) * G.tps + (rem - row_i * G.tps);
    if (tile >= G.tiles_per_row) return;
  }
  for (int j = tid; j < D; j += 256) {
    if constexpr (!TAB) s_d[j] = pmf_d[j];
    s_p[j] = make_double2(pmf_p[j], pmf_p[j] * P.gamma);
  }
  __syncthreads();

  if constexpr (RW == 4) {  // (G.n_rows counts GROUPS of four rows, G.rows_real the rows)
    row_i = row_i * 4 + wave;
    if (row_i >= G.rows_real) return;  // a wave without a row (the last group): nothing below meets a workgroup barrier
  }
  const int64_t row = row0 + (G.perm ? G.perm[row_i] : row_i);
  const int ic0 = (SHARE ? tile * 4 + wave : tile) * TS;  // (SHARE: G counts workgroup tiles of 4 TS points, one TS per wave)
  const int nc = (int)P.cur.nc;
  StateT s[NP];
  int nA[NP], icc[NP];
  int nA_max = 0;
#pragma unroll
  for (int w = 0; w < NP; ++w) {
    const int ic = ic0 + 128 * (w >> 1) + 2 * lane + (w & 1);
    icc[w] = ic < nc ? ic : nc - 1;
    decode_state_row<FAM>(P, row, icc[w], s[w]);  // x is the same in every lane and for both points
    nA[w] = n_actions<FAM>(P, s[w]);
    nA_max = nA[w] > nA_max ? nA[w] : nA_max;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_xor(nA_max, off, 64);
    nA_max = o > nA_max ? o : nA_max;
  }
  nA_max = __builtin_amdgcn_readfirstlane(nA_max);
  if (SHARE && ic0 >= nc) nA_max = 0;  // (a wave past the row's end: it still forms blocks and meets the barriers)

  const int NF = (D + 3) / 4 + 3;                // trip flags of one action
  RowEnt* const ent0 = s_ent + (size_t)wave * SLOTS * D;
  int* const uni0 = s_uni + wave * SLOTS * NF;
  RowEnt* ent = ent0;                            // (in-kernel setup; TAB: re-pointed at the current block every action)
  int* uni = uni0;
  const bool MAXDIR = P.maxdir != 0;
  double best[NP];
  int bestk[NP];
#pragma unroll
  for (int w = 0; w < NP; ++w) {
    best[w] = MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;
    bestk[w] = 0;
  }
  const double round_mult = P.round_mult, round_div = P.round_div;
  const double overhead = P.overhead;
  const int k_lo_next = (int)P.next.k_lo;
  const int k_hi_next = k_lo_next + (int)P.next.nc - 1;
  int key_lo_v, key_hi_v, key_hi1_v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(key_lo_v) : "s"(k_lo_next));
  asm volatile("v_mov_b32 %0, %1" : "=v"(key_hi_v) : "s"(k_hi_next));
  asm volatile("v_mov_b32 %0, %1" : "=v"(key_hi1_v) : "s"(k_hi_next - 1));
  int my_key[S], my_key8[S];  // key of each pair's first point; the second is my_key + 1 in a whole tile
#pragma unroll
  for (int t = 0; t < S; ++t) {
    my_key[t] = (int)P.cur.k_lo + icc[2 * t];
    my_key8[t] = my_key[t] * 8;
  }
  // clamp-free trips need every point of the tile to exist and to stay inside the grid under the step's shift
  const bool tile_whole = ic0 + TS <= nc;
  const int key_first = (int)P.cur.k_lo + ic0, key_last = key_first + TS - 1;
  // TAB: this wave's two block buffers, the row's blocks in the table, and the synchronous part of a block copy
  char* my_blocks = s_blocks + (size_t)wave * 2 * BS;
  const char* row_tab = TAB ? tab + ((size_t)row_i * tab_actions) * (size_t)BS : nullptr;
  [[maybe_unused]] auto copy_block = [&](int kk, int b, int from) {  // bytes [from, BS) of block kk -> buffer b
    const char* src = row_tab + (size_t)kk * BS;
    for (int o = from + lane * 16; o < BS; o += 1024)
      *reinterpret_cast<uint4*>(my_blocks + (size_t)b * BS + o) = *reinterpret_cast<const uint4*>(src + o);
  };
  int buf = 0;
  if constexpr (TAB) {
    if (wave < nA_max) copy_block(wave, 0, 0);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  // One action on this wave's points: the demand loop in the reference's order, then the strict-compare update of the arg-opt.
  auto consume = [&](const int k, const double fixed, const double var, [[maybe_unused]] const bool free_action) {
    double dep[NP];  // F3: the deposit term; F4 / F5: cashBalanceBefore - interest
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      if constexpr (!OD) {
        dep[w] = (s[w].cash - fixed - var) * P.one_plus_deposit;
      } else {
        const double before = FAM == FAM_OVERDRAFT ? s[w].cash - fixed - var - overhead : s[w].cash - var - overhead;
        dep[w] = before - overdraft_interest(P, before);
      }
    }
    // (F5: the plane of the next pipeline quantity, preQ' = action, in the scalar base address)
    const char* vbase = reinterpret_cast<const char*>(v_next + ((FAM == FAM_CASH_LEADTIME && !LAST) ? (int64_t)k * P.next.nx * P.next.nc : 0));

    // the cash-dependent increment of one point, the reference's operations in the reference's order
    auto increment = [&](const RowEnt& e, int w) -> double {
      double inc;
      if constexpr (OD)
        inc = (dep[w] + e.u) - s[w].cash;  // after = before - interest + revenue; cashIncrement = after - iniCash
      else if constexpr (FORMULA1)
        inc = e.u;
      else if constexpr (LEAN)
        inc = e.u + dep[w] - s[w].cash;
      else
        inc = e.u + dep[w] - e.hold - overhead - s[w].cash;
      if constexpr (LAST && !FORMULA1) inc += e.sal;
      return inc;
    };
#ifndef SDP_PAIR_U
#define SDP_PAIR_U 2
#endif
    constexpr int G = 4;           // demand steps per trip FLAG (the setup's granularity)
    // Demand steps per trip of the loop.  Trips of TWO steps keep half the gathered pairs and products in registers: 114 VGPRs
    // instead of 153, i.e. FOUR waves per SIMD instead of three -- on a kernel that loads the VALU issue port and the vector L1
    // to three quarters each, the extra wave is worth more than the trip overhead it doubles (CashConstraint.main's grid:
    // 35.6 against 37.3 ms per sweep, same box, back to back; -DSDP_PAIR_U=4 rebuilds the four-step form).
    constexpr int U = SDP_PAIR_U;
    double acc[NP];
#pragma unroll
    for (int w = 0; w < NP; ++w) acc[w] = 0.0;
    int j = 0;
    for (int jg = 0; jg + G <= D; jg += G)
    for (j = jg; j < jg + G; j += U) {
      double add1[U][NP], pg[U];
      if constexpr (!LAST) {
        const int f_raw = __builtin_amdgcn_readfirstlane(uni[jg / G]);
        const int f = BLK ? (f_raw ? (free_action ? 2 : 1) : 0) : f_raw;  // (a block's flag knows no tile)
        // one uniform-key trip; FREE: no point of the wave's tiles clamps (decided for the whole trip, so that the four
        // steps are straight-line code: the compiler turned a per-step choice into mask arithmetic on every step)
        auto uni_trip = [&](auto free_tag) {
          constexpr bool FREE = decltype(free_tag)::value;
          dpair_u v[U][S];
          [[maybe_unused]] bool hi_fold[U][S], lo_fold[U][S];
          SDP_TRIP_PHASE(1);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const RowEnt e = ent[j + u];
            const double2 pp = s_p[j + u];
            pg[u] = pp.y;
#pragma unroll
            for (int w = 0; w < NP; ++w) add1[u][w] = pp.x * increment(e, w);
#pragma unroll
            for (int t = 0; t < S; ++t) {
              uint32_t off;
              if constexpr (FREE) {  // the pair sits at the lane's own offset plus the step's (row offset + shift)
                off = (uint32_t)(my_key8[t] + row_ent_off8(e));
              } else {  // {V[c], V[c + 1]}, c = clamp(key, lo, hi - 1): at the ends both points may fold onto one entry
                const int ka = my_key[t] + e.dkey;
                off = (uint32_t)(e.rowoff8 + (med3_i32(ka, key_lo_v, key_hi1_v) << 3));
                hi_fold[u][t] = ka > k_hi_next - 1;  // first point at or beyond the last key: it reads the pair's second entry
                lo_fold[u][t] = ka < k_lo_next;      // second point at or below the first key: it reads the pair's first entry
              }
              v[u][t] = *reinterpret_cast<const dpair_u*>(vbase + off);
            }
          }
          SDP_TRIP_PHASE(0);
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < S; ++t) {
              double v0 = v[u][t].x, v1 = v[u][t].y;
              if constexpr (!FREE) {
                v0 = hi_fold[u][t] ? v[u][t].y : v[u][t].x;
                v1 = lo_fold[u][t] ? v[u][t].x : v[u][t].y;
              }
              acc[2 * t] += add1[u][2 * t];
              acc[2 * t] += pg[u] * v0;
              acc[2 * t + 1] += add1[u][2 * t + 1];
              acc[2 * t + 1] += pg[u] * v1;
            }
        };
        if (f == 2) {
          uni_trip(std::true_type{});
          continue;
        }
        if (f == 1) {
          uni_trip(std::false_type{});
          continue;
        }
      }
      // not a uniform trip (or period T): the quantiser and one gather per point
      double v[U][NP];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const RowEnt e = ent[j + u];
        const double2 pp = s_p[j + u];
        pg[u] = pp.y;
#pragma unroll
        for (int w = 0; w < NP; ++w) {
          const double inc = increment(e, w);
          add1[u][w] = pp.x * inc;
          if constexpr (!LAST) {
            const int key = cash_key_row<false>(s[w].cash + inc, key_lo_v, key_hi_v, round_mult, round_div);
            v[u][w] = *reinterpret_cast<const double*>(vbase + (uint32_t)(e.rowoff8 + (key << 3)));
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int w = 0; w < NP; ++w) {
          acc[w] += add1[u][w];
          if constexpr (!LAST) acc[w] += pg[u] * v[u][w];
        }
    }
    for (j = D & ~(G - 1); j < D; ++j) {
      const RowEnt e = ent[j];
      const double2 pp = s_p[j];
      if constexpr (!LAST) {
        // a uniform-key step on its own (D is not a multiple of four: D = 25 in CashConstraint.main): the pair gather at the
        // shifted key instead of the quantiser and one gather per point
        const int f_raw = __builtin_amdgcn_readfirstlane(uni[(D + 3) / 4 + (j & 3)]);
        const int f = BLK ? (f_raw ? (free_action ? 2 : 1) : 0) : f_raw;
        if (f == 2) {  // clamp-free: straight-line, as in the trips (the step's two forms under one flag cost 14 instructions per cell)
          dpair_u vv[S];
#pragma unroll
          for (int t = 0; t < S; ++t) vv[t] = *reinterpret_cast<const dpair_u*>(vbase + (uint32_t)(my_key8[t] + row_ent_off8(e)));
#pragma unroll
          for (int t = 0; t < S; ++t) {
            acc[2 * t] += pp.x * increment(e, 2 * t);
            acc[2 * t] += pp.y * vv[t].x;
            acc[2 * t + 1] += pp.x * increment(e, 2 * t + 1);
            acc[2 * t + 1] += pp.y * vv[t].y;
          }
          continue;
        }
        if (f != 0) {
#pragma unroll
          for (int t = 0; t < S; ++t) {
            const int ka = my_key[t] + e.dkey;
            const uint32_t off = (uint32_t)(e.rowoff8 + (med3_i32(ka, key_lo_v, key_hi1_v) << 3));
            const dpair_u vv = *reinterpret_cast<const dpair_u*>(vbase + off);
            const double v0 = ka > k_hi_next - 1 ? vv.y : vv.x;
            const double v1 = ka < k_lo_next ? vv.x : vv.y;
            acc[2 * t] += pp.x * increment(e, 2 * t);
            acc[2 * t] += pp.y * v0;
            acc[2 * t + 1] += pp.x * increment(e, 2 * t + 1);
            acc[2 * t + 1] += pp.y * v1;
          }
          continue;
        }
      }
#pragma unroll
      for (int w = 0; w < NP; ++w) {
        const double inc = increment(e, w);
        acc[w] += pp.x * inc;
        if constexpr (!LAST) {
          const int key = cash_key_row<false>(s[w].cash + inc, key_lo_v, key_hi_v, round_mult, round_div);
          acc[w] += pp.y * *reinterpret_cast<const double*>(vbase + (uint32_t)(e.rowoff8 + (key << 3)));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int w = 0; w < NP; ++w)
      if (k < nA[w] && (MAXDIR ? (acc[w] > best[w]) : (acc[w] < best[w]))) {
        best[w] = acc[w];
        bestk[w] = k;
      }
  };
  if constexpr (SHARE) {
    // the block of (row, k) into `dst`, lanes = demand indices (cash_row_table_kernel's body, writing LDS)
    auto form_block = [&](const int k, char* dst) {
      const double a = (double)k * P.step;
      const double y = s[0].x + a;
      const double fixed = a > 0 ? P.K : 0.0;
      const double var = P.v * a;
      RowEnt* e_out = reinterpret_cast<RowEnt*>(dst);
      int* u_out = reinterpret_cast<int*>(dst + (size_t)D * 32);
      int dmin = 0x7fffffff, dmax = -0x7fffffff;
      for (int j0 = 0; j0 < D; j0 += 64) {
        const int j = j0 + lane;
        bool is_uni = false;
        if (j < D) {
          const RowEnt e = row_entry<LAST, FORMULA1, LEAN>(P, y, fixed, var, s_d[j], k_lo_next, is_uni);
          e_out[j] = e;
          if (is_uni) {
            dmin = e.dkey < dmin ? e.dkey : dmin;
            dmax = e.dkey > dmax ? e.dkey : dmax;
          }
        }
        if constexpr (!LAST) {
          const unsigned long long mu = __ballot(is_uni);
          if (j < D && (lane & 3) == 0) u_out[j / 4] = (((mu >> lane) & 15ull) == 15ull) ? 1 : 0;
          if (j < D && j >= (D & ~3)) u_out[(D + 3) / 4 + (j & 3)] = is_uni ? 1 : 0;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const int a0 = __shfl_xor(dmin, off, 64), b0 = __shfl_xor(dmax, off, 64);
        dmin = a0 < dmin ? a0 : dmin;
        dmax = b0 > dmax ? b0 : dmax;
      }
      if (lane == 0) {
        RowTabHead hd;
        hd.dmin = dmax < dmin ? 0 : dmin;
        hd.dmax = dmax < dmin ? 0 : dmax;
        hd.pad0 = hd.pad1 = 0;
        *reinterpret_cast<RowTabHead*>(dst + row_tab_head_off(D)) = hd;
      }
    };
    if (lane == 0) s_na[wave] = nA_max;
    __syncthreads();
    int nA_wg = s_na[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) nA_wg = s_na[w] > nA_wg ? s_na[w] : nA_wg;
    nA_wg = __builtin_amdgcn_readfirstlane(nA_wg);
    for (int k0 = 0; k0 < nA_wg; k0 += 4) {
      if (k0 + wave < nA_wg) form_block(k0 + wave, s_blocks + (size_t)wave * BS);
      __syncthreads();
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + i;
        if (k >= nA_max) break;
        const char* blk = s_blocks + (size_t)i * BS;
        ent = reinterpret_cast<RowEnt*>(const_cast<char*>(blk));
        uni = reinterpret_cast<int*>(const_cast<char*>(blk) + (size_t)D * 32);
        const RowTabHead hd = *reinterpret_cast<const RowTabHead*>(blk + row_tab_head_off(D));
        const int dmin = __builtin_amdgcn_readfirstlane(hd.dmin), dmax = __builtin_amdgcn_readfirstlane(hd.dmax);
        const bool free_action = tile_whole && (int64_t)key_first + dmin >= (int64_t)k_lo_next && (int64_t)key_last + dmax <= (int64_t)k_hi_next;
        const double a = (double)k * P.step;
        consume(k, a > 0 ? P.K : 0.0, P.v * a, free_action);
      }
      __syncthreads();  // (the blocks are rewritten by the next group)
    }
    // a wave owns its tile: results go straight out (lane l holds the adjacent points 2 l, 2 l + 1 of each 128-point piece)
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      const int ic = ic0 + 128 * (w >> 1) + 2 * lane + (w & 1);
      const int64_t idx = row * nc + ic;
      if (ic < nc && idx >= lo && idx < hi) {
        v_cur[idx] = best[w];
        pol[idx] = bestk[w];
      }
    }
    return;
  }
  [[maybe_unused]] bool next_ready = false;  // (two slots) the entries of this action were formed by the previous pass
  // F4 / F5: "no point of this wave's tile pays interest under this order": interest(before) is a zero for before >= 0 when
  // r0 == 0 and for -interestFreeAmount <= before < 0 (CashOverdraft.java:87-95); `before` is the points' own
  [[maybe_unused]] auto interest_free_tile = [&](const double fixed_a, const double var_a) -> bool {
    bool f = true;
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      const double before = FAM == FAM_OVERDRAFT ? s[w].cash - fixed_a - var_a - overhead : s[w].cash - var_a - overhead;
      f = f && before >= -P.interest_free && (before < 0 || P.r0 == 0.0);
    }
    return __all(f) != 0;
  };
  for (int k = RW == 4 ? 0 : wave; k < nA_max; k += KS) {
    const double a = (double)k * P.step;
    const double fixed = FAM == FAM_CASH_LEADTIME ? 0.0 : (a > 0 ? P.K : 0.0);  // (F5's lambdas charge no fixed cost)
    const double var = P.v * a;
    [[maybe_unused]] bool free_action = false, has_next = false;
    [[maybe_unused]] uint4 pf[2];
    if constexpr (TAB) {
      // ---- the finished block of (row, k) is in buffer `buf`; request the first 2 KB of the block of k + 4 ---------------
      ent = reinterpret_cast<RowEnt*>(my_blocks + (size_t)buf * BS);
      uni = reinterpret_cast<int*>(my_blocks + (size_t)buf * BS + (size_t)D * 32);
      const RowTabHead hd = *reinterpret_cast<const RowTabHead*>(my_blocks + (size_t)buf * BS + row_tab_head_off(D));
      const int dmin = __builtin_amdgcn_readfirstlane(hd.dmin), dmax = __builtin_amdgcn_readfirstlane(hd.dmax);
      // clamp-free: every point of the tile exists and stays inside the next grid under every uniform shift of the action
      free_action = tile_whole && (int64_t)key_first + dmin >= (int64_t)k_lo_next && (int64_t)key_last + dmax <= (int64_t)k_hi_next;
      has_next = k + 4 < nA_max;
      if (has_next) {
        const char* src = row_tab + (size_t)(k + 4) * BS;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int o = u * 1024 + lane * 16;
          if (o < BS) pf[u] = *reinterpret_cast<const uint4*>(src + o);
        }
      }
    } else if (SLOTS == 2) {
      // ---- two actions per setup pass: lanes 0-31 form the entries of action k (slot 0), lanes 32-63 those of the wave's next
      // action k + KS (slot 1), which the NEXT trip of this loop consumes without a setup of its own.  Per lane the same row_entry()
      // on the same operands as the one-action pass below.
      if (next_ready) {
        next_ready = false;
        ent = ent0 + D;
        uni = uni0 + NF;
      } else {
        const int half = lane >> 5, j = lane & 31;
        const bool second = k + KS < nA_max;
        const double a_l = (double)(k + KS * half) * P.step;
        const double fixed_l = FAM == FAM_CASH_LEADTIME ? 0.0 : (a_l > 0 ? P.K : 0.0);
        const double var_l = P.v * a_l;
        const bool act = j < D && (half == 0 || second);
        bool is_uni = false, free_ = false;
        [[maybe_unused]] bool int_free_l = true;
        if constexpr (OD && !LAST) {
          const double a4 = (double)(k + KS) * P.step;
          const bool f0 = interest_free_tile(fixed, var);
          const bool f1 = interest_free_tile(FAM == FAM_CASH_LEADTIME ? 0.0 : (a4 > 0 ? P.K : 0.0), P.v * a4);
          int_free_l = half ? f1 : f0;
        }
        RowEnt e{};
        if (act) {
          e = row_entry<LAST, FORMULA1, LEAN, FAM>(P, FAM == FAM_CASH_LEADTIME ? s[0].x + s[0].preq : s[0].x + a_l, fixed_l, var_l,
                                                   s_d[j], k_lo_next, is_uni);
          if constexpr (OD) is_uni = is_uni && int_free_l;
          if constexpr (!LAST) {
            const double dn = (double)e.dkey;
            free_ = is_uni && tile_whole && (double)key_first + dn >= (double)k_lo_next && (double)key_last + dn <= (double)k_hi_next;
          }
        }
        if constexpr (!LAST) {
          const unsigned long long mu = __ballot(is_uni), mf = __ballot(free_);
          int* un = uni0 + half * NF;
          if (act && (lane & 3) == 0)
            un[j / 4] = (((mf >> lane) & 15ull) == 15ull) ? 2 : ((((mu >> lane) & 15ull) == 15ull) ? 1 : 0);
          if (act && j >= (D & ~3)) un[(D + 3) / 4 + (j & 3)] = free_ ? 2 : (is_uni ? 1 : 0);
        }
        if (act) ent0[half * D + j] = e;
        next_ready = second;
        ent = ent0;
        uni = uni0;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
      }
    } else {
      // ---- wave-uniform part, lanes = demand indices (as cash_row_kernel) ---------------------------------------
      const double y = FAM == FAM_CASH_LEADTIME ? s[0].x + s[0].preq : s[0].x + a;
      [[maybe_unused]] bool int_free = true;
      if constexpr (OD && !LAST) int_free = interest_free_tile(fixed, var);
      for (int j = lane; j < D; j += 64) {
        bool is_uni;
        const RowEnt e = row_entry<LAST, FORMULA1, LEAN, FAM>(P, y, fixed, var, s_d[j], k_lo_next, is_uni);
        if constexpr (OD) is_uni = is_uni && int_free;
        if constexpr (!LAST) {
          const double dn = (double)e.dkey;
          const bool free_ = is_uni && tile_whole && (double)key_first + dn >= (double)k_lo_next &&
                             (double)key_last + dn <= (double)k_hi_next;
          const unsigned long long mu = __ballot(is_uni), mf = __ballot(free_);
          if ((lane & 3) == 0)
            uni[j / 4] = (((mf >> lane) & 15ull) == 15ull) ? 2 : ((((mu >> lane) & 15ull) == 15ull) ? 1 : 0);
          // the D mod 4 steps behind the last whole trip carry a flag each (they run one at a time)
          if (j >= (D & ~3)) uni[(D + 3) / 4 + (j & 3)] = free_ ? 2 : (is_uni ? 1 : 0);
        }
        ent[j] = e;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
    }

    consume(k, fixed, var, free_action);
    if constexpr (TAB) {
      // the block of this wave's next action goes into the other buffer (nobody reads that one any more)
      if (has_next) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int o = u * 1024 + lane * 16;
          if (o < BS) *reinterpret_cast<uint4*>(my_blocks + (size_t)(buf ^ 1) * BS + o) = pf[u];
        }
        if (BS > 2048) copy_block(k + 4, buf ^ 1, 2048);
      }
      buf ^= 1;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);
    }
  }

  if constexpr (RW == 4) {  // a wave owns its row's tile: results go straight out
#pragma unroll
    for (int w = 0; w < NP; ++w) {
      const int ic = ic0 + 128 * (w >> 1) + 2 * lane + (w & 1);
      const int64_t idx = row * nc + ic;
      if (ic < nc && idx >= lo && idx < hi) {
        v_cur[idx] = best[w];
        pol[idx] = bestk[w];
      }
    }
    return;
  }
#pragma unroll
  for (int w = 0; w < NP; ++w) {
    s_val[wave * TS + 128 * (w >> 1) + 2 * lane + (w & 1)] = best[w];
    s_k[wave * TS + 128 * (w >> 1) + 2 * lane + (w & 1)] = bestk[w];
  }
  __syncthreads();
  for (int q = tid; q < TS; q += 256) {
    const int ic = ic0 + q;
    const int64_t idx = row * nc + ic;
    if (ic < nc && idx >= lo && idx < hi) {
      double bv = s_val[q];
      int bk = s_k[q];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const double ov = s_val[w * TS + q];
        const int ok = s_k[w * TS + q];
        if (MAXDIR ? better<true>(ov, ok, bv, bk) : better<false>(ov, ok, bv, bk)) {
          bv = ov;
          bk = ok;
        }
      }
      v_cur[idx] = bv;
      pol[idx] = bk;
    }
  }
}

}  // namespace sdp

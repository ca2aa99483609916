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
#include "sdp_device.hpp"

namespace sdp {

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
  int32_t tiles_per_row;  // ceil(nc / 64)
  int32_t row0;           // first inventory row launched
};

struct ShiftEntry {
  double t1;      // p_j * inc
  int32_t rowoff; // next inventory index * nc
  int32_t delta;  // cash index shift
};

template <bool MAXDIR, bool LAST>
__global__ __launch_bounds__(256) void cash_shift_kernel(CashShiftParams P, const double* __restrict__ v_next,
                                                         double* __restrict__ v_cur, int32_t* __restrict__ pol,
                                                         const double* __restrict__ pmf_d,
                                                         const double* __restrict__ pmf_p, int64_t lo, int64_t hi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int D = P.n_demand;
  double2* s_pmf = reinterpret_cast<double2*>(smem);                     // {d_j, p_j * gamma}
  ShiftEntry* s_ent = reinterpret_cast<ShiftEntry*>(smem + (size_t)D * 16);  // [4 waves][D]
  double* s_val = reinterpret_cast<double*>(smem + (size_t)D * 16 * 5);
  int* s_k = reinterpret_cast<int*>(s_val + 4 * 64);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int j = tid; j < D; j += 256) s_pmf[j] = make_double2(pmf_d[j], pmf_p[j] * P.gamma);
  __syncthreads();

  const int row = P.row0 + blockIdx.x / P.tiles_per_row;
  const int ic0 = (blockIdx.x % P.tiles_per_row) * 64;
  const int ic = ic0 + lane;
  const int64_t idx = (int64_t)row * P.nc + ic;
  const bool live = ic < P.nc && idx >= lo && idx < hi;
  const double x = P.x_lo + (double)row * P.step;

  // feasible action count per lane (CashConstraint.java:96-99) and the tile's maximum (cash ascending)
  const int ic_c = ic < P.nc ? ic : P.nc - 1;
  const double cash = (double)(P.k_lo + ic_c) / P.q;  // exact: q is a power of two
  double m = jmin(P.max_order_quantity, jmax(0.0, (cash - P.overhead - P.K) / P.v));
  const int nA = ((m != m) ? 0 : (int)m) + 1;
  int nA_max = nA;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_xor(nA_max, off, 64);
    nA_max = o > nA_max ? o : nA_max;
  }
  nA_max = __builtin_amdgcn_readfirstlane(nA_max);

  ShiftEntry* ent = s_ent + (size_t)wave * D;
  double best = MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;
  int bestk = 0;
  const int nc1 = P.nc - 1;
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
      e.rowoff = 0;
      e.delta = 0;
      if constexpr (!LAST) {
        double ninv = jmax(0.0, level);
        ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
        ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
        e.rowoff = (int)((ninv - P.next_x_lo) / P.step) * P.nc;
        e.delta = (int)jround_d(inc * P.q);
      }
      ent[j] = e;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
    // ---- the demand loop: serial in j, reference order (CashRecursion.java:113-122) ----
    double acc = 0.0;
    if constexpr (LAST) {
      for (int j = 0; j < D; ++j) acc += ent[j].t1;
    } else {
      for (int j = 0; j < D; ++j) {
        const ShiftEntry e = ent[j];
        const double pg = s_pmf[j].y;
        int t = ic_c + e.delta;
        t = t < 0 ? 0 : t;
        t = t > nc1 ? nc1 : t;
        acc += e.t1;
        acc += pg * v_next[e.rowoff + t];
      }
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
      if (better<MAXDIR>(ov, ok, bv, bk)) {
        bv = ov;
        bk = ok;
      }
    }
    v_cur[idx] = bv;
    pol[idx] = bk;
  }
}

}  // namespace sdp

// sdp_staff.hpp -- period kernel of the STAFF family: workforce.StaffRecursion.getExpectedValue
// (StaffRecursion.java:81-118) with the lambdas of WorkforcePlanning.java:84-101 / WorkforceTesting.java:91-107.
//
// What sets this family apart: the pmf is chosen by the hire-up-to level y = x + a (pmfs[t][min(y, rows-1)],
// :92-95), so there is no per-period demand tile to stage -- every (state, action) pair walks its own row, of its
// own length.  Layout: the table is stored TRANSPOSED, pT[j * rows + y], so that the 64 lanes of a wave -- 64
// consecutive staff numbers x, one action a, hence 64 consecutive levels y -- read one 512-byte line per
// realisation j; V_{t+1}[x + a - j] is likewise 64 consecutive doubles.  A level beyond the table clamps to the last
// row: those lanes read one address (a broadcast).
//
// One wave = 64 states x one group of consecutive actions; per (lane, action) the realisations are summed SERIALLY
// in the reference's order (:97-107), lanes with a shorter row simply drop out of the loop (exec mask).  The group's
// strict-compare arg-min (ascending action) goes to a partial row; combine_staff_kernel scans the groups in
// ascending order with the same strict compare, which is the reference's single scan (:110-113).
//
// Register block: the cells (a, j) and (a + 1, j + 1) of one state leave the same staff n = x + a - j behind, so the
// salary cost, the penalty, the clamp and the V_{t+1}[n] read are the same for both -- only the probability and the
// action's own hiring cost differ.  A lane therefore walks R consecutive actions along that anti-diagonal (step k:
// action r is at j = k - (R-1-r)), forming the n-dependent operands once per step and spending, per cell, the
// reference's two adds of totalCosts, two multiplies and two accumulator adds: hoisting an operation whose operands
// are identical does not change its result.  Off the ends of a row the table holds zeros (kStaffPadJ padded rows on
// either side of j = 0 .. maxj-1): a zero-probability step adds +0.0 twice, which leaves the accumulator as it is.
//
// Roofline: 8 B (V gather) + 8 B (probability) algorithmic per cell, both cache-resident (the table of a period is
// rows^2 * 8 B = 3-8 MB, V is kilobytes); the limiter is fp64/int VALU issue: 6 fp64 operations per cell plus ~10
// per step shared by R cells.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace sdp {

constexpr int kStaffPadJ = 8;  // zero rows before j = 0 and after j = maxj - 1 of the transposed table (>= R - 1)

struct StaffParams {
  double K, v, salary, pen;  // fixCost, unitVariCost, salary, unitPenalty
  int32_t min_staff;         // minStaffNum[t]
  int32_t n_actions;         // maxHireNum + 1
  int32_t n_rows;            // pmfs[t].length
  int32_t clamp, min_x, max_x;
  int32_t x_lo;       // staff number of state index 0 (this period)
  int32_t next_x_lo;  // ... of the next period
  int32_t nn_lo, nn_hi;  // bounds of the next staff number: [minX, maxX] when clamped, else the next period's box
  int32_t n_groups, group_actions;
  int64_t part_stride;  // elements between the partial rows of two groups
  int32_t uni_rows;     // staff_window_kernel: the scalar-probability form for blocks beyond the table's last row (0: never)
};

template <bool FUTURE>
__global__ __launch_bounds__(64) void staff_period_kernel(StaffParams P, const double* __restrict__ pT,
                                                          const int32_t* __restrict__ row_len,
                                                          const double* __restrict__ v_next,
                                                          double* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                          int64_t lo, int64_t hi) {
  const int64_t tile = blockIdx.x / P.n_groups;
  const int group = (int)(blockIdx.x - tile * P.n_groups);
  const int64_t idx = lo + tile * 64 + threadIdx.x;
  if (idx >= hi) return;
  const int x = P.x_lo + (int)idx;
  const int a_end = min(P.n_actions, (group + 1) * P.group_actions);
  double best = 1.7976931348623157e308;  // Double.MAX_VALUE (:89)
  int bestk = 0;                         // bestHireQty = 0 (:87)
  for (int a = group * P.group_actions; a < a_end; ++a) {
    const int y = x + a;
    const int row = y >= P.n_rows - 1 ? P.n_rows - 1 : y;  // :93-94
    const int nj = row_len[row];
    const double fixHire = a > 0 ? P.K : 0.0;
    const double variHire = P.v * (double)a;
    const double fv = fixHire + variHire;  // first add of totalCosts, the same for every realisation
    const double* prow = pT + row;
    double acc = 0.0;
    for (int j = 0; j < nj; ++j) {
      const int n = y - j;  // nextStaffNum
      const double salaryCost = P.salary * (double)n;
      const double penalty = n > P.min_staff ? 0.0 : P.pen * (double)(P.min_staff - n);
      const double imm = fv + salaryCost + penalty;
      const double p = prow[(size_t)j * (size_t)P.n_rows];
      acc += p * imm;
      if constexpr (FUTURE) {
        int nn = n;
        if (P.clamp) {
          nn = nn > P.max_x ? P.max_x : nn;
          nn = nn < P.min_x ? P.min_x : nn;
        }
        acc += p * v_next[nn - P.next_x_lo];
      }
    }
    if (acc < best) {
      best = acc;
      bestk = a;
    }
  }
  const int64_t at = (int64_t)group * P.part_stride + idx;
  out_val[at] = best;
  out_idx[at] = bestk;
}

// A handful of states (period 1 of a run from one initial staff number is ONE state): with lanes = states a wave of the kernels
// below holds one live lane and walks maxHireNum + 1 actions x a row of realisations serially -- 0.22 ms for the single state
// of WorkforceTesting.main's instance, 3 % of its sweep.  Here the lanes of a wave are 64 consecutive ACTIONS of one state,
// hence 64 consecutive levels: the same coalesced reads of the transposed table and of V_{t+1}, every cell's arithmetic as in
// staff_period_kernel, the realisations of a lane summed serially in the reference's order.  The wave's strict-compare arg-min
// in ascending action order (:110-113) is the minimum with the lowest action among equals, or the initial (Double.MAX_VALUE,
// 0) when no value is below it; the waves of a state are the `groups` combine_staff_kernel scans.
template <bool FUTURE>
__global__ __launch_bounds__(64) void staff_action_kernel(StaffParams P, const double* __restrict__ pT,
                                                          const int32_t* __restrict__ row_len,
                                                          const double* __restrict__ v_next,
                                                          double* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                          int64_t lo, int64_t hi) {
  const int64_t tile = blockIdx.x / P.n_groups;
  const int group = (int)(blockIdx.x - tile * P.n_groups);
  const int64_t idx = lo + tile;
  if (idx >= hi) return;
  const int x = P.x_lo + (int)idx;
  const int a = group * 64 + (int)threadIdx.x;
  double acc = 1.7976931348623157e308;
  if (a < P.n_actions) {
    const int y = x + a;
    const int row = y >= P.n_rows - 1 ? P.n_rows - 1 : y;
    const int nj = row_len[row];
    const double fixHire = a > 0 ? P.K : 0.0;
    const double variHire = P.v * (double)a;
    const double fv = fixHire + variHire;
    const double* prow = pT + row;
    acc = 0.0;
    // four realisations a trip, their reads issued together (a lane's chain is otherwise one dependent load per step); the steps
    // past the row's end read the zeros the table holds there (kStaffPadJ rows behind j = maxj - 1) and add +0.0, as in the
    // window kernel
    constexpr int U = 4;
    static_assert(U - 1 <= kStaffPadJ, "table padding behind the last row of realisations");
    for (int j0 = 0; j0 < nj; j0 += U) {
      double p[U], vn[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = y - (j0 + u);
        p[u] = prow[(size_t)(j0 + u) * (size_t)P.n_rows];
        if constexpr (FUTURE) {
          int nn = n > P.nn_hi ? P.nn_hi : n;
          nn = nn < P.nn_lo ? P.nn_lo : nn;
          vn[u] = v_next[nn - P.next_x_lo];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = y - (j0 + u);
        const double salaryCost = P.salary * (double)n;
        const double penalty = n > P.min_staff ? 0.0 : P.pen * (double)(P.min_staff - n);
        const double imm = fv + salaryCost + penalty;
        acc += p[u] * imm;
        if constexpr (FUTURE) acc += p[u] * vn[u];
      }
    }
  }
  // (an action beyond the range holds MAX_VALUE and a higher index than every real one: it never wins)
  double best = acc;
  int bestk = a;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_xor(best, off, 64);
    const int ok = __shfl_xor(bestk, off, 64);
    if (ov < best || (ov == best && ok < bestk)) {
      best = ov;
      bestk = ok;
    }
  }
  if (threadIdx.x == 0) {
    const int64_t at = (int64_t)group * P.part_stride + idx;
    const bool none = !(best < 1.7976931348623157e308);  // `totalCosts < val` never held: val and bestHireQty as initialised
    out_val[at] = none ? 1.7976931348623157e308 : best;
    out_idx[at] = none ? 0 : bestk;
  }
}

// The register-blocked form described at the top (pT0 = address of the j = 0 row inside the padded table).
template <int R, bool FUTURE>
__global__ __launch_bounds__(64) void staff_block_kernel(StaffParams P, const double* __restrict__ pT0,
                                                         const int32_t* __restrict__ row_len,
                                                         const double* __restrict__ v_next,
                                                         double* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                         int64_t lo, int64_t hi) {
  static_assert(R - 1 <= kStaffPadJ, "table padding");
  const int64_t tile = blockIdx.x / P.n_groups;
  const int group = (int)(blockIdx.x - tile * P.n_groups);
  const int64_t idx = lo + tile * 64 + threadIdx.x;
  if (idx >= hi) return;
  const int x = P.x_lo + (int)idx;
  const int a_end = min(P.n_actions, (group + 1) * P.group_actions);
  double best = 1.7976931348623157e308;
  int bestk = 0;
  for (int a0 = group * P.group_actions; a0 < a_end; a0 += R) {
    double fv[R], acc[R];
    int64_t poff[R];  // p of action r at step k: pT0[k * rows + poff[r]]
    int kmax = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int a = a0 + r;
      const int y = x + a;
      const int row = y >= P.n_rows - 1 ? P.n_rows - 1 : y;
      const int nj = a < P.n_actions ? row_len[row] : 0;
      kmax = max(kmax, nj > 0 ? nj + (R - 1 - r) : 0);
      const double fixHire = a > 0 ? P.K : 0.0;
      const double variHire = P.v * (double)a;
      fv[r] = fixHire + variHire;
      acc[r] = 0.0;
      poff[r] = (int64_t)row - (int64_t)(R - 1 - r) * P.n_rows;
    }
    const int ytop = x + a0 + R - 1;
    for (int k = 0; k < kmax; ++k) {
      const int n = ytop - k;  // nextStaffNum of every cell of this step (>= 0 while any of them is inside its row)
      const double salaryCost = P.salary * (double)n;
      const double penalty = n > P.min_staff ? 0.0 : P.pen * (double)(P.min_staff - n);
      double vn = 0.0;
      if constexpr (FUTURE) {
        int nn = n > P.nn_hi ? P.nn_hi : n;
        nn = nn < P.nn_lo ? P.nn_lo : nn;
        vn = v_next[nn - P.next_x_lo];
      }
      const double* pk = pT0 + (int64_t)k * P.n_rows;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double p = pk[poff[r]];
        const double imm = fv[r] + salaryCost + penalty;
        acc[r] += p * imm;
        if constexpr (FUTURE) acc[r] += p * vn;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (a0 + r < P.n_actions && acc[r] < best) {
        best = acc[r];
        bestk = a0 + r;
      }
  }
  const int64_t at = (int64_t)group * P.part_stride + idx;
  out_val[at] = best;
  out_idx[at] = bestk;
}

// Two ADJACENT states per lane.  The gather unit (TA) spends ~14 cycles on a 64-lane 8-byte gather and barely more on
// a 16-byte one, and at one probability gather per cell it is what bounds staff_block_kernel.  The states x and x + 1
// sit at the levels y and y + 1, whose table entries are neighbours in the transposed table, and they leave n and
// n + 1 behind, neighbours in V_{t+1}: one 16-byte gather serves both, picked apart where a clamp (the table's last
// row, the staff range) folds the two onto one entry.
typedef double staff_pair_u __attribute__((ext_vector_type(2), aligned(8)));

// One step k of the pair kernel: every cell of the step leaves n0 = ytop - k (state x) / n0 + 1 (state x + 1) behind.
// FAST: the WAVE has established that in this step no lane's staff number is clamped and none is at or below
// minStaffNum -- the selects that pick a clamped pair apart and the penalty arithmetic are skipped (the penalty is the
// constant +0.0 the reference adds there).  FOLD: some lane's level sits on the table's last row (its pair is folded).
// Addresses are a scalar base + a 32-bit byte offset (no 64-bit vector arithmetic): pk = table row k - (R - 1).
template <int R, bool FUTURE, bool FAST, bool FOLD>
__device__ __forceinline__ void staff_pair_step(const StaffParams& P, int n0, const char* __restrict__ pk,
                                                const char* __restrict__ vb, const uint32_t (&poff8)[R],
                                                const bool (&fold)[R], const double (&fv)[R], double (&acc)[2][R]) {
  const double sal0 = P.salary * (double)n0;
  const double sal1 = P.salary * (double)(n0 + 1);
  double pen0 = 0.0, pen1 = 0.0;
  if constexpr (!FAST) {
    pen0 = n0 > P.min_staff ? 0.0 : P.pen * (double)(P.min_staff - n0);
    pen1 = n0 + 1 > P.min_staff ? 0.0 : P.pen * (double)(P.min_staff - (n0 + 1));
  }
  double v0 = 0.0, v1 = 0.0;
  if constexpr (FUTURE) {
    if constexpr (FAST) {
      const staff_pair_u v = *reinterpret_cast<const staff_pair_u*>(vb + (uint32_t)((n0 - P.next_x_lo) * 8));
      v0 = v.x;
      v1 = v.y;
    } else {
      // {V[c], V[c + 1]}, c = clamp(n0, lo, hi - 1): where the bounds fold both onto one entry, select it
      int c = n0 > P.nn_hi - 1 ? P.nn_hi - 1 : n0;
      c = c < P.nn_lo ? P.nn_lo : c;
      const staff_pair_u v = *reinterpret_cast<const staff_pair_u*>(vb + (uint32_t)((c - P.next_x_lo) * 8));
      v0 = n0 > P.nn_hi - 1 ? v.y : v.x;  // n0 >= hi: the last entry
      v1 = n0 < P.nn_lo ? v.x : v.y;      // n0 + 1 <= lo: the first entry
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const staff_pair_u p = *reinterpret_cast<const staff_pair_u*>(pk + poff8[r]);
    double p0 = p.x;
    if constexpr (FOLD) p0 = fold[r] ? p.y : p.x;
    const double imm0 = fv[r] + sal0 + pen0;
    acc[0][r] += p0 * imm0;
    if constexpr (FUTURE) acc[0][r] += p0 * v0;
    const double imm1 = fv[r] + sal1 + pen1;
    acc[1][r] += p.y * imm1;
    if constexpr (FUTURE) acc[1][r] += p.y * v1;
  }
}

template <int R, bool FUTURE>
__global__ __launch_bounds__(64) void staff_pair_kernel(StaffParams P, const double* __restrict__ pT0,
                                                        const int32_t* __restrict__ row_len,
                                                        const double* __restrict__ v_next,
                                                        double* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                        int64_t lo, int64_t hi) {
  static_assert(R - 1 <= kStaffPadJ, "table padding");
  const int64_t tile = blockIdx.x / P.n_groups;
  const int group = (int)(blockIdx.x - tile * P.n_groups);
  const int64_t idx = lo + tile * 128 + 2 * threadIdx.x;  // this lane: states idx, idx + 1
  if (idx >= hi) return;
  const bool two = idx + 1 < hi;
  const int x = P.x_lo + (int)idx;
  const int x_first = __builtin_amdgcn_readfirstlane(x);  // lane 0 holds the tile's lowest staff number; the wave spans x_first .. x_first + 127
  const int a_end = min(P.n_actions, (group + 1) * P.group_actions);
  const int last_row = P.n_rows - 1;
  const int64_t row_bytes = (int64_t)P.n_rows * 8;
  const char* vb = reinterpret_cast<const char*>(v_next);
  double best[2] = {1.7976931348623157e308, 1.7976931348623157e308};
  int bestk[2] = {0, 0};
  for (int a0 = group * P.group_actions; a0 < a_end; a0 += R) {
    double fv[R], acc[2][R];
    uint32_t poff8[R];  // {p(y), p(y + 1)} of action r at step k: the pair at byte offset poff8[r] from table row k - (R - 1)
    bool fold[R];       // y >= last row: both levels use the last row (the pair's second entry)
    int kmax = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int a = a0 + r;
      const int y = x + a;
      const int row0 = y >= last_row ? last_row : y;
      const int row1 = y + 1 >= last_row ? last_row : y + 1;
      const int nj = a < P.n_actions ? max(row_len[row0], row_len[row1]) : 0;  // (zeros beyond the shorter row)
      kmax = max(kmax, nj > 0 ? nj + (R - 1 - r) : 0);
      const double fixHire = a > 0 ? P.K : 0.0;
      const double variHire = P.v * (double)a;
      fv[r] = fixHire + variHire;
      acc[0][r] = 0.0;
      acc[1][r] = 0.0;
      fold[r] = y >= last_row;
      const int c = y >= last_row ? last_row - 1 : y;  // rows >= 2 (the launcher checks)
      poff8[r] = (uint32_t)((c + r * P.n_rows) * 8);   // action r is at j = k - (R-1-r): row (k - (R-1)) + r
    }
    const int ytop = x + a0 + R - 1;
    const bool any_fold = __ballot(fold[R - 1]) != 0;  // (the level grows with r: r = R - 1 folds first)
    // Steps in which NO lane of the wave meets a clamp of the staff number or the penalty branch -- wave-uniform bounds:
    // at step k the wave's staff numbers n0 span [nlo, nlo + 126] (n1 = n0 + 1), nlo = x_first + a0 + R - 1 - k.
    //   top clamp free:    nlo + 126 <= nn_hi - 1  <=>  k >= ks
    //   bottom clamp free: nlo >= nn_lo,  penalty free: nlo > min_staff  <=>  k < ke
    const int ntop = x_first + a0 + R - 1;
    int ks = ntop + 127 - P.nn_hi;
    ks = ks < 0 ? 0 : ks;
    if constexpr (!FUTURE) ks = 0;  // (no V read in the last period: only the penalty matters)
    int ke = ntop - (FUTURE ? max(P.nn_lo - 1, P.min_staff) : P.min_staff);  // first step with nlo <= that bound
    ke = ke < ks ? ks : ke;
    const char* pk0 = reinterpret_cast<const char*>(pT0) - (int64_t)(R - 1) * row_bytes;
    int k = 0;
    for (const int e = min(kmax, ks); k < e; ++k)
      staff_pair_step<R, FUTURE, false, true>(P, ytop - k, pk0 + (int64_t)k * row_bytes, vb, poff8, fold, fv, acc);
    if (any_fold) {
      for (const int e = min(kmax, ke); k < e; ++k)
        staff_pair_step<R, FUTURE, true, true>(P, ytop - k, pk0 + (int64_t)k * row_bytes, vb, poff8, fold, fv, acc);
    } else {
      for (const int e = min(kmax, ke); k < e; ++k)
        staff_pair_step<R, FUTURE, true, false>(P, ytop - k, pk0 + (int64_t)k * row_bytes, vb, poff8, fold, fv, acc);
    }
    for (; k < kmax; ++k)
      staff_pair_step<R, FUTURE, false, true>(P, ytop - k, pk0 + (int64_t)k * row_bytes, vb, poff8, fold, fv, acc);
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (a0 + r < P.n_actions) {
        if (acc[0][r] < best[0]) {
          best[0] = acc[0][r];
          bestk[0] = a0 + r;
        }
        if (acc[1][r] < best[1]) {
          best[1] = acc[1][r];
          bestk[1] = a0 + r;
        }
      }
  }
  const int64_t at = (int64_t)group * P.part_stride + idx;
  out_val[at] = best[0];
  out_idx[at] = bestk[0];
  if (two) {
    out_val[at + 1] = best[1];
    out_idx[at + 1] = bestk[1];
  }
}

// ---------------------------------------------------------------------------------------------
// Window form (large staff ranges).  The pair kernel fetches one probability per cell through the vector L1 (8 B per
// cell, plus the V pairs): that path, not the arithmetic, is what bounds it (TA 74 % busy at 15 VALU instructions per
// cell).  But everything a cell reads depends on TWO integers only, the level y = x + a and the realisation j: the
// probability p(y, j), and -- through n = y - j -- the staff left, its salary, its penalty and V_{t+1}[n].  A lane that
// owns S ADJACENT states and carries R consecutive actions has R S cells per step but only R + S - 1 distinct levels
// (state s, action r: level Y + r + s), so per step it needs R + S - 1 probabilities, not R S; the products p V are
// formed once per level; and the cells (s, r, j) and (s + 1, r, j + 1) leave the same n behind with the same action, so
// the immediate cost (fv[r] + salary n) + penalty(n) -- the reference's two additions, on identical operands -- is
// formed once per (r, n) and handed down the lane's states one step at a time, exactly as the F1 window kernel does
// (sdp_window.hpp).  Executed fp64 operations per cell: 3 + 2/S + (R + S - 1)/(R S) = 3.94 for R = S = 4, against 6.
// The R + S - 1 entries {V, salary, penalty} over n slide by ONE new entry per step (registers, rotating slots); the
// probabilities of a step are one contiguous piece of row j of the transposed table (64 S + R + S - 2 levels), which
// the wave loads with 16-byte loads into LDS -- stored by residue class of the level modulo S, so that the lanes'
// reads, S levels apart, fall on consecutive slots (no bank conflicts) -- one step ahead of its use.  Every value that
// enters an accumulator is the reference's, in the reference's order (j ascending; zero-probability steps beyond a
// row's length add +0.0).
// ---------------------------------------------------------------------------------------------
// Three waves per SIMD: the (4, 4) block takes 207 VGPRs left alone (two waves); capped at 168 it spills 31 dwords to scratch and
// runs WorkforceTesting.main's instance 3 % faster (3.86 against 3.75e12 cells/s, two runs each, same box) -- the launches of
// that instance are short (0.2-1.4 ms, 28 tiles of states x action groups), which is what holds it, not the registers.
// -DSDP_STAFF_WIN_ATTR= (empty) rebuilds the two-wave form.
#ifndef SDP_STAFF_WIN_ATTR
#define SDP_STAFF_WIN_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
template <int R, int S, bool FUTURE>
__global__ __launch_bounds__(64) SDP_STAFF_WIN_ATTR void staff_window_kernel(StaffParams P, const double* __restrict__ pT0,
                                                          const int32_t* __restrict__ row_len,
                                                          const double* __restrict__ v_next,
                                                          double* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                          int64_t lo, int64_t hi) {
  constexpr int NW = R + S - 1;          // levels per lane and step = window entries
  constexpr int TS = 64 * S;             // states per tile
  constexpr int NLEV = TS + NW - 1;      // levels per step: Y0 .. Y0 + NLEV - 1
  constexpr int NPAIR = (NLEV + 1) / 2;  // ... loaded as pairs of adjacent levels
  constexpr int NPC = (NPAIR + 63) / 64;
  constexpr int H = (NLEV + S - 1) / S + 7;  // slots per residue class (odd multiple-free padding)
  static_assert(NW <= kStaffPadJ, "table padding behind the last row of realisations");
  __shared__ __attribute__((aligned(16))) double s_p[2][S * H];
  const int lane = threadIdx.x;
  const int64_t tile = blockIdx.x / P.n_groups;
  const int group = (int)(blockIdx.x - tile * P.n_groups);
  const int64_t idx0 = lo + tile * TS + (int64_t)S * lane;  // this lane: states idx0 .. idx0 + S - 1
  const int x0 = P.x_lo + (int)(lo + tile * TS);             // the tile's lowest staff number
  const int a_end = min(P.n_actions, (group + 1) * P.group_actions);
  const int last_row = P.n_rows - 1;
  const int64_t row_bytes = (int64_t)P.n_rows * 8;
  const char* vb = reinterpret_cast<const char*>(v_next);
  const char* pb = reinterpret_cast<const char*>(pT0);
  double best[S];
  int bestk[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    best[s] = 1.7976931348623157e308;
    bestk[s] = 0;
  }
  // staging: pair m = 64 c + lane holds the levels Y0 + 2 m, Y0 + 2 m + 1; level q goes to slot (q mod S) H + q / S
  // (lanes past the last pair of the last piece repeat it: the same values to the same slots, and no divergent branch in the loop)
  int w_slot[NPC], w_pair[NPC];
#pragma unroll
  for (int c = 0; c < NPC; ++c) {
    w_pair[c] = 64 * c + lane < NPAIR ? 64 * c + lane : NPAIR - 1;
    const int q0 = 2 * w_pair[c];
    w_slot[c] = (q0 % S) * H + q0 / S;  // (S even: the pair's second level sits one class further, same slot)
  }
  auto salary_of = [&](int n) { return P.salary * (double)n; };
  auto penalty_of = [&](int n) { return n > P.min_staff ? 0.0 : P.pen * (double)(P.min_staff - n); };
  auto v_of = [&](int n) {
    int nn = n > P.nn_hi ? P.nn_hi : n;
    nn = nn < P.nn_lo ? P.nn_lo : nn;
    return *reinterpret_cast<const double*>(vb + (uint32_t)((nn - P.next_x_lo) * 8));
  };
  for (int a0 = group * P.group_actions; a0 < a_end; a0 += R) {
    const int Y0 = x0 + a0;        // lowest level of the block
    const int Yl = Y0 + S * lane;  // this lane's lowest level
    // steps: the longest row among the block's levels, rounded up to whole trips of NW steps (zero rows behind the table)
    int kmax = 0;
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int y = Yl + u;
      kmax = max(kmax, row_len[y >= last_row ? last_row : y]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, 64));
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    const int ksteps = (kmax + NW - 1) / NW * NW;
    double fv[R], acc[S][R], immc[S][R];
    double vw[NW], salw[NW], penw[NW];
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      vw[u] = FUTURE ? v_of(Yl + u) : 0.0;
      salw[u] = salary_of(Yl + u);
      penw[u] = penalty_of(Yl + u);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int a = a0 + r;
      fv[r] = (a > 0 ? P.K : 0.0) + P.v * (double)a;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        acc[s][r] = 0.0;
        immc[s][r] = fv[r] + salw[r + s] + penw[r + s];  // (s = 0 unused)
      }
    }
    // one row piece: pairs {p(y), p(y + 1)} of row j; a level on or beyond the table's last row reads the last row
    staff_pair_u tmp[NPC];
    const bool folds = Y0 + NLEV >= last_row;
    auto row_load = [&](int j) {
      const char* pk = pb + (int64_t)j * row_bytes;
#pragma unroll
      for (int c = 0; c < NPC; ++c) {
        const int e0 = Y0 + 2 * w_pair[c];
        const int cc = e0 >= last_row ? last_row - 1 : e0;
        staff_pair_u t = *reinterpret_cast<const staff_pair_u*>(pk + (uint32_t)(cc * 8));
        if (folds) t.x = e0 >= last_row ? t.y : t.x;  // (wave-uniform: only blocks that reach the table's last row)
        tmp[c] = t;
      }
    };
    auto row_store = [&](int buf) {
#pragma unroll
      for (int c = 0; c < NPC; ++c) {
        s_p[buf][w_slot[c]] = tmp[c].x;
        s_p[buf][w_slot[c] + H] = tmp[c].y;
      }
    };
    // UNI: every level of the block sits on or beyond the table's last row (pmfs[t][min(y, rows - 1)], :93-94) -- all of a run's
    // staff numbers from rows - 1 up, six tiles of seven in the last period of WorkforceTesting.main's instance.  Every lane of
    // every level then reads the SAME probability p(rows - 1, j): one scalar load a step, a step ahead -- no row piece, no LDS
    // staging, no barriers, no fold selects; the cells' arithmetic is the same, on the same values.
    auto run_steps = [&](auto uni_tag) {
      constexpr bool UNI = decltype(uni_tag)::value;
      const char* p_last = pb + (int64_t)last_row * 8;
      [[maybe_unused]] double p_ahead = 0.0;
      if constexpr (UNI) {
        p_ahead = *reinterpret_cast<const double*>(p_last);
      } else {
        __builtin_amdgcn_wave_barrier();  // (the previous block's reads of the buffers are done)
        row_load(0);
        row_store(0);
        __builtin_amdgcn_wave_barrier();
      }
      int cur = 0;
      for (int jb = 0; jb < ksteps; jb += NW) {
#pragma unroll
        for (int t = 0; t < NW; ++t) {
          const int j = jb + t;
          [[maybe_unused]] double p_now = 0.0;
          if constexpr (UNI) {
            p_now = p_ahead;
            p_ahead = *reinterpret_cast<const double*>(p_last + (int64_t)(j + 1) * row_bytes);
          } else {
            row_load(j + 1);
          }
          double vnew = 0.0;
          if constexpr (FUTURE) vnew = v_of(Yl - (j + 1));
          // by level (anti-diagonal u = r + s): one probability read and one product p V per level, used by its cells at once
          double imm0[R];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int w0 = (r - t + NW) % NW;
            imm0[r] = fv[r] + salw[w0] + penw[w0];
          }
#pragma unroll
          for (int u = 0; u < NW; ++u) {
            double pu;
            if constexpr (UNI) pu = p_now; else pu = s_p[cur][(u % S) * H + lane + u / S];
            double pvu = 0.0;
            if constexpr (FUTURE) pvu = pu * vw[(u - t + NW) % NW];
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) {
              const int r = u - s2;
              if (r >= 0 && r < R) {
                acc[s2][r] += pu * (s2 == 0 ? imm0[r] : immc[s2][r]);
                if constexpr (FUTURE) acc[s2][r] += pvu;
              }
            }
          }
#pragma unroll
          for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int s2 = S - 1; s2 > 1; --s2) immc[s2][r] = immc[s2 - 1][r];
            if constexpr (S > 1) immc[1][r] = imm0[r];
          }
          // slide: the entry of level Yl at step j + 1 replaces the one of level Yl + NW - 1 just used
          const int nn = Yl - (j + 1);
          vw[(NW - 1 - t) % NW] = vnew;
          salw[(NW - 1 - t) % NW] = salary_of(nn);
          penw[(NW - 1 - t) % NW] = penalty_of(nn);
          if constexpr (!UNI) {
            __builtin_amdgcn_wave_barrier();
            row_store(cur ^ 1);
            __builtin_amdgcn_wave_barrier();
            cur ^= 1;
          }
          __builtin_amdgcn_sched_barrier(0);  // (one step at a time: left alone, the scheduler hoists the loads of all NW steps -- 330 VGPRs)
        }
      }
    };
    // (one kernel, two branches: as two launches over the same grid -- each form with its own registers, the staged one without
    // the 58 extra spilled dwords it carries here -- the two partially filled launches ran one after the other and the sweep took
    // 6.5 ms against 4.4)
    if (Y0 >= last_row && P.uni_rows)
      run_steps(std::true_type{});
    else
      run_steps(std::false_type{});
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (a0 + r < P.n_actions) {
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (acc[s][r] < best[s]) {
            best[s] = acc[s][r];
            bestk[s] = a0 + r;
          }
      }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int64_t idx = idx0 + s;
    if (idx < hi) {
      const int64_t at = (int64_t)group * P.part_stride + idx;
      out_val[at] = best[s];
      out_idx[at] = bestk[s];
    }
  }
}

// groups in ascending action order, strict compare: the first best wins (StaffRecursion.java:110-113)
__global__ __launch_bounds__(256) void combine_staff_kernel(const double* __restrict__ part_val,
                                                            const int32_t* __restrict__ part_idx, int n_groups,
                                                            int64_t part_stride, double* __restrict__ v_cur,
                                                            int32_t* __restrict__ pol, int64_t lo, int64_t hi) {
  const int64_t idx = lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= hi) return;
  // The scan keeps the winning GROUP and reads that group's action afterwards: with the action read inside the compare, every
  // step was a dependent load (251 groups of WorkforceTesting.main's instance: 61 us a period, 8 % of its sweep); the values
  // alone are independent reads, eight in flight per trip.
  double best = 1.7976931348623157e308;
  int bestg = -1;
  constexpr int U = 8;
  int g = 0;
  for (; g + U <= n_groups; g += U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = part_val[(int64_t)(g + u) * part_stride + idx];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (v[u] < best) {
        best = v[u];
        bestg = g + u;
      }
  }
  for (; g < n_groups; ++g) {
    const double v = part_val[(int64_t)g * part_stride + idx];
    if (v < best) {
      best = v;
      bestg = g;
    }
  }
  v_cur[idx] = best;
  pol[idx] = bestg >= 0 ? part_idx[(int64_t)bestg * part_stride + idx] : 0;  // (no value below Double.MAX_VALUE: bestHireQty = 0)
}

}  // namespace sdp

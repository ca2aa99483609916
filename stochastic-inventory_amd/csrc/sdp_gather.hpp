// sdp_gather.hpp -- generic period kernel: per-cell functor + gather from V_{t+1}.
//
// One launch = one period t = all states of this rank's slab.  Work item layout (wave64):
//   a 256-thread workgroup owns SX consecutive states (the fastest state axis, so that the
//   V_{t+1} gathers of neighbouring lanes fall in the same 128-B lines) and AS = 256 / SX
//   action slots; lane (sx, as) walks actions as, as+AS, ... and, per action, the demand
//   realisations j = 0..D-1 SERIALLY in the reference's order (Recursion.java:138-144) -- the
//   d-sum is never split across lanes, that would change the rounding of Q(s,a).
//   arg-opt over actions: in-lane strict compare (ascending), then a wave-shuffle
//   (value, index) reduction across the action slots of a state, then across the four waves
//   through LDS.  The demand PMF tile {d_j, p_j} is staged once per workgroup in LDS.
//
// Roofline: every cell reads one fp64 of V_{t+1} (8 B algorithmic); the table (80 KB .. 800 MB)
// lives in L2 / Infinity Cache, so the kernel is bound by fp64 VALU issue and L2 gather rate,
// not by HBM -- see DESIGN.md.
#pragma once
#include "sdp_device.hpp"

namespace sdp {

// XCD-aware tile order: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so
// give each XCD one contiguous run of state tiles (its L2 then holds one slice of V_{t+1}
// plus a halo).  Pure performance; any placement is correct.
__device__ __forceinline__ int64_t xcd_tile(int64_t b, int64_t nb) {
  constexpr int64_t NX = 8;
  int64_t per = nb / NX;  // tiles per XCD in the evenly divisible part
  int64_t main = per * NX;
  if (b >= main) return b;  // ragged tail keeps its own order
  return (b % NX) * per + b / NX;
}

// Query mode evaluates arbitrary state tuples (sdpgpu_eval_states) instead of grid indices.
struct QueryStates {
  const double* x;
  const double* cash;
  const double* preq;
  const double* preq2;
};

template <int FAM, bool MAXDIR, int SX, bool QUERY>
__global__ __launch_bounds__(256) void gather_period_kernel(DevParams P, const double* __restrict__ v_next,
                                                            double* __restrict__ v_cur, int32_t* __restrict__ pol,
                                                            const double* __restrict__ pmf_d,
                                                            const double* __restrict__ pmf_p, int64_t lo, int64_t hi,
                                                            QueryStates q) {
  constexpr int AS = 256 / SX;          // action slots per workgroup
  constexpr int AS_WAVE = (SX >= 64) ? 1 : 64 / SX;  // action slots inside one wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2* s_pmf = reinterpret_cast<double2*>(smem);  // {d_j, p_j} pairs: x = demand, y = probability
  double* s_val = reinterpret_cast<double*>(smem + (size_t)P.n_demand * 16);
  int* s_k = reinterpret_cast<int*>(s_val + 4 * 64);

  const int tid = threadIdx.x;
  for (int j = tid; j < P.n_demand; j += 256) s_pmf[j] = make_double2(pmf_d[j], pmf_p[j]);
  __syncthreads();

  const int64_t nb = gridDim.x;
  const int64_t tile = xcd_tile(blockIdx.x, nb);
  const int sx = tid % SX;
  const int as = tid / SX;
  const int64_t idx = lo + tile * SX + sx;
  const bool live = idx < hi;

  StateT s;
  if constexpr (QUERY) {
    s.x = live ? q.x[idx] : 0.0;
    s.cash = (live && q.cash) ? q.cash[idx] : 0.0;
    s.preq = (live && q.preq) ? q.preq[idx] : 0.0;
    s.preq2 = (live && q.preq2) ? q.preq2[idx] : 0.0;
    xr_state<FAM>(P, s, false);  // (x, R) family: the queried "cash" is R
  } else {
    decode_state<FAM>(P, live ? idx : lo, s);
  }
  int nA = live ? n_actions<FAM>(P, s) : 0;
  if constexpr (!QUERY) {
    if (P.counts && live) nA = P.counts[idx];  // the caller's own list lengths (sdpgpu_set_action_counts)
  }

  double best = MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;  // +-Double.MAX_VALUE
  int bestk = 0;                                                           // bestOrderQty = 0
  const int nD = P.n_demand;
  for (int k = as; k < nA; k += AS) {
    ActionCtx c;
    action_setup<FAM>(P, s, k, c);
    double acc = 0.0;
    if constexpr (FAM == FAM_SURVIVAL) {
      // RiskRecursion.java:78-98: probability of ending with non-negative cash
      if (P.is_last) {
        for (int j = 0; j < nD; ++j) {
          double2 dp = s_pmf[j];
          int64_t ni;
          double fin = s.cash + cell<FAM, 1>(P, s, c, dp.x, ni);
          acc += dp.y * (fin >= 0 ? 1.0 : 0.0);
        }
      } else {
        for (int j = 0; j < nD; ++j) {
          double2 dp = s_pmf[j];
          int64_t ni = 0;
          (void)cell<FAM, 0>(P, s, c, dp.x, ni);
          acc += (dp.y * P.gamma) * (ni < 0 ? 0.0 : v_next[ni]);
        }
      }
    } else if (P.is_last) {
      for (int j = 0; j < nD; ++j) {
        double2 dp = s_pmf[j];
        int64_t ni;
        double imm = cell<FAM, 1>(P, s, c, dp.x, ni);
        acc += dp.y * imm;
      }
    } else {
      for (int j = 0; j < nD; ++j) {
        double2 dp = s_pmf[j];
        int64_t ni = 0;
        double imm = cell<FAM, 0>(P, s, c, dp.x, ni);
        acc += dp.y * imm;
        acc += (dp.y * P.gamma) * v_next[ni];  // CashRecursion.java:120: p * gamma * V
      }
    }
    if (MAXDIR ? (acc > best) : (acc < best)) {
      best = acc;
      bestk = k;
    }
  }

  // wave-shuffle arg-opt across the action slots that share a state inside the wave
  if constexpr (AS_WAVE > 1) {
#pragma unroll
    for (int off = SX; off < 64; off <<= 1) {
      double ov = __shfl_xor(best, off, 64);
      int ok = __shfl_xor(bestk, off, 64);
      if (better<MAXDIR>(ov, ok, best, bestk)) {
        best = ov;
        bestk = ok;
      }
    }
  }
  // across waves through LDS
  const int wave = tid >> 6;
  const int lane = tid & 63;
  if constexpr (SX >= 64) {
    s_val[wave * 64 + lane] = best;
    s_k[wave * 64 + lane] = bestk;
  } else {
    if (lane < SX) {
      s_val[wave * 64 + lane] = best;
      s_k[wave * 64 + lane] = bestk;
    }
  }
  __syncthreads();
  if (tid < SX && live) {
    double bv = s_val[tid];
    int bk = s_k[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {  // all four waves hold action slots of the same states
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

// Forward reachable set: for every marked state of period t, mark the successor of every feasible
// (action, demand) pair in period t+1 -- the key set Recursion.java:90's memo map would hold.
// Byte stores of the same value race benignly.  QUERY mode seeds from explicit state tuples
// (the period-1 initial state, which need not be a grid point).
template <int FAM, bool QUERY>
__global__ __launch_bounds__(256) void reach_kernel(DevParams P, const uint8_t* __restrict__ mask_cur,
                                                    uint8_t* __restrict__ mask_next,
                                                    const double* __restrict__ pmf_d, int64_t n, QueryStates q) {
  const int sx = threadIdx.x & 63;
  const int as = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 64 + sx;
  if (idx >= n) return;
  StateT s;
  if constexpr (QUERY) {
    s.x = q.x[idx];
    s.cash = q.cash ? q.cash[idx] : 0.0;
    s.preq = q.preq ? q.preq[idx] : 0.0;
    s.preq2 = q.preq2 ? q.preq2[idx] : 0.0;
    xr_state<FAM>(P, s, false);
  } else {
    if (!mask_cur[idx]) return;
    decode_state<FAM>(P, idx, s);
  }
  int nA = n_actions<FAM>(P, s);
  if constexpr (!QUERY) {
    if (P.counts) nA = P.counts[idx];
  }
  for (int k = as; k < nA; k += 4) {
    ActionCtx c;
    action_setup<FAM>(P, s, k, c);
    for (int j = 0; j < P.n_demand; ++j) {
      int64_t ni = 0;
      (void)cell<FAM>(P, s, c, pmf_d[j], ni);
      if (ni >= 0) mask_next[ni] = 1;  // (negative only in the survival family: bankrupt successors are not visited)
    }
  }
}

// Forward rollout of the arg-opt policy along sampled demand paths (Simulation.java:59-69): one path per
// lane; per period a policy-table gather, the family's immediate value and transition.  The running
// sum is accumulated in period order exactly as the reference's `sum += ...` does.
struct SimPeriod {
  DevParams P;
  int64_t pol_off;  // element offset of the period's policy row (indexed by flat state index)
  int64_t n_states;
};

template <int FAM>
__global__ __launch_bounds__(256) void simulate_kernel(const SimPeriod* __restrict__ per, int T,
                                                       const int32_t* __restrict__ pol, const double* __restrict__ demand,
                                                       const double* __restrict__ disc, int64_t n, int64_t idx0,
                                                       StateT ini, int first_k, double* __restrict__ out_sum,
                                                       uint8_t* __restrict__ out_valid) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double sum = 0.0;
  int64_t idx = idx0;
  StateT s = ini;
  bool valid = true;
  bool lost = false;
  for (int t = 0; t < T && valid; ++t) {
    const DevParams& P = per[t].P;
    int k;
    if (t == 0 && idx0 < 0) {
      k = first_k;  // off-grid initial state: its action comes from sdpgpu_eval_states
    } else {
      decode_state<FAM>(P, idx, s);
      k = pol[per[t].pol_off + idx];
    }
    if constexpr (FAM == FAM_SURVIVAL) {
      // RiskSimulation.simulateLostSale (RiskSimulation.java:213-234): sum = 1 once the path has held negative
      // cash, bit 1 of the flags once a demand was lost; the walk itself continues through bankrupt states
      if (s.cash < 0) k = 0;
      ActionCtx c;
      action_setup<FAM>(P, s, k, c);
      const double d = demand[i * T + t];
      if (c.base < d) lost = true;
      int64_t ni = 0;
      const double imm = cell<FAM>(P, s, c, d, ni);
      if (s.cash + imm < 0) sum = 1.0;
      if (!P.is_last) {  // cell() marks a bankrupt successor with -1: the rollout needs its index all the same
        double ninv = jmax(0.0, c.base - d);
        ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
        ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
        idx = (int64_t)inv_index(P, ninv) * P.next.nc + cash_index(P, s.cash + imm);
      }
      continue;
    }
    ActionCtx c;
    action_setup<FAM>(P, s, k, c);
    const double d = demand[i * T + t];
    int64_t ni = 0;
    const double imm = cell<FAM>(P, s, c, d, ni);
    sum += disc[t] * imm;
    if (!P.is_last) {
      if (!P.clamp_inventory) {  // unclamped boxes cover the PMF support only
        const double level = c.base - d;
        const double hi = P.next.x_lo + (double)(P.next.nx - 1) * P.step;
        if (level < P.next.x_lo || level > hi) valid = false;
      }
      idx = ni;
    }
  }
  out_sum[i] = sum;
  out_valid[i] = (valid ? 1 : 0) | (lost ? 2 : 0);
}

}  // namespace sdp

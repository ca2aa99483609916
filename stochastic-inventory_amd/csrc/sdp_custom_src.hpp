// sdp_custom_src.hpp -- device source of the USER-DEFINED functor path, compiled at sdpgpu_create_custom()
// time with hipRTC for gfx950 (-ffp-contract=off, like the rest of the library).
//
// The reference injects a problem as three Java lambdas (Recursion.java:49-52: feasible actions,
// StateTransition.java:20-22, ImmediateValue.java:23-25).  The built-in families cover the in-scope drivers;
// any other driver's lambdas are handed over as three HIP device functions (see include/sdpgpu.h,
// sdpgpu_create_custom) and this engine source is compiled around them: same loop, same accumulation
// order, same strict-compare arg-opt as sdp_gather.hpp.
//
// Three pieces of text: kCustomPrelude (what the user's functions may use), the user's source, kCustomEngine.
#pragma once

namespace sdp {

// NOTE: struct CustomParams below (host) and `struct CParams` in kCustomEngine (device) must stay identical;
// both sides static_assert the size.
struct CustomGrid {
  double x_lo;
  long long nx, nc, nq, k_lo;
};
struct CustomParams {
  int has_cash, has_preq, maxdir, is_last;
  int n_demand, survival, cash_int_div, period;
  int T, pad0;
  double step, inv_step, gamma, round_mult, round_div;
  CustomGrid cur, next;
  const double* user;
};
static_assert(sizeof(CustomParams) == 168, "CustomParams layout is mirrored in kCustomEngine");

static const char* const kCustomPrelude = R"SDPSRC(
typedef long long sdp_i64;
// what the three user functions receive besides the state tuple
struct sdp_ctx {
  int period;            // 1-based period of the state (State.getPeriod())
  int T;                 // horizon, pmf.length
  double step;           // stepSize
  const double* params;  // the doubles given to sdpgpu_create_custom (the constants the lambdas close over)
};
// java.lang.Math.max / min / round for the operands that occur here (round: nearest, ties toward +infinity,
// returned as an integer-valued double)
__device__ inline double sdp_max(double a, double b) { return fmax(a, b); }
__device__ inline double sdp_min(double a, double b) { return fmin(a, b); }
__device__ inline double sdp_round(double x) { double f = floor(x); return (x - f >= 0.5) ? f + 1.0 : f; }
__device__ inline double sdp_trunc(double x) { return trunc(x); }  // (int) / (long) casts, long division
// Java's `long / int` on an integer-valued double, e.g. `Math.round(cash * 10) / 10` (CashOverdraft.java:116): truncating integer
// division, done in integers (|a| < 2^31: cash keys fit 32 bits, checked at create) -- with a literal divisor a few integer
// instructions, where sdp_trunc(a / b) costs a correctly rounded fp64 division per cell
__device__ inline double sdp_ldiv(double a, int b) { return (double)((int)a / b); }
#line 1 "user_functor"
)SDPSRC";

static const char* const kCustomEngine = R"SDPSRC(
#line 1 "sdp_custom_engine"
#ifndef SDP_NP
#define SDP_NP 1  // number of user parameters (-DSDP_NP=n at compile time)
#endif
struct CGrid { double x_lo; sdp_i64 nx, nc, nq, k_lo; };
struct CParams {
  int has_cash, has_preq, maxdir, is_last;
  int n_demand, survival, cash_int_div, period;
  int T, pad0;
  double step, inv_step, gamma, round_mult, round_div;
  CGrid cur, next;
  const double* user;
};
static_assert(sizeof(CParams) == 168, "CParams layout is mirrored in sdp_custom_src.hpp");

// The state shape, the loop variant and the direction are fixed when the handle is created, so sdpgpu_create_custom
// passes them as -D constants and the branches on them fold away; without the defines they are read from P.
#ifndef SDP_HAS_CASH
#define SDP_HAS_CASH P.has_cash
#define SDP_HAS_PREQ P.has_preq
#define SDP_SURVIVAL P.survival
#define SDP_MAXDIR P.maxdir
#define SDP_CASH_INT_DIV P.cash_int_div
#endif

// SDP_BAKE (sdpgpu_create_custom, clamped grids): the grid every period shares, the step and the user's constants as
// compile-time constants -- the index arithmetic of c_next_index folds (one axis: no 64-bit products), the user's formulas are
// specialised to their constants.
#ifdef SDP_BAKE
#define C_STEP (SDP_B_STEP)
#define C_INV_STEP (SDP_B_INV_STEP)
#define C_NEXT_XLO (SDP_B_XLO)
#define C_NEXT_NX ((sdp_i64)(SDP_B_NX))
#define C_NEXT_NC ((sdp_i64)(SDP_B_NC))
#define C_NEXT_NQ ((sdp_i64)(SDP_B_NQ))
#define C_NEXT_KLO ((sdp_i64)(SDP_B_KLO))
#else
#define C_STEP P.step
#define C_INV_STEP P.inv_step
#define C_NEXT_XLO P.next.x_lo
#define C_NEXT_NX P.next.nx
#define C_NEXT_NC P.next.nc
#define C_NEXT_NQ P.next.nq
#define C_NEXT_KLO P.next.k_lo
#endif

struct CState { double x, cash, preq; };

#ifdef SDP_SHAPE_LEVEL
// The user declared the LEVEL SHAPE: immediate value = sdp_action_cost(action) + sdp_level_cost(x + action - demand) -- one
// addition of two separately formed terms -- every order quantity 0 .. maxOrderQuantity on offer in every state, and the
// (clamped, as the descriptor says) level as the next inventory.  The three lambdas of the generic loop are these:
__device__ inline int sdp_feasible_count(const sdp_ctx& c, double x, double cash, double preq) { return SDP_LEVEL_NACT; }
__device__ inline double sdp_immediate(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand) {
  return sdp_action_cost(c, action) + sdp_level_cost(c, x + action - randomDemand);
}
__device__ inline void sdp_transition(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand,
                                      double& nx, double& ncash, double& npreq) {
  double n = x + action - randomDemand;
  if (SDP_LEVEL_CLAMP) {  // CLSP.java:257-258: upper clamp, then lower clamp
    n = n > SDP_LEVEL_MAX ? SDP_LEVEL_MAX : n;
    n = n < SDP_LEVEL_MIN ? SDP_LEVEL_MIN : n;
  }
  nx = n;
  ncash = 0;
  npreq = 0;
}
#endif

__device__ inline void c_decode(const CParams& P, sdp_i64 idx, CState& s) {
  sdp_i64 ic = idx % P.cur.nc;
  sdp_i64 r = idx / P.cur.nc;
  sdp_i64 ix = r % P.cur.nx;
  sdp_i64 iq = r / P.cur.nx;
  s.x = P.cur.x_lo + (double)ix * P.step;
  double k = (double)(P.cur.k_lo + ic);
  s.cash = SDP_HAS_CASH ? (SDP_CASH_INT_DIV ? k : k / P.round_div) : 0.0;
  s.preq = SDP_HAS_PREQ ? (double)iq * P.step : 0.0;
}

// Flat index of the state the user's transition returned, which must be a grid point of period + 1 (the Java
// lambda clamps and rounds itself).  Anything else sets `bad`; the index read is then some point of the grid (every axis
// index is clamped into its range: a safe address, no branch) and the solve fails with the flag.  A coordinate is a grid
// point exactly when its clamped integer index converts back to it -- one test for "integer" and "in range" together.
// (`bad` is a register: an atomic inside the demand loop would make the compiler reload the user's parameters after every cell.)
__device__ inline int c_clamp_index(int i, int n) {  // (max, then min: one v_med3_i32)
  const int t = i > 0 ? i : 0;
  return t < n - 1 ? t : n - 1;
}
__device__ inline sdp_i64 c_next_index(const CParams& P, double nx, double ncash, double npreq, bool& bad) {
  // (32-bit conversions: every axis is shorter than 2^31 points and cash keys fit 32 bits, checked at create;
  // a value out of int range saturates and fails the test)
  const double fx = (nx - C_NEXT_XLO) * C_INV_STEP;
  const int ix = c_clamp_index((int)fx, (int)C_NEXT_NX);
  bool ok = (double)ix == fx;
  int ic = 0, iq = 0;
  if (SDP_HAS_CASH) {
    int k;
    double back;
    if (SDP_CASH_INT_DIV) {
      k = c_clamp_index((int)((unsigned)(int)ncash - (unsigned)(int)C_NEXT_KLO), (int)C_NEXT_NC) + (int)C_NEXT_KLO;
      back = (double)k;
    } else {
      k = c_clamp_index((int)((unsigned)(int)sdp_round(ncash * P.round_mult) - (unsigned)(int)C_NEXT_KLO), (int)C_NEXT_NC) + (int)C_NEXT_KLO;
      back = (double)k / P.round_div;
    }
    ic = k - (int)C_NEXT_KLO;
    ok = ok && back == ncash;
  }
  if (SDP_HAS_PREQ) {
    const double fq = npreq * C_INV_STEP;
    iq = c_clamp_index((int)fq, (int)C_NEXT_NQ);
    ok = ok && (double)iq == fq;
  }
  bad = bad || !ok;
  return ((sdp_i64)iq * C_NEXT_NX + ix) * C_NEXT_NC + ic;
}

__device__ inline bool c_better(bool maxdir, double v2, int k2, double v, int k) {
  return maxdir ? (v2 > v || (v2 == v && k2 < k)) : (v2 < v || (v2 == v && k2 < k));
}

// One period: 256 threads = SX consecutive states x 256/SX action slots (SX = 64, 16 or 4: small grids get more
// action slots per state so that the launch still fills the chip); a lane walks its actions and, per action,
// the demand index serially in the reference's order (Recursion.java:138-144).  q* != nullptr: evaluate the
// given state tuples instead of grid states (getExpectedValue on an off-grid state).
template <int SX, bool LAST>
__device__ inline void custom_period_body(
    const CParams& P, const double* __restrict__ v_next, double* __restrict__ v_cur, int* __restrict__ pol,
    const double* __restrict__ pmf_d, const double* __restrict__ pmf_p, sdp_i64 lo, sdp_i64 hi,
    const double* __restrict__ qx, const double* __restrict__ qcash, const double* __restrict__ qpreq,
    unsigned long long* __restrict__ cells, int* __restrict__ err) {
  constexpr int AS = 256 / SX;  // action slots
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2* s_pmf = reinterpret_cast<double2*>(smem);
  double* s_val = reinterpret_cast<double*>(smem + (size_t)P.n_demand * 16);
  int* s_k = reinterpret_cast<int*>(s_val + 4 * SX);
  const int tid = threadIdx.x;
  for (int j = tid; j < P.n_demand; j += 256) s_pmf[j] = make_double2(pmf_d[j], pmf_p[j]);
  __syncthreads();

  const int sx = tid % SX;
  const int as = tid / SX;
  const sdp_i64 idx = lo + (sdp_i64)blockIdx.x * SX + sx;
  const bool live = idx < hi;
  CState s;
  if (qx) {
    s.x = live ? qx[idx] : 0.0;
    s.cash = (live && qcash) ? qcash[idx] : 0.0;
    s.preq = (live && qpreq) ? qpreq[idx] : 0.0;
  } else {
    c_decode(P, live ? idx : lo, s);
  }
  // The user's constants are copied into a private array first: constant indices then become registers, a
  // period-dependent index one hoisted load -- read through the global pointer, every use inside a branch of the
  // user's code would be a scalar load with a full wait in the demand loop (the compiler may not speculate it).
#ifdef SDP_BAKE
  const double prm[SDP_NP] = SDP_B_PARAMS;
#else
  double prm[SDP_NP];
#pragma unroll
  for (int i = 0; i < SDP_NP; ++i) prm[i] = P.user[i];
#endif
  sdp_ctx U;
  U.period = P.period;
  U.T = P.T;
  U.step = C_STEP;
  U.params = prm;
  const int nA = live ? sdp_feasible_count(U, s.x, s.cash, s.preq) : 0;
  const int nD = P.n_demand;

  double best = SDP_MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;
  int bestk = 0;
  bool bad = false;
  for (int k = as; k < nA; k += AS) {
    const double a = (double)k * C_STEP;
    double acc = 0.0;
    for (int j = 0; j < nD; ++j) {
      const double2 dp = s_pmf[j];
#ifdef SDP_USER_CELL
      // the user's fused callback: immediate value AND successor of the cell in one evaluation (a cash-type transition is
      // `cash + immediateValue(...)`, CashConstraint.java:125: written as two functions the increment is spelled out twice)
      double imm, nx, nc, nq;
      sdp_cell(U, s.x, s.cash, s.preq, a, dp.x, imm, nx, nc, nq);
#else
      const double imm = sdp_immediate(U, s.x, s.cash, s.preq, a, dp.x);
      double nx = 0, nc = 0, nq = 0;
      if (!LAST) sdp_transition(U, s.x, s.cash, s.preq, a, dp.x, nx, nc, nq);
#endif
      if (SDP_SURVIVAL) {  // RiskRecursion.java:78-98
        if (LAST) {
          acc += dp.y * ((s.cash + imm) >= 0 ? 1.0 : 0.0);
        } else {
          acc += (dp.y * P.gamma) * (nc < 0 ? 0.0 : v_next[c_next_index(P, nx, nc, nq, bad)]);
        }
      } else {
        acc += dp.y * imm;
        if (!LAST) acc += (dp.y * P.gamma) * v_next[c_next_index(P, nx, nc, nq, bad)];
      }
    }
    if (SDP_MAXDIR ? (acc > best) : (acc < best)) {
      best = acc;
      bestk = k;
    }
  }
  if (bad) atomicOr(err, 1);
  // the action slots of a state that live in this wave (lanes sx, sx + SX, ...)
#pragma unroll
  for (int off = SX; off < 64; off <<= 1) {
    double ov = __shfl_xor(best, off, 64);
    int ok = __shfl_xor(bestk, off, 64);
    if (c_better(SDP_MAXDIR, ov, ok, best, bestk)) {
      best = ov;
      bestk = ok;
    }
  }
  const int wave = tid >> 6;
  const int lane = tid & 63;
  if (lane < SX) {
    s_val[wave * SX + lane] = best;
    s_k[wave * SX + lane] = bestk;
  }
  __syncthreads();
  if (tid < SX) {
    double bv = s_val[tid];
    int bk = s_k[tid];
    for (int w = 1; w < 4; ++w) {
      double ov = s_val[w * SX + tid];
      int ok = s_k[w * SX + tid];
      if (c_better(SDP_MAXDIR, ov, ok, bv, bk)) {
        bv = ov;
        bk = ok;
      }
    }
    if (live) {
      v_cur[idx] = bv;
      pol[idx] = bk;
    }
    // cells of this workgroup = sum over its states of nA * D (tid < SX holds as == 0: nA of state sx)
    unsigned long long c = live ? (unsigned long long)nA * (unsigned long long)nD : 0ull;
#pragma unroll
    for (int off = SX / 2; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (tid == 0 && cells) atomicAdd(cells, c);
  }
}

#ifdef SDP_SHAPE_LEVEL
// M(m) = sdp_level_cost(lev0 + m step) for the n_m window levels of a period starting at m_min, c(a) = sdp_action_cost(a step):
// what the library's F1 window kernel reads in place of its built-in costs (sdp_window.hpp: WinParams::m_tab / c_tab).
extern "C" __global__ __launch_bounds__(256) void sdp_custom_tabulate(CParams P, double lev0, int m_min, int n_m,
                                                                     double* __restrict__ m_tab, int n_a,
                                                                     double* __restrict__ c_tab) {
  double prm[SDP_NP];
#pragma unroll
  for (int i = 0; i < SDP_NP; ++i) prm[i] = P.user[i];
  sdp_ctx U;
  U.period = P.period;
  U.T = P.T;
  U.step = P.step;
  U.params = prm;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_m) m_tab[i] = sdp_level_cost(U, lev0 + (double)(m_min + i) * P.step);
  if (i < n_a) c_tab[i] = sdp_action_cost(U, (double)i * P.step);
}
#endif

#define SDP_CUSTOM_PERIOD(SX)                                                                                        \
  extern "C" __global__ __launch_bounds__(256) void sdp_custom_period_##SX(                                          \
      CParams P, const double* __restrict__ v_next, double* __restrict__ v_cur, int* __restrict__ pol,               \
      const double* __restrict__ pmf_d, const double* __restrict__ pmf_p, sdp_i64 lo, sdp_i64 hi,                    \
      const double* __restrict__ qx, const double* __restrict__ qcash, const double* __restrict__ qpreq,             \
      unsigned long long* __restrict__ cells, int* __restrict__ err) {                                               \
    if (P.is_last)                                                                                                   \
      custom_period_body<SX, true>(P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, qx, qcash, qpreq, cells, err);       \
    else                                                                                                             \
      custom_period_body<SX, false>(P, v_next, v_cur, pol, pmf_d, pmf_p, lo, hi, qx, qcash, qpreq, cells, err);      \
  }
SDP_CUSTOM_PERIOD(64)
SDP_CUSTOM_PERIOD(16)
SDP_CUSTOM_PERIOD(4)

// Forward reachable set (Recursion.java:90's key set): successors of every marked state over all feasible
// actions and all demands; bankrupt successors of the survival loop are not visited.
extern "C" __global__ __launch_bounds__(256) void sdp_custom_reach(
    CParams P, const unsigned char* __restrict__ mask_cur, unsigned char* __restrict__ mask_next,
    const double* __restrict__ pmf_d, sdp_i64 n, const double* __restrict__ qx, const double* __restrict__ qcash,
    const double* __restrict__ qpreq, int* __restrict__ err) {
  const int sx = threadIdx.x & 63;
  const int as = threadIdx.x >> 6;
  const sdp_i64 idx = (sdp_i64)blockIdx.x * 64 + sx;
  if (idx >= n) return;
  CState s;
  if (qx) {
    s.x = qx[idx];
    s.cash = qcash ? qcash[idx] : 0.0;
    s.preq = qpreq ? qpreq[idx] : 0.0;
  } else {
    if (!mask_cur[idx]) return;
    c_decode(P, idx, s);
  }
  // The user's constants are copied into a private array first: constant indices then become registers, a
  // period-dependent index one hoisted load -- read through the global pointer, every use inside a branch of the
  // user's code would be a scalar load with a full wait in the demand loop (the compiler may not speculate it).
  double prm[SDP_NP];
#pragma unroll
  for (int i = 0; i < SDP_NP; ++i) prm[i] = P.user[i];
  sdp_ctx U;
  U.period = P.period;
  U.T = P.T;
  U.step = P.step;
  U.params = prm;
  const int nA = sdp_feasible_count(U, s.x, s.cash, s.preq);
  bool bad = false;
  for (int k = as; k < nA; k += 4) {
    const double a = (double)k * P.step;
    for (int j = 0; j < P.n_demand; ++j) {
      double nx, nc, nq;
      sdp_transition(U, s.x, s.cash, s.preq, a, pmf_d[j], nx, nc, nq);
      if (SDP_SURVIVAL && nc < 0) continue;
      mask_next[c_next_index(P, nx, nc, nq, bad)] = 1;
    }
  }
  if (bad) atomicOr(err, 1);
}
)SDPSRC";

}  // namespace sdp

// sdp_device.hpp -- device-side parameter block and the per-cell functors (gfx950 / CDNA4).
//
// One "cell" = one (state, action, demand) evaluation of the reference's inner loop
// (Recursion.java:138-144).  The functors below are the closed-form lambda families of the
// in-scope drivers, written in the reference's operation order so that every intermediate
// rounds exactly as the Java double arithmetic does.  Compile with -ffp-contract=off: an FMA
// anywhere in here changes low bits of Q(s,a) and can flip a near-tied arg-min.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdp {

constexpr int FAM_BACKORDER = 1;
constexpr int FAM_LEADTIME = 2;
constexpr int FAM_CASH = 3;
constexpr int FAM_OVERDRAFT = 4;
constexpr int FAM_CASH_LEADTIME = 5;
constexpr int FAM_SURVIVAL = 6;  // RiskRecursion.getSurvProb over the lambdas of cashSurvival.java

struct Grid {
  double x_lo;   // inventory value of ix = 0
  int64_t nx;    // inventory points
  int64_t nc;    // cash points (1 when the family has no cash axis)
  int64_t nq;    // pipeline (preQ) points (1 when no lead time); lead time 2: nq1 * nq2, iq = iq2 * nq1 + iq1
  int64_t k_lo;  // cash key of ic = 0
  int64_t nq1;   // points of the inner pipeline axis (== nq unless lead time 2)
};

// Everything a period kernel needs, passed by value in the kernarg segment.
struct DevParams {
  int32_t family;
  int32_t maxdir;     // OptDirection.MAX
  int32_t is_last;    // period == T: no future term (Recursion.java:140), salvage applies
  int32_t n_demand;   // pmf[t].length
  int32_t clamp_inventory;
  int32_t cash_formula;
  int32_t cash_round_int_div;
  int32_t n_actions_full;  // (int)(maxOrderQuantity/step)+1, already 1 when zero_order_last_period hits
  int32_t lead2;           // LEADTIME family with a two-stage pipeline (x, q1, q2)
  int32_t pad0;
  double step, inv_step;
  double min_inventory, max_inventory;
  double max_order_quantity;
  double K, v, h, pi;  // fixed / unit ordering cost, holding, penalty
  double price, salvage, one_plus_deposit, overhead, one_minus_overhead_rate, gamma;
  double min_cash, max_cash, round_mult, round_div;
  double r0, r2, r3, limit, interest_free;
  Grid cur, next;
  // sdpgpu_set_action_counts: getFeasibleActions.apply(state).length of every GRID state of this period, indexed by
  // flat state index (nullptr: the family's own rule)
  const int32_t* counts;
};

// java.lang.Math.max / min for the non-NaN operands that occur here.  v_max_f64 / v_min_f64
// order -0.0 below +0.0, which is Java's rule too.
__device__ __forceinline__ double jmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double jmin(double a, double b) { return fmin(a, b); }

// java.lang.Math.round(double) as an integer-valued double: nearest, ties toward +infinity.
__device__ __forceinline__ double jround_d(double x) {
  double f = floor(x);
  double diff = x - f;  // exact
  return diff >= 0.5 ? f + 1.0 : f;
}

struct StateT {
  double x, cash, preq;
  double preq2 = 0;  // lead time 2: the order arriving next period
  double r = 0;      // CASH family, cash_formula 2 (CashConstraintXR.java): working capital R = cash + variCost * x
};

// CASH family with cash_formula 2 = the (x, R) state of sdp.cash.CashRecursionXR over the lambdas of
// cash.singleItem.CashConstraintXR (CashConstraintXR.java:84-125; Chao 2008).  The grid is still (inventory, cash key):
// the transition rounds the cash balance to a grid value and stores R = nextCash + variCost * nextInventory (:121),
// and every lambda starts from initCash = R - variCost * x (:94, :112), which need not equal the rounded balance bit
// for bit when variCost * x is inexact -- so R is formed and subtracted exactly as the reference does.
// from_grid: s.cash holds the grid's cash value; else (a queried state) s.cash holds R itself.
template <int FAM>
__device__ __forceinline__ void xr_state(const DevParams& P, StateT& s, bool from_grid) {
  if constexpr (FAM == FAM_CASH) {
    if (P.cash_formula == 2) {
      const double cx = P.v * s.x;
      s.r = from_grid ? s.cash + cx : s.cash;
      s.cash = s.r - cx;
    }
  }
}

// getFeasibleActions.apply(state).length -- see n_actions() in oracle/sdpref.c for the citations.
template <int FAM>
__device__ __forceinline__ int n_actions(const DevParams& P, const StateT& s) {
  if constexpr (FAM == FAM_CASH) {
    if (P.cash_formula == 2) {
      // CashConstraintXR.java:84-88: order-up-to levels y = x, x + 1, ... , `(int) (maxY - x) + 1` of them with
      // maxY = R / variCost < x ? x : R / variCost.  Action index k stands for y = x + k.
      const double ry = s.r / P.v;
      const double maxY = ry < s.x ? s.x : ry;
      const double len = maxY - s.x;
      return ((len != len) ? 0 : (int)len) + 1;
    }
    // CashConstraint.java:96-99: (int) Math.min(maxOrderQuantity, Math.max(0, (cash - overhead - K) / v))
    double m = jmin(P.max_order_quantity, jmax(0.0, (s.cash - P.overhead - P.K) / P.v));
    int mq = (m != m) ? 0 : (int)m;  // Java (int)NaN == 0
    return mq + 1;
  } else if constexpr (FAM == FAM_SURVIVAL) {
    // cashSurvival.java:98-105: maxQ = Math.min(cash / variCost, maxOrderQuantity); maxQ = Math.max(maxQ, 0)
    double m = jmax(jmin(s.cash / P.v, P.max_order_quantity), 0.0);
    int mq = (m != m) ? 0 : (int)m;
    return mq + 1;
  } else {
    return P.n_actions_full;
  }
}

// Piecewise overdraft interest, CashOverdraft.java:87-95.
__device__ __forceinline__ double overdraft_interest(const DevParams& P, double before) {
  double interest;
  if (before >= 0)
    interest = -P.r0 * before;
  else if (before >= -P.interest_free)
    interest = 0;
  else if (before >= -P.limit)
    interest = P.r2 * (-before - P.interest_free);
  else
    interest = P.r3 * (-before - P.limit) + P.r2 * (P.limit - P.interest_free);
  return interest;
}

// Per-(state, action) invariants hoisted out of the demand loop.  Hoisting keeps every
// operation and its operands identical to the reference; it only avoids recomputing them.
struct ActionCtx {
  double a;         // order quantity
  double base;      // x + a (F1/F3/F4) or x + preQ (F2/F5): the level before demand
  double fixed;     // action > 0 ? K : 0
  double var;       // v * action
  double fv;        // fixed + var (F1/F2: first add of totalCosts)
  double deposit;   // F3 formula 0: (cash - fixed - var) * (1 + depositeRate)
  double before;    // F4/F5: cashBalanceBefore
  double interest;  // F4/F5
  int64_t next_q_off;  // F2/F5: flat offset of the next state's preQ plane (iq' = action index)
};

template <int FAM>
__device__ __forceinline__ void action_setup(const DevParams& P, const StateT& s, int k, ActionCtx& c) {
  c.a = (double)k * P.step;
  c.fixed = c.a > 0 ? P.K : 0.0;
  c.var = P.v * c.a;
  c.fv = c.fixed + c.var;
  c.deposit = 0;
  c.before = 0;
  c.interest = 0;
  c.next_q_off = 0;
  if constexpr (FAM == FAM_BACKORDER) {
    c.base = s.x + c.a;
  } else if constexpr (FAM == FAM_LEADTIME) {
    c.base = s.x + s.preq;
    if (P.lead2)  // next state (x', q1' = q2, q2' = action): plane iq' = k * nq1 + iq2
      c.next_q_off = ((int64_t)k * P.next.nq1 + (int64_t)(s.preq2 * P.inv_step)) * P.next.nx;
    else
      c.next_q_off = (int64_t)k * P.next.nx * P.next.nc;
  } else if constexpr (FAM == FAM_CASH || FAM == FAM_SURVIVAL) {
    c.base = s.x + c.a;
    c.deposit = (s.cash - c.fixed - c.var) * P.one_plus_deposit;
  } else if constexpr (FAM == FAM_OVERDRAFT) {
    c.base = s.x + c.a;
    c.before = s.cash - c.fixed - c.var - P.overhead;
    c.interest = overdraft_interest(P, c.before);
  } else {  // FAM_CASH_LEADTIME
    c.base = s.x + s.preq;
    c.before = s.cash - c.var - P.overhead;
    c.interest = overdraft_interest(P, c.before);
    c.next_q_off = (int64_t)k * P.next.nx * P.next.nc;
  }
}

// Inventory index of a (clamped) next inventory value in the next period's grid.
// (v_cvt_i32_f64: grids are < 2^31 points per axis, checked at create.)
__device__ __forceinline__ int inv_index(const DevParams& P, double next_inv) {
  return (int)((next_inv - P.next.x_lo) * P.inv_step);
}

// Cash index after clamp + Math.round quantisation (CashConstraint.java:126-131).
__device__ __forceinline__ int cash_index(const DevParams& P, double next_cash) {
  next_cash = next_cash > P.max_cash ? P.max_cash : next_cash;
  next_cash = next_cash < P.min_cash ? P.min_cash : next_cash;
  double r = jround_d(next_cash * P.round_mult);
  // `/ 10`: long division truncates.  (r / 1.0 == r: the common integer-cash quantiser skips the divide.)
  if (P.cash_round_int_div && P.round_div != 1.0) r = trunc(r / P.round_div);
  return (int)r - (int)P.next.k_lo;
}

// One cell: immediate value and (when the period has a future) flat index of the next state.
// LASTMODE: 1 = period T (no transition, salvage applies), 0 = a period with a future, -1 = read P.is_last.
template <int FAM, int LASTMODE = -1>
__device__ __forceinline__ double cell(const DevParams& P, const StateT& s, const ActionCtx& c, double d,
                                       int64_t& next_idx) {
  const bool is_last = LASTMODE < 0 ? (P.is_last != 0) : (LASTMODE != 0);
  if constexpr (FAM == FAM_BACKORDER || FAM == FAM_LEADTIME) {
    // CLSP.java:263-272 / Leadtime.java:71-81
    double level = c.base - d;
    double hold = P.h * jmax(level, 0.0);
    double pen = P.pi * jmax(-level, 0.0);
    double imm = c.fv + hold + pen;
    if (!is_last) {
      double nx = level;  // CLSP.java:255-258 / Leadtime.java:62-63
      if (P.clamp_inventory) {
        nx = nx > P.max_inventory ? P.max_inventory : nx;
        nx = nx < P.min_inventory ? P.min_inventory : nx;
      }
      next_idx = c.next_q_off + inv_index(P, nx);
    }
    return imm;
  } else if constexpr (FAM == FAM_CASH) {
    // CashConstraint.java:103-119 (formula 0) / CashConstraintTesting.java:117-132 (formula 1)
    double revenue = P.price * jmin(c.base, d);
    double level = c.base - d;
    double hold = P.h * jmax(level, 0.0);
    double inc;
    if (P.cash_formula != 1)  // formula 0 and the (x, R) state of CashConstraintXR.java:91-105 (s.cash = initCash there)
      inc = P.one_minus_overhead_rate * revenue + c.deposit - hold - P.overhead - s.cash;
    else
      inc = revenue - c.fixed - c.var - hold - P.overhead;
    double sal = is_last ? P.salvage * jmax(level, 0.0) : 0.0;
    inc += sal;
    double end_cash = s.cash + inc;
    if (end_cash < 0) inc += P.pi * end_cash;
    if (!is_last) {
      // CashConstraint.java:124-131
      double ninv = jmax(0.0, level);
      double ncash = s.cash + inc;
      ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
      ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
      next_idx = (int64_t)inv_index(P, ninv) * P.next.nc + cash_index(P, ncash);
    }
    return inc;
  } else if constexpr (FAM == FAM_SURVIVAL) {
    // cashSurvival.java:112-125 (immediate value) and :128-143 (transition)
    double revenue = P.price * jmin(c.base, d);
    double level = c.base - d;
    double hold = P.h * jmax(level, 0.0);
    double inc = revenue + c.deposit - hold - P.overhead - s.cash;
    double sal = is_last ? P.salvage * jmax(level, 0.0) : 0.0;
    inc += sal;
    if (!is_last) {
      double ninv = jmax(0.0, level);
      double ncash = s.cash + inc;
      ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
      ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
      const int ci = cash_index(P, ncash);
      // RiskRecursion.java:90-92: a successor with negative cash is worth 0 and is never visited.  The rounded
      // cash is negative exactly when its integer key is.
      next_idx = (int64_t)ci + P.next.k_lo < 0 ? -1 : (int64_t)inv_index(P, ninv) * P.next.nc + ci;
    }
    return inc;
  } else {
    // CashOverdraft.java:80-104 / SingleProductLeadtime.java:82-104
    double revenue = P.price * jmin(c.base, d);
    double level = c.base - d;
    double after = c.before - c.interest + revenue;
    double inc = after - s.cash;
    double sal = is_last ? P.salvage * jmax(level, 0.0) : 0.0;
    inc += sal;
    if (!is_last) {
      double ninv = jmax(0.0, level);
      double ncash = s.cash + inc;
      ninv = ninv > P.max_inventory ? P.max_inventory : ninv;
      ninv = ninv < P.min_inventory ? P.min_inventory : ninv;
      next_idx = c.next_q_off + (int64_t)inv_index(P, ninv) * P.next.nc + cash_index(P, ncash);
    }
    return inc;
  }
}

// Decode a flat state index of the current period: idx = (iq * nx + ix) * nc + ic.
template <int FAM>
__device__ __forceinline__ void decode_state(const DevParams& P, int64_t idx, StateT& s) {
  s.cash = 0;
  s.preq = 0;
  if constexpr (FAM == FAM_BACKORDER) {
    s.x = P.cur.x_lo + (double)idx * P.step;
  } else if constexpr (FAM == FAM_LEADTIME) {
    int64_t iq = idx / P.cur.nx;
    int64_t ix = idx - iq * P.cur.nx;
    int64_t iq2 = iq / P.cur.nq1;  // 0 with lead time 1 (nq1 == nq)
    s.x = P.cur.x_lo + (double)ix * P.step;
    s.preq = (double)(iq - iq2 * P.cur.nq1) * P.step;
    s.preq2 = (double)iq2 * P.step;
  } else {
    int64_t ic = idx % P.cur.nc;
    int64_t r = idx / P.cur.nc;
    int64_t ix = r % P.cur.nx;
    int64_t iq = r / P.cur.nx;
    s.x = P.cur.x_lo + (double)ix * P.step;
    double k = (double)(P.cur.k_lo + ic);
    s.cash = P.cash_round_int_div ? k : k / P.round_div;  // the double Math.round(c*m)/div yields
    if constexpr (FAM == FAM_CASH_LEADTIME) s.preq = (double)iq * P.step;
    xr_state<FAM>(P, s, true);
  }
}

// Lexicographic (value, action index) order == the reference's strict-compare scan in ascending
// action order (Recursion.java:146-157): the lowest index wins ties.
template <bool MAXDIR>
__device__ __forceinline__ bool better(double v2, int k2, double v, int k) {
  if constexpr (MAXDIR)
    return v2 > v || (v2 == v && k2 < k);
  else
    return v2 < v || (v2 == v && k2 < k);
}

}  // namespace sdp

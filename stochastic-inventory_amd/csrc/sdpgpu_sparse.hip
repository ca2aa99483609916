// sdpgpu_sparse.hip -- reachable-set ("sparse") SDP engine for state spaces that are not grids.
//
// The dense engine (sdpgpu.hip) needs a gridded state space.  The reference's two-product classes have
// none: `CashRecursionMultiLead` (src/sdp/cash/multiItem/CashRecursionMultiLead.java:54-90) driven by the
// lambdas of `MultiProductLeadtime.main` (src/cash/overdraft/MultiProductLeadtime.java:150-223) carries
// an UN-ROUNDED cash balance in its 5-tuple state (I1, I2, preQ1, preQ2, cash), so only the states
// actually reachable from the initial one exist -- which is exactly what the reference's memoised
// recursion enumerates.  This file does the same on the GPU, level by level:
//
//   forward  t = 1..T-1 : expand every state of S_t over all (action, demand) pairs, sort the candidates
//                          by a 64-bit hash of the tuple (rocPRIM radix sort through hipCUB), keep one
//                          representative per distinct tuple -> S_{t+1}; every candidate remembers the id of
//                          its representative, so the backward pass needs no searching.
//   backward t = T..1   : one workgroup per state: Q(s, a) for the q_bound^2 actions (lanes = actions,
//                          demand loop serial and in the reference's order), then the reference's
//                          tolerance scan `if (Q > val + 0.1)` in action order.
//
// A hash collision can only make two equal tuples land in two representatives (both are then evaluated,
// to the same value); it cannot merge different tuples, because representatives are cut by comparing
// the full tuples.  This family is the one the reference records outputs for (KAT-1/KAT-2,
// MultiProductLeadtime.java:30-50): tests/test_gpu_multilead.py reproduces them on the GPU.
#include "../../include/sdpgpu.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>
#include <type_traits>

namespace {

thread_local std::string g_ml_error;
thread_local sdpgpu_multi_table* g_ml_table = nullptr;  // read-out request for the next solve of this thread

struct MLParams {
  double price[2], vari[2], sal[2];
  double r0, r1, r2, limit, interest_free;
  double min_inventory, max_inventory, min_cash, max_cash, discount;
  double overhead;  // of the period being processed
  int32_t qb;       // actions (i, j), i, j in [0, qb)
  int32_t nd;       // demand pairs, index = i1 * n2 + i2 (GetPmfMulti.java:157-172)
  int32_t is_last;  // period == T: salvage applies, no transition
  int32_t cash_int_cast;  // MultiProductLeadtime.java:219
  int32_t model;          // 0: CashRecursionMultiLead + MultiProductLeadtime lambdas; 1: CashRecursionMulti + MultiItemCash;
                          // 2: CashRecursionMultiXR + MultiItemCashXR (state (x1, x2, R), actions = order-up-to levels)
  double one_minus_deposit;  // model 2: 1 - depositeRate
};

struct Tuple {
  double i1, i2, q1, q2, cash;
};

__device__ __forceinline__ double jmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ double jmin(double a, double b) { return fmin(a, b); }

// Piecewise overdraft interest, MultiProductLeadtime.java:186-194.
__device__ __forceinline__ double ml_interest(const MLParams& P, double before) {
  double interest;
  if (before >= 0)
    interest = -P.r0 * before;
  else if (before >= -P.interest_free)
    interest = 0;
  else if (before >= -P.limit)
    interest = P.r1 * (-before - P.interest_free);
  else
    interest = P.r2 * (-before - P.limit) + P.r1 * (P.limit - P.interest_free);
  return interest;
}

// Demand-dependent pieces of one state, shared by all actions (MultiProductLeadtime.java:170-183).
struct DemandTerms {
  double revenue, sal_value, next_i1, next_i2;
};

__device__ __forceinline__ DemandTerms demand_terms(const MLParams& P, const Tuple& s, double d1, double d2) {
  DemandTerms t;
  const double end1 = jmax(0.0, s.i1 + s.q1 - d1);
  const double end2 = jmax(0.0, s.i2 + s.q2 - d2);
  const double revenue1 = P.price[0] * jmin(d1, s.i1 + s.q1);
  const double revenue2 = P.price[1] * jmin(s.i2 + s.q2, d2);
  t.revenue = revenue1 + revenue2;
  t.sal_value = P.is_last ? (P.sal[0] * end1 + P.sal[1] * end2) : 0.0;
  // the transition's own end inventories (:210-221): upper clamp on product 1 only, lower clamp on
  // product 2 only, then (int) casts
  double n1 = s.i1 + s.q1 - d1;
  n1 = jmax(0.0, n1);
  double n2 = s.i2 + s.q2 - d2;
  n2 = jmax(0.0, n2);
  n1 = n1 > P.max_inventory ? P.max_inventory : n1;
  n2 = n2 < P.min_inventory ? P.min_inventory : n2;
  t.next_i1 = (double)(int)n1;
  t.next_i2 = (double)(int)n2;
  return t;
}

// cashIncrement of (state, action, demand): MultiProductLeadtime.java:177-198 with the action-only part
// (`bi` = cashBalanceBefore - interest) hoisted by the caller.
__device__ __forceinline__ double cash_increment(const Tuple& s, double bi, const DemandTerms& t) {
  const double after = bi + t.revenue + t.sal_value;
  return after - s.cash;
}

__device__ __forceinline__ double next_cash(const MLParams& P, const Tuple& s, double inc) {
  double c = s.cash + inc;
  c = c > P.max_cash ? P.max_cash : c;
  c = c < P.min_cash ? P.min_cash : c;
  if (P.cash_int_cast) c = (double)(int)c;  // `nextCash = (int) nextCash` (:219)
  return c;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long h, double v) {
  v += 0.0;  // -0.0 -> +0.0 (the reference's equals() compares with ==)
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  h ^= u + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  return h;
}

__device__ __forceinline__ unsigned long long tuple_hash(const Tuple& t) {
  unsigned long long h = 0x243f6a8885a308d3ull;
  h = mix64(h, t.i1);
  h = mix64(h, t.i2);
  h = mix64(h, t.q1);
  h = mix64(h, t.q2);
  h = mix64(h, t.cash);
  return h;
}

__device__ __forceinline__ bool tuple_eq(const Tuple& a, const Tuple& b) {
  return a.i1 == b.i1 && a.i2 == b.i2 && a.q1 == b.q1 && a.q2 == b.q2 && a.cash == b.cash;
}

// ---- integer-lattice states (models 1 and 2) ---------------------------------------------------------------
// After their (int) casts the successors of the two cash-constrained families live on an integer lattice
// (i1, i2, cash-or-R) inside a box the parameters bound.  A period's state set is then a BITMAP over the box and the
// id of a state is its rank among the set bits (exclusive prefix of the words' popcounts + the bits below it):
// nothing per candidate is stored, so the horizon is not limited by states x actions x demand pairs the way the
// sort-based path is.  The price: the backward pass recomputes the successor of every cell and reads two words to
// rank it.
// Axis order (i1, R, i2) with R = cash + variCost . x and i2 fastest: as long as a product does not stock out, the R
// a (state, demand pair) leads to does not depend on the order quantities (revenue - cost of the units sold), and
// neighbouring second-product quantities lead to neighbouring i2 -- so the 64 lanes of a wave (consecutive second
// actions) rank and read a handful of words instead of 64 lines.  (XR family: the state's third field IS R, skew 0;
// MultiItemCash family: R is formed from the cash with the integer unit costs, or stays the cash when they are not.)
struct Lattice {
  long long n2, nr;  // idx = (i1 * nr + (R - r0)) * n2 + i2
  long long r0;
  long long skew1, skew2;  // R = cash-field + skew1 * i1 + skew2 * i2
  long long bits;
};

__device__ __forceinline__ long long lattice_index(const Lattice& L, const Tuple& t) {
  const long long i1 = (long long)t.i1, i2 = (long long)t.i2;
  const long long r = (long long)t.cash + L.skew1 * i1 + L.skew2 * i2;
  return (i1 * L.nr + (r - L.r0)) * L.n2 + i2;
}

__device__ __forceinline__ Tuple lattice_tuple(const Lattice& L, long long idx) {
  Tuple t;
  const long long i2 = idx % L.n2, q = idx / L.n2;
  const long long r = q % L.nr + L.r0, i1 = q / L.nr;
  t.cash = (double)(r - L.skew1 * i1 - L.skew2 * i2);
  t.i2 = (double)i2;
  t.i1 = (double)i1;
  t.q1 = 0.0;
  t.q2 = 0.0;
  return t;
}

// rank of a set bit: states of a period are numbered in lattice order
// (word and prefix interleaved: one 8-byte gather ranks a state)
__device__ __forceinline__ int lattice_rank(const uint2* __restrict__ rank_info, long long idx) {
  const uint2 wp = rank_info[idx >> 5];
  const unsigned int below = wp.x & ((1u << (idx & 31)) - 1u);
  return (int)(wp.y + __popc(below));
}

// ---- model 1: sdp.cash.multiItem.CashRecursionMulti over the lambdas of cash.multiItem.MultiItemCash ----------
// buildActionList (MultiItemCash.java:66-76): (i, j) is offered iff variCost[0] * i + variCost[1] * j < iniCash + 0.1
__device__ __forceinline__ bool mc_feasible(const MLParams& P, const Tuple& s, int a1, int a2) {
  return P.vari[0] * a1 + P.vari[1] * a2 < s.cash + 0.1;
}

// immediateValue (MultiItemCash.java:79-99)
__device__ __forceinline__ double mc_immediate(const MLParams& P, const Tuple& s, int a1, int a2, double demand1,
                                               double demand2) {
  const double action1 = (double)a1, action2 = (double)a2;
  const double endInventory1 = jmax(0.0, s.i1 + action1 - demand1);
  const double endInventory2 = jmax(0.0, s.i2 + action2 - demand2);
  const double revenue1 = P.price[0] * (s.i1 + action1 - endInventory1);
  const double revenue2 = P.price[1] * (s.i2 + action2 - endInventory2);
  const double revenue = revenue1 + revenue2;
  const double orderingCost1 = P.vari[0] * action1;
  const double orderingCost2 = P.vari[1] * action2;
  const double orderingCosts = orderingCost1 + orderingCost2;
  double salValue = 0;
  if (P.is_last) salValue = P.sal[0] * endInventory1 + P.sal[1] * endInventory2;
  return revenue - orderingCosts + salValue;
}

// stateTransition (MultiItemCash.java:103-118): upper clamp on product 1 only, lower clamp on product 2 only, (int) casts
__device__ __forceinline__ Tuple mc_successor(const MLParams& P, const Tuple& s, int a1, int a2, double demand1,
                                              double demand2) {
  double endInventory1 = s.i1 + (double)a1 - demand1;
  endInventory1 = jmax(0.0, endInventory1);
  double endInventory2 = s.i2 + (double)a2 - demand2;
  endInventory2 = jmax(0.0, endInventory2);
  double nextCash = s.cash + mc_immediate(P, s, a1, a2, demand1, demand2);
  nextCash = nextCash > P.max_cash ? P.max_cash : nextCash;
  nextCash = nextCash < P.min_cash ? P.min_cash : nextCash;
  endInventory1 = endInventory1 > P.max_inventory ? P.max_inventory : endInventory1;
  endInventory2 = endInventory2 < P.min_inventory ? P.min_inventory : endInventory2;
  Tuple n;
  n.cash = (double)(int)nextCash;
  n.i1 = (double)(int)endInventory1;
  n.i2 = (double)(int)endInventory2;
  n.q1 = 0.0;
  n.q2 = 0.0;
  return n;
}

// ---- model 2: sdp.cash.multiItem.CashRecursionMultiXR over the lambdas of cash.multiItem.MultiItemCashXR ------
// The state is (x1, x2, R) with R = cash + variCost . x (kept in Tuple::cash); an action is a pair of order-up-to
// levels (y1, y2) = ((int) x1 + i, (int) x2 + j), i, j in [0, Qbound) (MultiItemCashXR.java:92-105).
// immediateValue (MultiItemCashXR.java:108-128)
__device__ __forceinline__ double xr_immediate(const MLParams& P, const Tuple& s, double action1, double action2,
                                               double demand1, double demand2) {
  const double endInventory1 = jmax(0.0, action1 - demand1);
  const double endInventory2 = jmax(0.0, action2 - demand2);
  const double revenue1 = P.price[0] * (action1 - endInventory1);
  const double revenue2 = P.price[1] * (action2 - endInventory2);
  const double revenue = revenue1 + revenue2;
  const double initialCash = s.cash - P.vari[0] * s.i1 - P.vari[1] * s.i2;
  const double orderingCostY1 = P.vari[0] * action1;
  const double orderingCostY2 = P.vari[1] * action2;
  const double orderingCostsY = orderingCostY1 + orderingCostY2;
  double salValue = 0;
  if (P.is_last) salValue = P.sal[0] * endInventory1 + P.sal[1] * endInventory2;
  return revenue + P.one_minus_deposit * (s.cash - orderingCostsY) + salValue - initialCash;
}

// stateTransition (MultiItemCashXR.java:132-148)
__device__ __forceinline__ Tuple xr_successor(const MLParams& P, const Tuple& s, double action1, double action2,
                                              double demand1, double demand2) {
  double endInventory1 = action1 - demand1;
  endInventory1 = jmax(0.0, endInventory1);
  double endInventory2 = action2 - demand2;
  endInventory2 = jmax(0.0, endInventory2);
  const double initialCash = s.cash - P.vari[0] * s.i1 - P.vari[1] * s.i2;
  double nextCash = initialCash + xr_immediate(P, s, action1, action2, demand1, demand2);
  nextCash = nextCash > P.max_cash ? P.max_cash : nextCash;
  nextCash = nextCash < P.min_cash ? P.min_cash : nextCash;
  endInventory1 = endInventory1 > P.max_inventory ? P.max_inventory : endInventory1;
  endInventory2 = endInventory2 < P.min_inventory ? P.min_inventory : endInventory2;
  nextCash = (double)(int)nextCash;
  endInventory1 = (double)(int)endInventory1;
  endInventory2 = (double)(int)endInventory2;
  Tuple n;
  n.i1 = endInventory1;
  n.i2 = endInventory2;
  n.q1 = 0.0;
  n.q2 = 0.0;
  n.cash = (double)(int)nextCash + P.vari[0] * endInventory1 + P.vari[1] * endInventory2;  // nextR
  return n;
}

// The successor of (state, action a, demand j).
__device__ __forceinline__ Tuple successor(const MLParams& P, const Tuple& s, int a, const double2* dem, int j) {
  const int a1 = a / P.qb, a2 = a - a1 * P.qb;
  if (P.model == 1) {
    // an action the state is not offered has no successors of its own: it stands in for (0, 0), which is always
    // offered (0 < cash + 0.1 with cash >= min_cash >= 0), so the candidate list gains nothing new
    const bool ok = mc_feasible(P, s, a1, a2);
    return mc_successor(P, s, ok ? a1 : 0, ok ? a2 : 0, dem[j].x, dem[j].y);
  }
  if (P.model == 2)
    return xr_successor(P, s, (double)((int)s.i1 + a1), (double)((int)s.i2 + a2), dem[j].x, dem[j].y);
  const double oc = P.vari[0] * (double)a1 + P.vari[1] * (double)a2;
  const double before = s.cash - oc - P.overhead;
  const double bi = before - ml_interest(P, before);
  const DemandTerms t = demand_terms(P, s, dem[j].x, dem[j].y);
  Tuple n;
  n.i1 = t.next_i1;
  n.i2 = t.next_i2;
  n.q1 = (double)a1;
  n.q2 = (double)a2;
  n.cash = next_cash(P, s, cash_increment(s, bi, t));
  return n;
}

// ---- forward: candidates of one period ------------------------------------------------------------
// candidate index c = (s * NA + a) * ND + j
__global__ __launch_bounds__(256) void expand_kernel(MLParams P, const Tuple* __restrict__ states, int64_t n_states,
                                                     const double2* __restrict__ dem, unsigned long long* __restrict__ hash,
                                                     unsigned int* __restrict__ order) {
  const int NA = P.qb * P.qb;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (s, a)
  if (g >= n_states * NA) return;
  const int64_t s = g / NA;
  const int a = (int)(g - s * NA);
  const Tuple st = states[s];
  for (int j = 0; j < P.nd; ++j) {
    const Tuple n = successor(P, st, a, dem, j);
    const int64_t c = g * P.nd + j;
    hash[c] = tuple_hash(n);
    order[c] = (unsigned int)c;
  }
}

// heads of runs of equal tuples in hash-sorted order
__global__ __launch_bounds__(256) void mark_heads_kernel(MLParams P, const Tuple* __restrict__ states,
                                                         const double2* __restrict__ dem,
                                                         const unsigned int* __restrict__ sorted, int64_t n,
                                                         int* __restrict__ head) {
  const int NA = P.qb * P.qb;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  auto tuple_of = [&](unsigned int c) {
    const int64_t g = c / P.nd;
    const int j = (int)(c - g * P.nd);
    const int64_t s = g / NA;
    return successor(P, states[s], (int)(g - s * NA), dem, j);
  };
  if (k == 0) {
    head[0] = 1;
    return;
  }
  head[k] = tuple_eq(tuple_of(sorted[k]), tuple_of(sorted[k - 1])) ? 0 : 1;
}

// rank[k] = inclusive scan of head; representative id = rank - 1
__global__ __launch_bounds__(256) void scatter_kernel(MLParams P, const Tuple* __restrict__ states,
                                                      const double2* __restrict__ dem,
                                                      const unsigned int* __restrict__ sorted,
                                                      const int* __restrict__ head, const int* __restrict__ rank,
                                                      int64_t n, Tuple* __restrict__ next_states,
                                                      int* __restrict__ uid) {
  const int NA = P.qb * P.qb;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const unsigned int c = sorted[k];
  const int id = rank[k] - 1;
  uid[c] = id;
  if (head[k]) {
    const int64_t g = c / P.nd;
    const int j = (int)(c - g * P.nd);
    const int64_t s = g / NA;
    next_states[id] = successor(P, states[s], (int)(g - s * NA), dem, j);
  }
}

// ---- forward on the lattice: mark the successors of every (state, offered action, demand pair) ---------------
__global__ __launch_bounds__(256) void lattice_mark_kernel(MLParams P, Lattice L, const Tuple* __restrict__ states,
                                                           int64_t n_states, const double2* __restrict__ dem,
                                                           unsigned int* __restrict__ words, int* __restrict__ oob) {
  const int NA = P.qb * P.qb;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (s, a)
  if (g >= n_states * NA) return;
  const int64_t s = g / NA;
  const int a = (int)(g - s * NA);
  const Tuple st = states[s];
  if (P.model == 1 && !mc_feasible(P, st, a / P.qb, a % P.qb)) return;  // not offered: no successors
#pragma unroll 4
  for (int j = 0; j < P.nd; ++j) {
    const long long idx = lattice_index(L, successor(P, st, a, dem, j));
    if (idx < 0 || idx >= L.bits) {
      *oob = 1;  // the host's box was too small: reported, never silently dropped
      continue;
    }
    const unsigned int bit = 1u << (idx & 31);
    if (!(words[idx >> 5] & bit)) atomicOr(&words[idx >> 5], bit);  // (most candidates are already marked)
  }
}

__global__ __launch_bounds__(256) void lattice_popc_kernel(const unsigned int* __restrict__ words, long long n_words,
                                                           unsigned int* __restrict__ counts) {
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w < n_words) counts[w] = (unsigned int)__popc(words[w]);
}

__global__ __launch_bounds__(256) void lattice_list_kernel(Lattice L, const unsigned int* __restrict__ words,
                                                           const unsigned int* __restrict__ prefix, long long n_words,
                                                           Tuple* __restrict__ states, uint2* __restrict__ rank_info) {
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  unsigned int bits = words[w];
  unsigned int k = prefix[w];
  rank_info[w] = make_uint2(bits, k);
  while (bits) {
    const int b = __ffs((int)bits) - 1;
    bits &= bits - 1;
    states[k++] = lattice_tuple(L, (w << 5) + b);
  }
}

// ---- forward on the lattice, XR family with exact data: marking through the POST-ORDER triples -------------------------
// MultiItemCashXR's transition reads its state through three numbers only when its arithmetic is exact: the order-up-to
// levels y1, y2 and W = R - c . y.  initialCash = R - c . x; immediate = revenue + 1.0 (R - c . y) - initialCash; nextCash =
// initialCash + immediate (MultiItemCashXR.java:108-148) -- with no deposit rate, integer unit costs, R and x integers (lattice
// states are), prices and demands on a 1/8 grid and everything far below 2^53, every one of those operations is exact, so
// nextCash = revenue + W whatever initialCash was, and the end inventories read y and the demand alone.  The host checks those
// conditions (xr_exact_data).  Every (state, order) pair then marks its triple in a bitmap of its own, and the successors are
// marked from the DISTINCT triples -- each as the stand-in state (x = y, R = W + c . y) ordering up to its own level, through the
// same xr_successor -- instead of from every (state, order, demand pair): 4.4e9 pairs fold onto a few 1e7 triples in period 3 of
// MultiItemCashXR's four-period run, whose marking pass was 1.08 of 3.45 s.
__global__ __launch_bounds__(256) void xr_triple_mark_kernel(MLParams P, Lattice B, long long w0, const Tuple* __restrict__ states,
                                                             int64_t n_states, unsigned int* __restrict__ words,
                                                             int* __restrict__ oob) {
  const int NA = P.qb * P.qb;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (s, a)
  if (g >= n_states * NA) return;
  const int64_t s = g / NA;
  const int a = (int)(g - s * NA);
  const Tuple st = states[s];
  const int a1 = a / P.qb, a2 = a - a1 * P.qb;
  const long long y1 = (long long)((int)st.i1 + a1), y2 = (long long)((int)st.i2 + a2);
  const long long w = (long long)st.cash - (long long)P.vari[0] * y1 - (long long)P.vari[1] * y2 - w0;
  if (y1 < 0 || y2 < 0 || y2 >= B.nr || w < 0 || w >= B.n2) {
    *oob = 1;
    return;
  }
  const long long idx = (y1 * B.nr + y2) * B.n2 + w;  // (the triples' box as a Lattice: i1 = y1, R - r0 = y2, i2 = W - w0)
  if (idx >= B.bits) {
    *oob = 1;
    return;
  }
  const unsigned int bit = 1u << (idx & 31);
  if (!(words[idx >> 5] & bit)) atomicOr(&words[idx >> 5], bit);
}

__global__ __launch_bounds__(256) void xr_triple_succ_kernel(MLParams P, Lattice L, long long w0, const Tuple* __restrict__ triples,
                                                             int64_t n_triples, const double2* __restrict__ dem,
                                                             unsigned int* __restrict__ words, int* __restrict__ oob) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (triple, demand pair)
  if (g >= n_triples * P.nd) return;
  const int64_t k = g / P.nd;
  const int j = (int)(g - k * P.nd);
  const Tuple tr = triples[k];  // (lattice_tuple of the triples' box: i1 = y1, cash = y2, i2 = W - w0)
  Tuple st;
  st.i1 = tr.i1;
  st.i2 = tr.cash;
  st.q1 = st.q2 = 0.0;
  st.cash = (tr.i2 + (double)w0) + P.vari[0] * tr.i1 + P.vari[1] * tr.cash;
  const long long idx = lattice_index(L, xr_successor(P, st, st.i1, st.i2, dem[j].x, dem[j].y));
  if (idx < 0 || idx >= L.bits) {
    *oob = 1;
    return;
  }
  const unsigned int bit = 1u << (idx & 31);
  if (!(words[idx >> 5] & bit)) atomicOr(&words[idx >> 5], bit);
}

// ---- backward: one workgroup per state ------------------------------------------------------------
// The lead-time family's actions of one state (model 0), NI actions per lane (a = tid + 256 i), demand pairs in the OUTER loop: a
// pair's terms are read from LDS once per lane instead of once per cell (round 3: three LDS instructions per cell beside ~10
// vector ones -- 17.6 instructions per cell on the recorded instances, whose cells are almost all period T's), and every
// accumulator still takes its addends in the reference's order, demand index ascending (CashRecursionMultiLead.java:72-80):
// thisActionsValue of action a is acc[i], and trip j of the outer loop adds pair j's two terms to each of them.  Per cell in
// period T: the three additions of cash_increment, the product with p, the accumulation.
constexpr int kActPerLane = 10;  // Qbound up to 50 (2560 pairs)
template <int NI>
__device__ __forceinline__ void lead_actions(const MLParams& P, const Tuple& st, int64_t s, int tid, int NA, const DemandTerms* s_t,
                                             const double* s_p, const int* __restrict__ uid, const double* __restrict__ v_next,
                                             double* s_q) {
  double bi[NI], acc[NI];
  const int* urow[NI];
  // (i, j) of the lane's actions without a division per action (an emulated 32-bit division is ~25 instructions, more than the
  // 20 a four-pair action's cells take): one division for a = tid, then steps of 256
  const int step1 = 256 / P.qb, step2 = 256 - step1 * P.qb;
  int a1 = tid / P.qb, a2 = tid - a1 * P.qb;
  const int* urow0 = P.is_last ? nullptr : uid + (int64_t)s * NA * P.nd;  // (period T reads no successor)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int a = tid + 256 * i;
    const double oc = P.vari[0] * (double)a1 + P.vari[1] * (double)a2;
    const double before = st.cash - oc - P.overhead;
    bi[i] = before - ml_interest(P, before);
    acc[i] = 0.0;  // thisActionsValue, CashRecursionMultiLead.java:72-80
    // (ids are non-negative 32-bit ranks: zero-extended byte offsets from the table's base instead of sign-extended 64-bit
    // index arithmetic per cell)
    urow[i] = urow0 + (a < NA ? a : 0) * P.nd;
    a1 += step1;
    a2 += step2;
    if (a2 >= P.qb) {
      a2 -= P.qb;
      ++a1;
    }
  }
  const char* vb = reinterpret_cast<const char*>(v_next);
  if (P.is_last) {
    for (int j = 0; j < P.nd; ++j) {
      const DemandTerms t = s_t[j];
      const double p = s_p[j];
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i] += p * cash_increment(st, bi[i], t);
    }
  } else {
    for (int j = 0; j < P.nd; ++j) {
      const DemandTerms t = s_t[j];
      const double p = s_p[j];
      const double pg = p * P.discount;
      double v[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i)
        v[i] = tid + 256 * i < NA ? *reinterpret_cast<const double*>(vb + ((uint64_t)(uint32_t)urow[i][j] << 3)) : 0.0;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        acc[i] += p * cash_increment(st, bi[i], t);
        acc[i] += pg * v[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (tid + 256 * i < NA) s_q[tid + 256 * i] = acc[i];
}

__global__ __launch_bounds__(256) void backward_kernel(MLParams P, const Tuple* __restrict__ states, int64_t s_first,
                                                       int64_t n_states,
                                                       const double2* __restrict__ dem, const double* __restrict__ prob,
                                                       const double* __restrict__ v_next, const int* __restrict__ uid,
                                                       double* __restrict__ v_out, int* __restrict__ act_out,
                                                       unsigned long long* __restrict__ cell_count, Lattice L,
                                                       const uint2* __restrict__ lat_rank) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NA = P.qb * P.qb;
  double* s_q = reinterpret_cast<double*>(smem);                       // Q(s, a)
  DemandTerms* s_t = reinterpret_cast<DemandTerms*>(s_q + NA);         // per demand pair
  double* s_p = reinterpret_cast<double*>(s_t + P.nd);
  const int64_t s = s_first + blockIdx.x;
  if (s >= n_states) return;
  const Tuple st = states[s];
  const int tid = threadIdx.x;
  for (int j = tid; j < P.nd; j += 256) {
    s_t[j] = demand_terms(P, st, dem[j].x, dem[j].y);
    s_p[j] = prob[j];
  }
  __syncthreads();
  // Model 1: the state is offered only the pairs with variCost . (i, j) < cash + 0.1 -- for each i a prefix of the
  // j's (the test is monotone in j).  The lanes walk that list itself, in the reference's order (i outer, j inner),
  // not the Qbound x Qbound box: s_off[i] = number of offered pairs before row i, Q values stored compactly.
  int* s_off = reinterpret_cast<int*>(s_p + P.nd);
  int n_offered = NA;
  if (P.model == 1) {
    if (tid < P.qb) {
      int n2 = 0;
      while (n2 < P.qb && mc_feasible(P, st, tid, n2)) ++n2;
      s_off[tid + 1] = n2;
    }
    __syncthreads();
    if (tid == 0) {
      s_off[0] = 0;
      for (int i = 0; i < P.qb; ++i) s_off[i + 1] += s_off[i];
    }
    __syncthreads();
    n_offered = s_off[P.qb];
  }
  auto offered_pair = [&](int k, int& a1, int& a2) {  // k-th offered pair: the last row with s_off[row] <= k
    int lo_r = 0, hi_r = P.qb - 1;
    while (lo_r < hi_r) {
      const int mid = (lo_r + hi_r + 1) >> 1;
      if (s_off[mid] <= k) lo_r = mid; else hi_r = mid - 1;
    }
    a1 = lo_r;
    a2 = k - s_off[lo_r];
  };
  for (int k = tid; k < n_offered && P.model == 1; k += 256) {
    int a1, a2;
    offered_pair(k, a1, a2);
    const int a = a1 * P.qb + a2;
    double acc = 0.0;  // thisActionsValue, CashRecursionMulti.java:97-105
#pragma unroll 4
    for (int j = 0; j < P.nd; ++j) {
      const double p = s_p[j];
      acc += p * mc_immediate(P, st, a1, a2, dem[j].x, dem[j].y);
      if (!P.is_last) {
        const int id = lat_rank ? lattice_rank(lat_rank, lattice_index(L, mc_successor(P, st, a1, a2, dem[j].x, dem[j].y)))
                                 : uid[((int64_t)s * NA + a) * P.nd + j];
        acc += p * P.discount * v_next[id];
      }
    }
    s_q[k] = acc;
  }
  if (P.model == 1 && tid == 0 && cell_count) atomicAdd(cell_count, (unsigned long long)n_offered * (unsigned long long)P.nd);
  for (int a = tid; a < NA && P.model == 2; a += 256) {
    const int a1 = a / P.qb, a2 = a - a1 * P.qb;
    const double y1 = (double)((int)st.i1 + a1), y2 = (double)((int)st.i2 + a2);
    double acc = 0.0;  // thisActionsValue, CashRecursionMultiXR.java:76-86
#pragma unroll 4
    for (int j = 0; j < P.nd; ++j) {
      const double p = s_p[j];
      acc += p * xr_immediate(P, st, y1, y2, dem[j].x, dem[j].y);
      if (!P.is_last) {
        const int id = lat_rank ? lattice_rank(lat_rank, lattice_index(L, xr_successor(P, st, y1, y2, dem[j].x, dem[j].y)))
                                 : uid[((int64_t)s * NA + a) * P.nd + j];
        acc += p * P.discount * v_next[id];
      }
    }
    s_q[a] = acc;
  }
  if (P.model == 0 && NA <= 256 * kActPerLane) {
    // (the lane's action count is a template argument: straight-line code per demand pair, no per-action guards)
    switch ((NA + 255) / 256) {
      case 1: lead_actions<1>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      case 2: lead_actions<2>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      case 3: lead_actions<3>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      case 4: lead_actions<4>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      case 5: lead_actions<5>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      case 6: lead_actions<6>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      case 7: lead_actions<7>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      case 8: lead_actions<8>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      case 9: lead_actions<9>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
      default: lead_actions<10>(P, st, s, tid, NA, s_t, s_p, uid, v_next, s_q); break;
    }
  } else
  for (int a = tid; a < NA && P.model == 0; a += 256) {  // (more than 256 * kActPerLane actions: one action at a time)
    const int a1 = a / P.qb, a2 = a - a1 * P.qb;
    const double oc = P.vari[0] * (double)a1 + P.vari[1] * (double)a2;
    const double before = st.cash - oc - P.overhead;
    const double bi = before - ml_interest(P, before);
    double acc = 0.0;  // thisActionsValue, CashRecursionMultiLead.java:72-80
    const int* urow = uid + ((int64_t)s * NA + a) * P.nd;
    const char* vb = reinterpret_cast<const char*>(v_next);
    if (P.is_last) {
      for (int j = 0; j < P.nd; ++j) acc += s_p[j] * cash_increment(st, bi, s_t[j]);
    } else {
      for (int j = 0; j < P.nd; ++j) {
        const double p = s_p[j];
        acc += p * cash_increment(st, bi, s_t[j]);
        acc += p * P.discount * *reinterpret_cast<const double*>(vb + ((uint64_t)(uint32_t)urow[j] << 3));
      }
    }
    s_q[a] = acc;
  }
  __syncthreads();
  // `if (actionValues[i] > val + 0.1)` in action order (CashRecursionMultiLead.java:82): a serial
  // scan, done 64 candidates at a time by the first wave (each pass jumps to the next improvement)
  if (tid < 64) {
    double val = -1.7976931348623157e308;
    int best = 0;  // new Actions(0, 0)
    for (int base = 0; base < n_offered; base += 64) {  // (n_offered == NA outside model 1)
      const int a = base + tid;
      const double q = a < n_offered ? s_q[a] : -1.7976931348623157e308;
      int from = 0;
      while (true) {
        const unsigned long long m = __ballot(a < n_offered && tid >= from && q > val + 0.1);
        if (!m) break;
        const int first = __ffsll((long long)m) - 1;
        val = __shfl(q, first, 64);
        best = base + first;
        from = first + 1;
      }
    }
    if (tid == 0) {
      if (P.model == 1) {  // position in the offered list -> (i, j)
        int a1, a2;
        offered_pair(best, a1, a2);
        best = a1 * P.qb + a2;
      }
      v_out[s] = val;
      act_out[s] = best;
    }
  }
}

// ---- backward, the two cash-constrained families (models 1 and 2) with a FACTORED demand list --------------------------------
// What a cell of these families computes splits by product: endInventory_k, revenue_k, the salvage term and the successor's
// inventory of product k read the k-th order quantity and the k-th demand only (MultiItemCash.java:79-118,
// MultiItemCashXR.java:108-148).  With the period's pair list factored into its distinct first and second demands (the lists
// GetPmfMulti builds are products of two marginals: 22 x 16 values for MultiItemCashXR.main's 352 pairs), a state's workgroup
// tabulates them once per (order quantity, distinct demand) -- 50 x (22 + 16) entries against 2500 x 352 cells -- and a cell is
// two table reads, the additions that join the two products, and the accumulation: 7 vector operations in period T where round
// 3's kernel executed the whole lambda per cell (17.6 instructions per cell; MultiItemCashXR with four periods: 11.4 s).  Every
// table entry is formed by the reference's expressions on the reference's operands, and a cell joins them in the reference's
// order, so the values are the same doubles.
struct FactSide {  // per (order index of one product, distinct demand of that product)
  double rev;      // price_k * (level_k - endInventory_k)
  double w;        // period T: salvage_k * endInventory_k; earlier: variCost_k * nextInventory_k (model 2's next R)
  long long lat;   // the successor's lattice-index share of this product (see lattice_index)
};
struct FactList {
  const double* u1;  // distinct first demands, distinct second demands
  const double* u2;
  const int* idx;    // per pair j: k1 | k2 << 16
  int nu1, nu2;
  int run;           // the list is made of runs of `run` pairs with one first index each (fact_run); 0: any other list
};

// The length of the list's runs of one first index, when the list is made of runs of equal length (the product lists GetPmfMulti
// builds: the second product's demands under every first one); 0 otherwise.
constexpr int kFactThreads = 512;  // eight waves share a state's tables: twice the waves per compute unit for the same LDS
constexpr int kFactNI = 5;         // order pairs a lane carries through a pass of the demand list
// LDS of a backward_fact_kernel workgroup (its carve-up): one pass of Q(s, .), the pmf and the pairs' index words, the tables
// (16 bytes an entry; before period T up to 8 more for the lattice share), the rows' offsets, the chunk maxima; slack for alignment
inline size_t fact_lds_bytes(int NA, int nd, int qb, size_t n_distinct, bool last, bool mark = false) {
  const size_t qn = mark ? 0 : (size_t)std::min(NA, kFactThreads * kFactNI);
  return qn * 8 + (size_t)nd * 12 + (size_t)qb * n_distinct * (last ? 16 : 24) + (size_t)(qb + 1) * 4 +
         (size_t)(kFactThreads * kFactNI / 64) * 8 + 32;
}

inline int fact_run(const std::vector<int>& idx) {
  if (idx.empty()) return 0;
  size_t run = 1;
  while (run < idx.size() && (idx[run] & 0xffff) == (idx[0] & 0xffff)) ++run;
  if (idx.size() % run != 0) return 0;
  for (size_t j = 0; j < idx.size(); j += run)
    for (size_t c = 1; c < run; ++c)
      if ((idx[j + c] & 0xffff) != (idx[j] & 0xffff)) return 0;
  return (int)run;
}

// V_{t+1} on the lattice: vdense[lattice index of state k] = v[k].  The backward pass of the bitmap path ranks every successor
// (a gather of the bitmap word and its prefix) and then gathers the value by that rank: two dependent reads per cell, which is
// what binds the long horizons (MultiItemCashXR with four periods: the 1.5e12 cells of period 3 took 8 of 11.4 s).  Where the
// lattice's box is small enough for 8 bytes per POINT (<= SDPGPU_MULTI_DENSE_GB, default 16 GB of the 288), the values are laid
// out by lattice index and a cell reads its successor's value directly.  Only reachable points are ever written or read.
__global__ __launch_bounds__(256) void dense_scatter_kernel(Lattice L, const Tuple* __restrict__ states, int64_t n,
                                                           const double* __restrict__ v, double* __restrict__ vdense) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n) vdense[lattice_index(L, states[k])] = v[k];
}

// LK: where a successor's value comes from in the periods before T -- 0: the rank stored per candidate by the sorted-candidates
// forward pass (uid), 1: the lattice's rank word, then the value by rank, 2: V_{t+1} laid out on the lattice itself.  A template
// argument, and the loop below forms the NI actions' reads of a demand pair first, issues them together and consumes them
// afterwards: with the three forms behind run-time branches every cell was its own basic block, one gather and one full wait
// each -- the not-last periods of the long horizons ran at 0.27e12 cells/s on dependent reads.
// I32: the lattice has fewer than 2^32 - 1 points and its R and i2 extents are below 2^24 (the recorded instances: 4.4e8 points):
// the successor's index is formed in 32-bit words -- 4-byte table entries, a 24-bit multiply-add -- where the 64-bit form costs
// a quarter-rate v_mad_u64_u32, two 2-word additions and 8-byte LDS reads per cell.
template <int MODEL, bool LAST, int LK = 0, bool I32 = false>
__global__ __launch_bounds__(kFactThreads, 4) void backward_fact_kernel(MLParams P, const Tuple* __restrict__ states, int64_t s_first,
                                                           int64_t n_states, FactList F, const double* __restrict__ prob,
                                                           const double* __restrict__ v_next, const int* __restrict__ uid,
                                                           double* __restrict__ v_out, int* __restrict__ act_out,
                                                           unsigned long long* __restrict__ cell_count, Lattice L,
                                                           const uint2* __restrict__ lat_rank, const double* __restrict__ vdense,
                                                           unsigned int* __restrict__ mark_words, int* __restrict__ mark_oob) {
  // LK == 3: the FORWARD pass on the lattice with the same tables -- no values, no probabilities: every (state, offered action,
  // demand pair) marks its successor's bit (lattice_mark_kernel evaluates the whole transition lambda per cell: 3.5 of the 7.5 s
  // of MultiItemCashXR's four periods)
  constexpr bool MARK = LK == 3;
  static_assert(!MARK || !LAST, "the forward pass has no period T");
  static_assert(!I32 || (!LAST && LK != 0), "the index width only matters where a lattice index is formed");
  using LI = std::conditional_t<I32, unsigned int, long long>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NA = P.qb * P.qb;
  // Q(s, a) of ONE pass of the action range (kFactThreads x NI order pairs); the tolerance scan runs behind every pass, its `val`
  // carried on.  Held for the whole range (8 B x Qbound^2: 80 KB at MultiItemCash.main's Qbound 100) the array left room for one
  // workgroup a compute unit -- two waves per SIMD, the vector unit 0.60 busy.
  const int QN = (LK == 3) ? 0 : (NA < kFactThreads * kFactNI ? NA : kFactThreads * kFactNI);  // (the forward pass keeps no values)
  double* s_q = reinterpret_cast<double*>(smem);
  double* s_p = s_q + QN;
  // the tables as separate arrays, [distinct demand][order index]: the lanes of a wave are consecutive second order quantities
  // (and at most two first ones), so a read of a pair's entries is 64 consecutive 8-byte words -- as [order][demand] records of
  // 24 bytes the lanes were 384 bytes apart, two LDS banks for 64 lanes (measured: 6x slower than round 3's kernel)
  const int n_e = P.qb * (F.nu1 + F.nu2);
  // (rev and w of an entry side by side: one 16-byte read and one address per entry -- as two arrays the second product's entries
  // of model 1, read per cell, cost three address operations and two reads each: 12.6 vector operations per period-T cell
  // counted where the arithmetic is 7)
  double2* s_rw = reinterpret_cast<double2*>(s_p + ((P.nd + QN + 1) & ~1) - QN);  // [nu1][qb] then [nu2][qb], 16-byte aligned
  LI* s_lat = reinterpret_cast<LI*>(s_rw + n_e);
  int* s_idx = reinterpret_cast<int*>(s_lat + (LAST ? 0 : n_e));
  int* s_off = s_idx + P.nd;
  // maxima of a pass's 64-value chunks (see the scan): kFactThreads * kFactNI / 64 doubles, 8-byte aligned
  double* s_max = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(s_off + P.qb + 1) + 7) & ~(uintptr_t)7);
  const int64_t s = s_first + blockIdx.x;
  if (s >= n_states) return;
  const Tuple st = states[s];
  const int tid = threadIdx.x;
  for (int j = tid; j < P.nd; j += kFactThreads) {
    s_p[j] = prob[j];
    s_idx[j] = F.idx[j];
  }
  for (int e = tid; e < P.qb * (F.nu1 + F.nu2); e += kFactThreads) {
    const bool first = e < P.qb * F.nu1;
    const int ee = first ? e : e - P.qb * F.nu1;
    const int nu = first ? F.nu1 : F.nu2;
    const int ai = ee / nu, k = ee - ai * nu;
    const double d = first ? F.u1[k] : F.u2[k];
    const double x = first ? st.i1 : st.i2;
    // the level the demand meets: model 1 x_k + action_k (MultiItemCash.java:84-85), model 2 the order-up-to level
    // y_k = (int) x_k + i (MultiItemCashXR.java:95-103, :113-114)
    const double level = MODEL == 2 ? (double)((int)x + ai) : x + (double)ai;
    const double end = jmax(0.0, level - d);
    FactSide t;
    t.rev = P.price[first ? 0 : 1] * (level - end);
    t.lat = 0;
    const int slot = (first ? 0 : P.qb * F.nu1) + k * P.qb + ai;
    if constexpr (LAST) {
      t.w = P.sal[first ? 0 : 1] * end;
    } else {
      // the transition's own end inventory: upper clamp on product 1 only, lower clamp on product 2 only, (int) cast
      double n = end;
      if (first)
        n = n > P.max_inventory ? P.max_inventory : n;
      else
        n = n < P.min_inventory ? P.min_inventory : n;
      n = (double)(int)n;
      t.w = P.vari[first ? 0 : 1] * n;  // (model 2: nextR = (int) nextCash + variCost[0] * n1 + variCost[1] * n2)
      const long long i = (long long)n;
      t.lat = first ? (i * L.nr + L.skew1 * i) * L.n2 : L.skew2 * i * L.n2 + i;
    }
    s_rw[slot] = make_double2(t.rev, t.w);
    if constexpr (!LAST) s_lat[slot] = (LI)t.lat;  // (I32: modulo 2^32 -- the sum of the three shares is below it)
  }
  __syncthreads();
  int n_offered = NA;
  if constexpr (MODEL == 1) {  // the offered prefix of every row of the action box (see backward_kernel)
    if (tid < P.qb) {
      int n2 = 0;
      if (P.vari[1] >= 0) {
        // the first second order quantity the row does not offer, by bisection: with a non-negative unit cost the ordering cost
        // fl(fl(c1 a1) + fl(c2 a2)) never decreases in a2 (rounding is monotone), so the offered quantities are a prefix and
        // the walk below stops at the same one -- in 7 tests instead of up to Qbound (MultiItemCash.main: 100)
        int lo_f = 0, hi_f = P.qb;
        while (lo_f < hi_f) {
          const int mid = (lo_f + hi_f) >> 1;
          if (mc_feasible(P, st, tid, mid)) lo_f = mid + 1; else hi_f = mid;
        }
        n2 = lo_f;
      } else {
        while (n2 < P.qb && mc_feasible(P, st, tid, n2)) ++n2;
      }
      s_off[tid + 1] = n2;
    }
    __syncthreads();
    if (tid < 64) {  // running sum of the rows' counts: the first wave, 64 rows a step (one thread adding Qbound entries one
                     // after the other was a chain of Qbound dependent LDS round trips per state)
      int carry = 0;
      for (int base_r = 0; base_r < P.qb; base_r += 64) {
        const int r = base_r + tid;
        int v = r < P.qb ? s_off[r + 1] : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int u = __shfl_up(v, off, 64);
          if (tid >= off) v += u;
        }
        if (r < P.qb) s_off[r + 1] = carry + v;
        carry += __shfl(v, 63, 64);
      }
      if (tid == 0) s_off[0] = 0;
    }
    __syncthreads();
    n_offered = s_off[P.qb];
    if (tid == 0 && cell_count) atomicAdd(cell_count, (unsigned long long)n_offered * (unsigned long long)P.nd);
  }
  auto offered_pair = [&](int k, int& a1, int& a2) {
    int lo_r = 0, hi_r = P.qb - 1;
    while (lo_r < hi_r) {
      const int mid = (lo_r + hi_r + 1) >> 1;
      if (s_off[mid] <= k) lo_r = mid; else hi_r = mid - 1;
    }
    a1 = lo_r;
    a2 = k - s_off[lo_r];
  };
  const double ini_cash = MODEL == 2 ? st.cash - P.vari[0] * st.i1 - P.vari[1] * st.i2 : st.cash;  // (model 2: initialCash)
  const double pdisc = P.discount;
  // NI actions of a lane at a time (k = tid + 256 i), demand pairs in the outer loop: the pair's index word and probability are
  // read once per lane, every accumulator takes its addends demand index ascending
  constexpr int NI = kFactNI;  // (512 threads x 5: the 2500 order pairs of Qbound 50 in one pass)
  // Model 2's action box is regular (a = i * Qbound + j): with the lanes' actions a multiple of Qbound apart (TS = the largest
  // multiple of Qbound that fits the workgroup; lanes beyond it idle: 12 of 512 at Qbound 50) a lane's NI actions share the SECOND
  // order index, so the second product's table entries of a demand pair are read once per lane instead of once per cell -- 12
  // LDS reads per pair and lane instead of 20 in period T, where those reads are what binds (2.5e12 cells/s at four per cell).
  constexpr bool SAME2 = MODEL == 2;
  const int TS = (SAME2 && P.qb <= kFactThreads) ? (kFactThreads / P.qb) * P.qb : kFactThreads;
  const bool lane_on = tid < TS;
  double scan_val = -1.7976931348623157e308;  // (the first wave's: `val` and the best action so far of the tolerance scan)
  int scan_best = 0;
  for (int k0 = 0; k0 < n_offered; k0 += TS * NI) {
    int r1[NI], r2[NI];  // slot of (distinct demand 0, this action's order index): + k * qb per pair
    double base[NI], acc[NI];
    const int* urow[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = k0 + tid + TS * i;
      int a1 = 0, a2 = 0;
      if (k < n_offered && lane_on) {
        if constexpr (MODEL == 1) {
          offered_pair(k, a1, a2);
        } else {
          a1 = k / P.qb;
          a2 = k - a1 * P.qb;
        }
      }
      r1[i] = a1;
      r2[i] = P.qb * F.nu1 + a2;
      if constexpr (MODEL == 1) {
        const double orderingCost1 = P.vari[0] * (double)a1, orderingCost2 = P.vari[1] * (double)a2;
        base[i] = orderingCost1 + orderingCost2;  // orderingCosts
      } else {
        const double y1 = (double)((int)st.i1 + a1), y2 = (double)((int)st.i2 + a2);
        const double orderingCostY1 = P.vari[0] * y1, orderingCostY2 = P.vari[1] * y2;
        base[i] = P.one_minus_deposit * (st.cash - (orderingCostY1 + orderingCostY2));
      }
      acc[i] = 0.0;
      urow[i] = uid ? uid + ((int64_t)s * NA + (a1 * P.qb + a2)) * P.nd : nullptr;
    }
    // A product's table entries of a demand pair depend on that product's distinct demand only, and the lists GetPmfMulti builds
    // run through the second product's demands under every first one (22 x 16: the first index changes every 16th pair).  The
    // first product's entries are kept in registers and read again only when a pair's first index differs from the previous
    // pair's -- a wave-uniform test; in period T, where the LDS reads were what bound the kernel (12 reads of 8 bytes per pair
    // and five cells: 6.07e12 cells in 1.84 s), a pair then costs the reads of the second product's entries alone.  The pairs
    // are taken CH at a time: where the CH pairs share their first index (always, inside a run of such a list) their index
    // words, probabilities and second-product entries are read together, ahead of the arithmetic -- one LDS round trip per CH
    // pairs instead of two per pair; any other group of pairs goes one by one.  The pairs are walked, and every accumulator fed,
    // in list order either way.
    constexpr int N2 = SAME2 ? 1 : NI;       // (SAME2: r2[i] is the same slot for every i -- except past the last action,
    constexpr int CH = 4;                    //  where it is slot 0's: harmless, unused)
    int cur1 = -1;
    double c_rev1[NI], c_w1[NI];
    [[maybe_unused]] LI c_lat1[NI];
    auto load1 = [&](int k1) {
      cur1 = k1;
      const int o1 = k1 * P.qb;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const double2 e = s_rw[r1[i] + o1];
        c_rev1[i] = e.x;
        c_w1[i] = e.y;
        if constexpr (!LAST && LK != 0) c_lat1[i] = s_lat[r1[i] + o1];
      }
    };
    auto load2 = [&](int k2, double (&rev2)[N2], double (&w2)[N2], LI (&lat2)[N2]) {
      const int o2 = k2 * P.qb;
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        const double2 e = s_rw[r2[i] + o2];
        rev2[i] = e.x;
        w2[i] = e.y;
        if constexpr (!LAST && LK != 0) lat2[i] = s_lat[r2[i] + o2]; else lat2[i] = 0;
      }
    };
    // the NI cells of demand pair j
    auto pair_body = [&](int j, double p, const double (&c_rev2)[N2], const double (&c_w2)[N2], const LI (&c_lat2)[N2]) {
      [[maybe_unused]] const double pg = p * pdisc;
      [[maybe_unused]] LI li[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const double rev1 = c_rev1[i], rev2 = c_rev2[SAME2 ? 0 : i];
        const double w1 = c_w1[i], w2 = c_w2[SAME2 ? 0 : i];
        const double revenue = rev1 + rev2;
        // Before period T the salvage term the reference adds is the constant 0.0, and x + 0.0 is x for every x but -0.0, which
        // it turns into +0.0: the sign of a zero immediate value reaches neither the accumulator (it starts at +0.0, and
        // +0.0 + -0.0 = +0.0) nor the successor (a zero of either sign casts to 0) -- the addition is left out there.
        double imm;
        if constexpr (LAST) {
          const double sal = w1 + w2;
          if constexpr (MODEL == 1)
            imm = revenue - base[i] + sal;  // MultiItemCash.java:98
          else
            imm = revenue + base[i] + sal - ini_cash;  // MultiItemCashXR.java:127
        } else {
          if constexpr (MODEL == 1)
            imm = revenue - base[i];
          else
            imm = revenue + base[i] - ini_cash;
        }
        if constexpr (!MARK) acc[i] += p * imm;
        if constexpr (!LAST && LK != 0) {
          double nc = ini_cash + imm;  // (model 1: s.cash + immediate; model 2: initialCash + immediate)
          // `nextCash > maxCash ? maxCash : nextCash`, then the lower bound the same way: v_min_f64 / v_max_f64 give the same
          // doubles for finite operands (a +-0 tie is the only difference, and the (int) cast below maps both to 0)
          nc = __builtin_fmax(__builtin_fmin(nc, P.max_cash), P.min_cash);
          int r;
          if constexpr (MODEL == 2) {
            // nextR = (int) nextCash + variCost . nextInventory (inside the lattice's box: below 2^31).  I32 (the cash bounds are
            // below 2^31 too): (double)(int) nc is trunc(nc) but for the sign of a zero, which the additions and the cast absorb
            const double whole = I32 ? __builtin_trunc(nc) : (double)(int)nc;
            r = (int)(whole + w1 + w2);
          } else {
            r = (int)nc;
          }
          const LI lat12 = c_lat1[i] + c_lat2[SAME2 ? 0 : i];
          if constexpr (I32) {
            const unsigned int dr = (unsigned int)(r - (int)L.r0);  // (a successor below r0 wraps to a large value)
            const unsigned int idx = lat12 + __umul24(dr, (unsigned int)L.n2);
            // (the forward pass also has to notice a successor outside the box: no index, the all-ones word)
            li[i] = (MARK && dr >= (unsigned int)L.nr) ? 0xffffffffu : idx;
          } else if constexpr (MARK) {
            li[i] = (r < L.r0 || r - L.r0 >= L.nr) ? -1 : lat12 + (long long)(r - L.r0) * L.n2;
          } else {
            li[i] = lat12 + (long long)((unsigned long long)(unsigned)(r - (int)L.r0) * (unsigned)L.n2);
          }
        }
      }
      if constexpr (MARK) {
        // (the words of the NI successors are read together, then tested: most candidates are already marked)
        unsigned int wd[NI];
        bool in_box[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const bool offered = lane_on && k0 + tid + TS * i < n_offered;
          if constexpr (I32)
            in_box[i] = offered && li[i] < (unsigned int)L.bits;
          else
            in_box[i] = offered && li[i] >= 0 && li[i] < L.bits;
          if (offered && !in_box[i]) *mark_oob = 1;  // the host's box was too small: reported, never silently dropped
          wd[i] = in_box[i] ? mark_words[li[i] >> 5] : 0xffffffffu;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const unsigned int bit = 1u << (li[i] & 31);
          if (in_box[i] && !(wd[i] & bit)) atomicOr(&mark_words[li[i] >> 5], bit);
        }
      } else if constexpr (!LAST) {
        double v[NI];
        if constexpr (LK == 2) {
#pragma unroll
          for (int i = 0; i < NI; ++i) v[i] = vdense[li[i]];
        } else if constexpr (LK == 1) {
          uint2 wp[NI];
#pragma unroll
          for (int i = 0; i < NI; ++i) wp[i] = lat_rank[li[i] >> 5];
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const unsigned int below = wp[i].x & ((1u << (li[i] & 31)) - 1u);
            v[i] = v_next[(int)(wp[i].y + __popc(below))];
          }
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) v[i] = v_next[urow[i][j]];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] += pg * v[i];
      }
    };
    if (F.run > 0) {  // (the host's test of the list: runs of F.run pairs, each with one first index)
      for (int j0 = 0; j0 < P.nd; j0 += F.run) {
        load1(__builtin_amdgcn_readfirstlane(s_idx[j0]) & 0xffff);
        int jj = 0;
        for (; jj + CH <= F.run; jj += CH) {
          const int j = j0 + jj;
          int kk[CH];
          double pp[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            kk[c] = __builtin_amdgcn_readfirstlane(s_idx[j + c]);
            pp[c] = s_p[j + c];
          }
          double rev2[CH][N2], w2[CH][N2];
          LI lat2[CH][N2];
#pragma unroll
          for (int c = 0; c < CH; ++c) load2(kk[c] >> 16, rev2[c], w2[c], lat2[c]);
#pragma unroll
          for (int c = 0; c < CH; ++c) pair_body(j + c, pp[c], rev2[c], w2[c], lat2[c]);
        }
        for (; jj < F.run; ++jj) {  // (a run that is not a multiple of CH pairs long: its last pairs one by one)
          const int j = j0 + jj;
          const int kk = __builtin_amdgcn_readfirstlane(s_idx[j]);
          double rev2[N2], w2[N2];
          LI lat2[N2];
          load2(kk >> 16, rev2, w2, lat2);
          pair_body(j, s_p[j], rev2, w2, lat2);
        }
      }
    } else {
      for (int j = 0; j < P.nd; ++j) {
        const int kk = __builtin_amdgcn_readfirstlane(s_idx[j]);
        if ((kk & 0xffff) != cur1) load1(kk & 0xffff);
        double rev2[N2], w2[N2];
        LI lat2[N2];
        load2(kk >> 16, rev2, w2, lat2);
        pair_body(j, s_p[j], rev2, w2, lat2);
      }
    }
    if constexpr (!MARK) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int k = k0 + tid + TS * i;
        if (lane_on && k < n_offered) s_q[k - k0] = acc[i];
      }
      __syncthreads();
      // `if (actionValues[i] > val + 0.1)` in action order (as in backward_kernel) over this pass's values, in two steps.  `val`
      // only grows, so a chunk of 64 values whose largest is not above val + 0.1 when the scan reaches it holds no value that is:
      // every wave forms the maxima of its share of the chunks, and the first wave then walks the CHUNKS in order -- 64 maxima a
      // step -- opening only those that can move `val`.  (Opening every chunk had one wave walk all the action values while
      // seven waited: a third of a state's time at MultiItemCash.main's 10000 order pairs.)
      const int n_pass = (n_offered - k0) < TS * NI ? (n_offered - k0) : TS * NI;
      const int n_chunks = (n_pass + 63) >> 6;
      {
        const int wave_id = tid >> 6, lane_id = tid & 63;
        for (int c = wave_id; c < n_chunks; c += kFactThreads / 64) {
          const int a = c * 64 + lane_id;
          double m = a < n_pass ? s_q[a] : -1.7976931348623157e308;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(m, off, 64);
            m = o > m ? o : m;
          }
          if (lane_id == 0) s_max[c] = m;
        }
      }
      __syncthreads();
      if (tid < 64) {
        for (int base_c = 0; base_c < n_chunks; base_c += 64) {
          const int cc = base_c + tid;
          const double mc = cc < n_chunks ? s_max[cc] : -1.7976931348623157e308;
          int from_c = 0;
          while (true) {
            const unsigned long long open = __ballot(cc < n_chunks && tid >= from_c && mc > scan_val + 0.1);
            if (!open) break;
            const int first_c = __ffsll((long long)open) - 1;
            const int base_k = (base_c + first_c) * 64;
            const int a = base_k + tid;
            const double q = a < n_pass ? s_q[a] : -1.7976931348623157e308;
            int from = 0;
            while (true) {
              const unsigned long long m = __ballot(a < n_pass && tid >= from && q > scan_val + 0.1);
              if (!m) break;
              const int first = __ffsll((long long)m) - 1;
              scan_val = __shfl(q, first, 64);
              scan_best = k0 + base_k + first;
              from = first + 1;
            }
            from_c = first_c + 1;
          }
        }
      }
      __syncthreads();  // (the next pass writes s_q and s_max again)
    }
  }
  if constexpr (MARK) return;
  if (tid < 64) {
    const double val = scan_val;
    int best = scan_best;
    if (tid == 0) {
      if constexpr (MODEL == 1) {
        int a1, a2;
        offered_pair(best, a1, a2);
        best = a1 * P.qb + a2;
      }
      v_out[s] = val;
      act_out[s] = best;
    }
  }
}

// ---- backward, lead-time family (model 0): ONE WAVE per state ------------------------------------------------------------
// Round 3's form -- a 256-thread workgroup per state, Q(s, .) in LDS, two barriers, then the first wave alone scanning the 2500
// action values while three waves hold their registers -- ran the recorded instances (1.7e7 period-T states x 2500 order pairs x
// 4-9 demand pairs) at 40 % of the vector issue rate: the 5 operations a period-T cell costs were under half of what a state
// executed.  Here a wave owns a state: lane l holds the order pairs a = l + 64 i, so chunk i of the `> val + 0.1` scan
// (CashRecursionMultiLead.java:82, serial in action order) IS the 64 lanes' accumulators acc[i] -- no Q array, no LDS traffic, no
// workgroup barrier, no idle wave.  The action range is walked in passes of kLeadChunks chunks (registers), the scan's `val`
// carried from pass to pass.  Per (lane, chunk) the order quantities step by 64 without a division.  Every accumulator takes its
// addends demand index ascending, exactly as before.
constexpr int kLeadChunks = 10;
// orderingCosts of every order pair (see backward_lead_wave_kernel)
__global__ __launch_bounds__(256) void order_cost_kernel(MLParams P, double* __restrict__ oc_tab) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= P.qb * P.qb) return;
  const int a1 = a / P.qb, a2 = a - a1 * P.qb;
  oc_tab[a] = P.vari[0] * (double)a1 + P.vari[1] * (double)a2;
}

template <bool LAST>
__global__ __launch_bounds__(256) void backward_lead_wave_kernel(MLParams P, const Tuple* __restrict__ states, int64_t s_first,
                                                                int64_t n_states, const double2* __restrict__ dem,
                                                                const double* __restrict__ prob, const double* __restrict__ v_next,
                                                                const int* __restrict__ uid, double* __restrict__ v_out,
                                                                int* __restrict__ act_out, const double* __restrict__ oc_tab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NA = P.qb * P.qb;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  DemandTerms* s_t = reinterpret_cast<DemandTerms*>(smem) + (size_t)wave * P.nd;  // this wave's state: per demand pair
  double* s_p = reinterpret_cast<double*>(reinterpret_cast<DemandTerms*>(smem) + (size_t)4 * P.nd);
  const int64_t s = s_first + (int64_t)blockIdx.x * 4 + wave;
  // (the pmf is the same for the four states: every wave writes the same values, nobody waits for anybody)
  for (int j = lane; j < P.nd; j += 64) s_p[j] = prob[j];
  if (s >= n_states) return;
  const Tuple st = states[s];
  for (int j = lane; j < P.nd; j += 64) s_t[j] = demand_terms(P, st, dem[j].x, dem[j].y);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own LDS writes have landed
  // oc_tab[a] = variCost[0] * i + variCost[1] * j of order pair a = i * Qbound + j (MultiProductLeadtime.java:184), formed once
  // per solve by order_cost_kernel with these very operations: a state's setup per order pair is then one coalesced load instead
  // of two conversions, two products, a sum and the stepping of (i, j) -- a third of the ~30 instructions it took
  [[maybe_unused]] const int* urow0 = LAST ? nullptr : uid + (int64_t)s * NA * P.nd;  // (period T reads no successor)
  const char* vb = reinterpret_cast<const char*>(v_next);
  double val = -1.7976931348623157e308;
  int best = 0;  // new Actions(0, 0)
  for (int c0 = 0; c0 * 64 < NA; c0 += kLeadChunks) {
    double bi[kLeadChunks], acc[kLeadChunks];
    [[maybe_unused]] const int* urow[kLeadChunks];
#pragma unroll
    for (int i = 0; i < kLeadChunks; ++i) {
      const int a = (c0 + i) * 64 + lane;
      const double oc = oc_tab[a < NA ? a : NA - 1];
      const double before = st.cash - oc - P.overhead;
      bi[i] = before - ml_interest(P, before);
      acc[i] = 0.0;  // thisActionsValue, CashRecursionMultiLead.java:72-80
      if constexpr (!LAST) urow[i] = urow0 + (a < NA ? a : 0) * P.nd;
    }
    if constexpr (LAST) {
      for (int j = 0; j < P.nd; ++j) {
        const DemandTerms t = s_t[j];
        const double p = s_p[j];
#pragma unroll
        for (int i = 0; i < kLeadChunks; ++i) acc[i] += p * cash_increment(st, bi[i], t);
      }
    } else {
      for (int j = 0; j < P.nd; ++j) {
        const DemandTerms t = s_t[j];
        const double p = s_p[j];
        const double pg = p * P.discount;
        double v[kLeadChunks];
#pragma unroll
        for (int i = 0; i < kLeadChunks; ++i)
          v[i] = (c0 + i) * 64 + lane < NA ? *reinterpret_cast<const double*>(vb + ((uint64_t)(uint32_t)urow[i][j] << 3)) : 0.0;
#pragma unroll
        for (int i = 0; i < kLeadChunks; ++i) {
          acc[i] += p * cash_increment(st, bi[i], t);
          acc[i] += pg * v[i];
        }
      }
    }
    // `if (actionValues[i] > val + 0.1)` in action order (:82): chunk by chunk, each pass of the inner loop jumps to the next
    // improvement of the chunk
#pragma unroll
    for (int i = 0; i < kLeadChunks; ++i) {
      const int base = (c0 + i) * 64;
      const bool live = base + lane < NA;  // (a chunk past the last action: no lane is live, the first ballot is empty)
      const double q = acc[i];
      int from = 0;
      while (true) {
        const unsigned long long m = __ballot(live && lane >= from && q > val + 0.1);
        if (!m) break;
        const int first = __ffsll((long long)m) - 1;
        val = __shfl(q, first, 64);
        best = base + first;
        from = first + 1;
      }
    }
  }
  if (lane == 0) {
    v_out[s] = val;
    act_out[s] = best;
  }
}

#define ML_TRY(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      g_ml_error = std::string(#expr) + " (sdpgpu_sparse.hip:" + std::to_string(__LINE__) + "): " + hipGetErrorString(e_); \
      goto fail;                                                                         \
    }                                                                                    \
  } while (0)

// No C++ exception crosses the C ABI: the entry points of this file build host vectors of up to ~1.8e7
// tuples (the memo read-out) and a std::bad_alloc there must come back as a status code.  ML_GUARD wraps an entry point's
// body; sparse_solve catches inside itself as well, so that its device buffers are released on that path too.
//   SDPGPU_TEST_THROW = bad_alloc | runtime | other : thrown at the top of the guarded region (tests, no device needed)
//   SDPGPU_TEST_HOST_ALLOC_CAP = BYTES              : a host vector of the read-out larger than this throws bad_alloc
void test_throw_hook() {
  const char* e = std::getenv("SDPGPU_TEST_THROW");
  if (!e) return;
  if (std::strcmp(e, "bad_alloc") == 0) throw std::bad_alloc();
  if (std::strcmp(e, "runtime") == 0) throw std::runtime_error("injected (SDPGPU_TEST_THROW)");
  if (std::strcmp(e, "other") == 0) throw 42;
}

template <class V>
void checked_resize(V& v, size_t n) {
  if (const char* cap = std::getenv("SDPGPU_TEST_HOST_ALLOC_CAP"))
    if ((double)n * sizeof(typename V::value_type) > std::atof(cap)) throw std::bad_alloc();
  v.resize(n);
}

template <class Body>
int ml_guard(const char* who, Body&& body) {
  try {
    test_throw_hook();
    return body();
  } catch (const std::bad_alloc&) {
    g_ml_error = std::string(who) + ": host allocation failed (std::bad_alloc)";
    return SDPGPU_ERR_ALLOC;
  } catch (const std::exception& e) {
    g_ml_error = std::string(who) + ": internal error: " + e.what();
    return SDPGPU_ERR_INTERNAL;
  } catch (...) {
    g_ml_error = std::string(who) + ": internal error (unknown exception)";
    return SDPGPU_ERR_INTERNAL;
  }
}

constexpr size_t kLdsPerCUSparse = 160 * 1024;  // (gfx950; sdpgpu_internal.hpp has the same figure for the grid kernels)

struct SparseProblem {
  int T = 0;
  MLParams P{};
  double overhead[16] = {0};
  Tuple ini{};
  std::vector<int> off;  // demand pairs of period t+1: [off[t], off[t+1])
  std::vector<double2> dem;
  std::vector<double> prob;
  bool lattice_ok = false;  // models 1 / 2 with integer-lattice successors inside the box `lat`
  Lattice lat{};
};

int sparse_solve(const SparseProblem& sp, double* final_value, int32_t* q1, int32_t* q2, int64_t* states_per_period,
                 int64_t* cells, double* gpu_ms) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_ml_error = "no HIP device available; this library has no CPU path";
    return SDPGPU_ERR_DEVICE;
  }
  const int T = sp.T, NA = sp.P.qb * sp.P.qb;
  const int nd_all = sp.off[(size_t)T];
  const std::vector<double2>& h_dem = sp.dem;
  const std::vector<double>& h_prob = sp.prob;
  MLParams P = sp.P;
  const Tuple ini = sp.ini;

  std::vector<Tuple*> d_states((size_t)T, nullptr);
  std::vector<int*> d_uid((size_t)T, nullptr);
  // lattice path: bitmap and rank prefix of S_{t+1}, kept for the backward pass of period t
  std::vector<uint2*> d_lat_rank((size_t)T, nullptr);
  unsigned int *d_lat_words = nullptr, *d_lat_prefix = nullptr, *d_lat_counts = nullptr;
  unsigned int *d_tw = nullptr, *d_tc = nullptr, *d_tp = nullptr;  // the XR family's post-order triples: bitmap, counts, prefix
  uint2* d_tr_rank = nullptr;
  Tuple* d_triples = nullptr;
  int* d_oob = nullptr;
  const char* lat_env = std::getenv("SDPGPU_MULTI_LATTICE");
  const bool lat_force = lat_env && std::atoi(lat_env) == 1, lat_never = lat_env && std::atoi(lat_env) == 0;
  std::vector<int64_t> n_states((size_t)T, 0);
  double2* d_dem = nullptr;
  double* d_prob = nullptr;
  unsigned long long *d_hash = nullptr, *d_hash2 = nullptr;
  unsigned int *d_order = nullptr, *d_order2 = nullptr;
  int *d_head = nullptr, *d_rank = nullptr;
  void* d_tmp = nullptr;
  double *d_vcur = nullptr, *d_vnext = nullptr;
  int* d_act = nullptr;
  // the XR family's marking pass through post-order triples: exact data only (see xr_triple_mark_kernel)
  auto on_grid8 = [](double v) { return std::isfinite(v) && std::fabs(v) < 1048576.0 && v * 8.0 == std::floor(v * 8.0); };
  auto whole = [](double v) { return std::isfinite(v) && std::fabs(v) < 2147483648.0 && v == std::floor(v); };
  bool xr_exact = sp.lattice_ok && P.model == 2 && P.one_minus_deposit == 1.0 && on_grid8(P.price[0]) && on_grid8(P.price[1]) &&
                  whole(P.vari[0]) && whole(P.vari[1]) && P.vari[0] >= 0 && P.vari[1] >= 0 && whole(ini.i1) && whole(ini.i2) &&
                  whole(ini.cash) && ini.i1 >= 0 && ini.i2 >= 0;
  for (int j = 0; xr_exact && j < nd_all; ++j) xr_exact = whole(h_dem[(size_t)j].x) && whole(h_dem[(size_t)j].y);
  const char* tr_env = std::getenv("SDPGPU_MULTI_TRIPLES");
  const bool triples_force = tr_env && std::atoi(tr_env) == 1, triples_never = tr_env && std::atoi(tr_env) == 0;
  // backward_fact_kernel's 32-bit index form (SDPGPU_MULTI_I32=0: the 64-bit form everywhere)
  const bool lat_i32 = sp.lattice_ok && sp.lat.bits < (1LL << 31) && sp.lat.nr < (1LL << 24) && sp.lat.n2 < (1LL << 24) &&
                       std::fabs(P.min_cash) < 2147483648.0 && std::fabs(P.max_cash) < 2147483648.0 &&
                       !(std::getenv("SDPGPU_MULTI_I32") && std::atoi(std::getenv("SDPGPU_MULTI_I32")) == 0);
  unsigned long long* d_cells = nullptr;
  double* d_oc = nullptr;      // backward_lead_wave_kernel: orderingCosts of every order pair
  double* d_vdense = nullptr;  // backward_fact_kernel: V_{t+1} by lattice index (dense_scatter_kernel)
  char* d_fact = nullptr;  // backward_fact_kernel: the period's distinct demands and the pairs' index words
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int64_t total_cells = 0;
  int rc = SDPGPU_ERR_DEVICE;
  bool table_rows_ok = false;
  std::vector<int64_t> table_off((size_t)T, 0);
  std::vector<Tuple> h_tuples;
  std::vector<int> h_acts;
  try {
    ML_TRY(hipEventCreate(&ev0));
    ML_TRY(hipEventCreate(&ev1));
    ML_TRY(hipMalloc((void**)&d_dem, (size_t)nd_all * sizeof(double2)));
    ML_TRY(hipMalloc((void**)&d_prob, (size_t)nd_all * sizeof(double)));
    ML_TRY(hipMemcpy(d_dem, h_dem.data(), (size_t)nd_all * sizeof(double2), hipMemcpyHostToDevice));
    ML_TRY(hipMemcpy(d_prob, h_prob.data(), (size_t)nd_all * sizeof(double), hipMemcpyHostToDevice));
    ML_TRY(hipMalloc((void**)&d_cells, 8));
    ML_TRY(hipMemset(d_cells, 0, 8));
    ML_TRY(hipMalloc((void**)&d_states[0], sizeof(Tuple)));
    ML_TRY(hipMemcpy(d_states[0], &ini, sizeof(Tuple), hipMemcpyHostToDevice));
    n_states[0] = 1;
    ML_TRY(hipEventRecord(ev0, 0));
    // ---- forward ----
    for (int t = 0; t + 1 < T; ++t) {
      const int nd = sp.off[(size_t)t + 1] - sp.off[(size_t)t];
      const double2* dem_t = d_dem + sp.off[(size_t)t];
      P.nd = nd;
      const int64_t nc = n_states[t] * NA * nd;
      if (sp.lattice_ok && !lat_never && (lat_force || nc >= 2000000000LL)) {
        // ---- lattice path: no per-candidate storage ----
        const Lattice L = sp.lat;
        const long long n_words = (L.bits + 31) / 32;
        P.overhead = sp.overhead[t];
        P.is_last = 0;
        ML_TRY(hipMalloc((void**)&d_lat_words, (size_t)n_words * 4));
        ML_TRY(hipMemset(d_lat_words, 0, (size_t)n_words * 4));
        ML_TRY(hipMalloc((void**)&d_lat_prefix, (size_t)n_words * 4));
        ML_TRY(hipMalloc((void**)&d_lat_rank[t], (size_t)n_words * 8));
        ML_TRY(hipMalloc((void**)&d_lat_counts, (size_t)n_words * 4));
        if (!d_oob) {
          ML_TRY(hipMalloc((void**)&d_oob, 4));
          ML_TRY(hipMemset(d_oob, 0, 4));
        }
        bool marked = false;
        // the XR family with exact data: the distinct post-order triples mark the successors (xr_triple_*_kernel);
        // SDPGPU_MULTI_TRIPLES=0 never, =1 whenever the data allow it (default: periods of 2e10 candidates and more)
        if (xr_exact && !triples_never && (triples_force || (double)nc >= 2e10)) {
          const long long n1 = L.bits / (L.nr * L.n2);
          const long long ny1 = n1 + P.qb, ny2 = L.n2 + P.qb;
          const long long cmax = (long long)P.vari[0] * (ny1 - 1) + (long long)P.vari[1] * (ny2 - 1);
          const long long w0 = L.r0 - cmax;
          Lattice B;
          B.n2 = L.nr + cmax;  // W - w0
          B.nr = ny2;          // y2
          B.r0 = 0;
          B.skew1 = B.skew2 = 0;
          const double tb = (double)ny1 * (double)ny2 * (double)B.n2;
          B.bits = tb < 1.6e10 ? (long long)tb : 0;
          if (B.bits > 0) {
            const long long n_tw = (B.bits + 31) / 32;
            {
              ML_TRY(hipMalloc((void**)&d_tw, (size_t)n_tw * 4));
              ML_TRY(hipMemset(d_tw, 0, (size_t)n_tw * 4));
              ML_TRY(hipMalloc((void**)&d_tc, (size_t)n_tw * 4));
              ML_TRY(hipMalloc((void**)&d_tp, (size_t)n_tw * 4));
              const int64_t per_batch_s = std::max<int64_t>(1, ((int64_t)1 << 30) / NA);
              for (int64_t first = 0; first < n_states[t]; first += per_batch_s) {
                const int64_t ns = std::min<int64_t>(per_batch_s, n_states[t] - first);
                hipLaunchKernelGGL(xr_triple_mark_kernel, dim3((unsigned)((ns * NA + 255) / 256)), dim3(256), 0, 0, P, B, w0,
                                   d_states[t] + first, ns, d_tw, d_oob);
                ML_TRY(hipGetLastError());
              }
              const unsigned gtw = (unsigned)((n_tw + 255) / 256);
              hipLaunchKernelGGL(lattice_popc_kernel, dim3(gtw), dim3(256), 0, 0, d_tw, n_tw, d_tc);
              ML_TRY(hipGetLastError());
              size_t tmp_b = 0;
              ML_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_b, d_tc, d_tp, (int)n_tw));
              ML_TRY(hipMalloc(&d_tmp, tmp_b));
              ML_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_b, d_tc, d_tp, (int)n_tw));
              ML_TRY(hipFree(d_tmp));
              d_tmp = nullptr;
              unsigned int lp = 0, lc = 0;
              ML_TRY(hipMemcpy(&lp, d_tp + (n_tw - 1), 4, hipMemcpyDeviceToHost));
              ML_TRY(hipMemcpy(&lc, d_tc + (n_tw - 1), 4, hipMemcpyDeviceToHost));
              const int64_t n_tr = (int64_t)lp + lc;
              // (nothing has touched the successors' bitmap yet: where the triples do not fold the pairs at least four to one, or
              // their list would not fit comfortably, the per-cell marking pass below serves the period)
              const bool worth = n_tr < 2000000000LL && (double)n_tr * 4.0 <= (double)n_states[t] * (double)NA &&
                                 (double)n_tr * sizeof(Tuple) <= 64e9;
              if (!worth && !triples_force) goto triples_done;
              ML_TRY(hipMalloc((void**)&d_triples, (size_t)std::max<int64_t>(n_tr, 1) * sizeof(Tuple)));
              ML_TRY(hipMalloc((void**)&d_tr_rank, (size_t)n_tw * 8));
              hipLaunchKernelGGL(lattice_list_kernel, dim3(gtw), dim3(256), 0, 0, B, d_tw, d_tp, n_tw, d_triples, d_tr_rank);
              ML_TRY(hipGetLastError());
              const int64_t per_batch_t = std::max<int64_t>(1, ((int64_t)1 << 30) / nd);
              for (int64_t first = 0; first < n_tr; first += per_batch_t) {
                const int64_t nt = std::min<int64_t>(per_batch_t, n_tr - first);
                hipLaunchKernelGGL(xr_triple_succ_kernel, dim3((unsigned)((nt * nd + 255) / 256)), dim3(256), 0, 0, P, L, w0,
                                   d_triples + first, nt, dem_t, d_lat_words, d_oob);
                ML_TRY(hipGetLastError());
              }
              ML_TRY(hipDeviceSynchronize());
              marked = true;
            }
          triples_done:
            (void)hipFree(d_tw);
            (void)hipFree(d_tc);
            (void)hipFree(d_tp);
            (void)hipFree(d_tr_rank);
            (void)hipFree(d_triples);
            d_tw = d_tc = d_tp = nullptr;
            d_tr_rank = nullptr;
            d_triples = nullptr;
          }
        }
        // the factored form of the marking pass (backward_fact_kernel, LK = 3) where the pair list factors and the tables fit
        if (!marked && !(std::getenv("SDPGPU_MULTI_FACT") && std::atoi(std::getenv("SDPGPU_MULTI_FACT")) == 0)) {
          std::vector<double> u1, u2;
          std::vector<int> idx((size_t)nd);
          bool ok = true;
          for (int j = 0; j < nd && ok; ++j) {
            const double2 d = h_dem[(size_t)sp.off[(size_t)t] + j];
            size_t k1 = 0, k2 = 0;
            while (k1 < u1.size() && u1[k1] != d.x) ++k1;
            if (k1 == u1.size()) u1.push_back(d.x);
            while (k2 < u2.size() && u2[k2] != d.y) ++k2;
            if (k2 == u2.size()) u2.push_back(d.y);
            ok = u1.size() <= 4096 && u2.size() <= 4096;
            idx[(size_t)j] = (int)k1 | ((int)k2 << 16);
          }
          const size_t smem_m = fact_lds_bytes(NA, nd, P.qb, u1.size() + u2.size(), false, true);
          if (ok && smem_m <= kLdsPerCUSparse) {
            if (d_fact) (void)hipFree(d_fact);
            d_fact = nullptr;
            ML_TRY(hipMalloc((void**)&d_fact, (u1.size() + u2.size()) * 8 + (size_t)nd * 4));
            ML_TRY(hipMemcpy(d_fact, u1.data(), u1.size() * 8, hipMemcpyHostToDevice));
            ML_TRY(hipMemcpy(d_fact + u1.size() * 8, u2.data(), u2.size() * 8, hipMemcpyHostToDevice));
            ML_TRY(hipMemcpy(d_fact + (u1.size() + u2.size()) * 8, idx.data(), (size_t)nd * 4, hipMemcpyHostToDevice));
            FactList F;
            F.u1 = reinterpret_cast<const double*>(d_fact);
            F.u2 = F.u1 + u1.size();
            F.idx = reinterpret_cast<const int*>(F.u2 + u2.size());
            F.nu1 = (int)u1.size();
            F.nu2 = (int)u2.size();
            F.run = fact_run(idx);
#define ML_MARK(MD, W32)                                                                                                       \
  do {                                                                                                                        \
    if (smem_m > 64 * 1024)                                                                                                   \
      ML_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(backward_fact_kernel<MD, false, 3, W32>),                          \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_m));                                  \
    for (int64_t first = 0; first < n_states[t]; first += (int64_t)1 << 21) {                                                 \
      const int64_t nb = std::min<int64_t>((int64_t)1 << 21, n_states[t] - first);                                            \
      hipLaunchKernelGGL((backward_fact_kernel<MD, false, 3, W32>), dim3((unsigned)nb), dim3(kFactThreads), smem_m, 0, P, d_states[t], \
                         first, n_states[t], F, d_prob + sp.off[(size_t)t], (const double*)nullptr, (const int*)nullptr,      \
                         (double*)nullptr, (int*)nullptr, (unsigned long long*)nullptr, L, (const uint2*)nullptr,             \
                         (const double*)nullptr, d_lat_words, d_oob);                                                         \
      ML_TRY(hipGetLastError());                                                                                              \
    }                                                                                                                         \
  } while (0)
            if (P.model == 1) {
              if (lat_i32) ML_MARK(1, true); else ML_MARK(1, false);
            } else {
              if (lat_i32) ML_MARK(2, true); else ML_MARK(2, false);
            }
#undef ML_MARK
            marked = true;
          }
        }
        // a dispatch carries at most 2^32 work-items: batches of states
        const int64_t per_batch = std::max<int64_t>(1, ((int64_t)1 << 30) / NA);
        for (int64_t first = 0; !marked && first < n_states[t]; first += per_batch) {
          const int64_t ns = std::min<int64_t>(per_batch, n_states[t] - first);
          hipLaunchKernelGGL(lattice_mark_kernel, dim3((unsigned)((ns * NA + 255) / 256)), dim3(256), 0, 0, P, L,
                             d_states[t] + first, ns, dem_t, d_lat_words, d_oob);
          ML_TRY(hipGetLastError());
        }
        const unsigned gw = (unsigned)((n_words + 255) / 256);
        hipLaunchKernelGGL(lattice_popc_kernel, dim3(gw), dim3(256), 0, 0, d_lat_words, n_words, d_lat_counts);
        ML_TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        ML_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_lat_counts, d_lat_prefix, (int)n_words));
        ML_TRY(hipMalloc(&d_tmp, tmp_bytes));
        ML_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_lat_counts, d_lat_prefix, (int)n_words));
        ML_TRY(hipFree(d_tmp));
        d_tmp = nullptr;
        unsigned int last_prefix = 0, last_count = 0;
        int oob = 0;
        ML_TRY(hipMemcpy(&last_prefix, d_lat_prefix + (n_words - 1), 4, hipMemcpyDeviceToHost));
        ML_TRY(hipMemcpy(&last_count, d_lat_counts + (n_words - 1), 4, hipMemcpyDeviceToHost));
        ML_TRY(hipMemcpy(&oob, d_oob, 4, hipMemcpyDeviceToHost));
        (void)hipFree(d_lat_counts);
        d_lat_counts = nullptr;
        if (oob) {
          g_ml_error = "two-product solver: a successor left the state box computed from the parameters (negative demands?)";
          rc = SDPGPU_ERR_UNSUPPORTED;
          goto fail;
        }
        const int64_t n_next = (int64_t)last_prefix + last_count;
        if (n_next >= 2000000000LL) {
          g_ml_error = "two-product solver: more than 2e9 reachable states in one period";
          rc = SDPGPU_ERR_UNSUPPORTED;
          goto fail;
        }
        n_states[t + 1] = n_next;
        ML_TRY(hipMalloc((void**)&d_states[t + 1], (size_t)std::max<int64_t>(n_next, 1) * sizeof(Tuple)));
        hipLaunchKernelGGL(lattice_list_kernel, dim3(gw), dim3(256), 0, 0, L, d_lat_words, d_lat_prefix, n_words,
                           d_states[t + 1], d_lat_rank[t]);
        ML_TRY(hipGetLastError());
        ML_TRY(hipDeviceSynchronize());
        (void)hipFree(d_lat_words);
        (void)hipFree(d_lat_prefix);
        d_lat_words = d_lat_prefix = nullptr;
        continue;
      }
      if (nc >= 2000000000LL) {
        g_ml_error = "multilead: the reachable set outgrows 32-bit candidate indices (the reference's own comment calls such instances unsolvable)";
        rc = SDPGPU_ERR_UNSUPPORTED;
        goto fail;
      }
      P.overhead = sp.overhead[t];
      P.is_last = 0;
      ML_TRY(hipMalloc((void**)&d_hash, (size_t)nc * 8));
      ML_TRY(hipMalloc((void**)&d_hash2, (size_t)nc * 8));
      ML_TRY(hipMalloc((void**)&d_order, (size_t)nc * 4));
      ML_TRY(hipMalloc((void**)&d_order2, (size_t)nc * 4));
      ML_TRY(hipMalloc((void**)&d_head, (size_t)nc * 4));
      ML_TRY(hipMalloc((void**)&d_rank, (size_t)nc * 4));
      ML_TRY(hipMalloc((void**)&d_uid[t], (size_t)nc * 4));
      const unsigned gsa = (unsigned)((n_states[t] * NA + 255) / 256);
      hipLaunchKernelGGL(expand_kernel, dim3(gsa), dim3(256), 0, 0, P, d_states[t], n_states[t], dem_t, d_hash, d_order);
      ML_TRY(hipGetLastError());
      size_t tmp_bytes = 0;
      ML_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_hash, d_hash2, d_order, d_order2, (int)nc));
      ML_TRY(hipMalloc(&d_tmp, tmp_bytes));
      ML_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_hash, d_hash2, d_order, d_order2, (int)nc));
      ML_TRY(hipFree(d_tmp));
      d_tmp = nullptr;
      const unsigned gc = (unsigned)((nc + 255) / 256);
      hipLaunchKernelGGL(mark_heads_kernel, dim3(gc), dim3(256), 0, 0, P, d_states[t], dem_t, d_order2, nc, d_head);
      ML_TRY(hipGetLastError());
      ML_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, d_head, d_rank, (int)nc));
      ML_TRY(hipMalloc(&d_tmp, tmp_bytes));
      ML_TRY(hipcub::DeviceScan::InclusiveSum(d_tmp, tmp_bytes, d_head, d_rank, (int)nc));
      ML_TRY(hipFree(d_tmp));
      d_tmp = nullptr;
      int n_next = 0;
      ML_TRY(hipMemcpy(&n_next, d_rank + (nc - 1), 4, hipMemcpyDeviceToHost));
      n_states[t + 1] = n_next;
      ML_TRY(hipMalloc((void**)&d_states[t + 1], (size_t)n_next * sizeof(Tuple)));
      hipLaunchKernelGGL(scatter_kernel, dim3(gc), dim3(256), 0, 0, P, d_states[t], dem_t, d_order2, d_head, d_rank, nc,
                         d_states[t + 1], d_uid[t]);
      ML_TRY(hipGetLastError());
      ML_TRY(hipDeviceSynchronize());
      (void)hipFree(d_hash); (void)hipFree(d_hash2); (void)hipFree(d_order); (void)hipFree(d_order2);
      (void)hipFree(d_head); (void)hipFree(d_rank);
      d_hash = d_hash2 = nullptr;
      d_order = d_order2 = nullptr;
      d_head = d_rank = nullptr;
    }
    // ---- read-out request (sdpgpu_multi_set_table): room for every state of every period? ----
    if (g_ml_table) {
      int64_t rows = 0;
      for (int t = 0; t < T; ++t) {
        table_off[(size_t)t] = rows;
        rows += n_states[t];
      }
      g_ml_table->rows = rows;
      table_rows_ok = rows <= g_ml_table->capacity && g_ml_table->period && g_ml_table->i1 && g_ml_table->i2 &&
                      g_ml_table->q1 && g_ml_table->q2 && g_ml_table->cash && g_ml_table->value && g_ml_table->a1 &&
                      g_ml_table->a2;
    }
    // ---- backward ----
    for (int t = T - 1; t >= 0; --t) {
      const int nd = sp.off[(size_t)t + 1] - sp.off[(size_t)t];
      P.nd = nd;
      P.overhead = sp.overhead[t];
      P.is_last = (t == T - 1);
      ML_TRY(hipMalloc((void**)&d_vcur, (size_t)n_states[t] * 8));
      if (d_act) (void)hipFree(d_act);
      d_act = nullptr;
      ML_TRY(hipMalloc((void**)&d_act, (size_t)n_states[t] * 4));
      bool fact_done = false;
      const size_t smem = (size_t)NA * 8 + (size_t)nd * (sizeof(DemandTerms) + 8) + (size_t)(P.qb + 1) * 4;
      // (Q(s, a) of every action pair of a state sits in LDS: Qbound up to ~140 within the 160 KiB of a compute unit)
      constexpr size_t kLdsPerCU = 160 * 1024;  // (gfx950; sdpgpu_internal.hpp has the same figure for the grid kernels)
      if (smem > kLdsPerCU) {
        char buf[200];
        std::snprintf(buf, sizeof buf, "two-product solver: Qbound %d with %d demand pairs needs %zu B of LDS per state, over the %zu B of a compute unit",
                      (int)P.qb, nd, smem, kLdsPerCU);
        g_ml_error = buf;
        rc = SDPGPU_ERR_UNSUPPORTED;
        goto fail;
      }
      if (smem > 64 * 1024)  // above the legacy limit of a launch: raise the kernel's dynamic-LDS limit
        ML_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      // models 1 / 2: the factored kernel when the period's pair list has few distinct first / second demands and the tables
      // fit the LDS (backward_fact_kernel); SDPGPU_MULTI_FACT=0: the per-cell lambdas of round 3
      if (P.model != 0 && !(std::getenv("SDPGPU_MULTI_FACT") && std::atoi(std::getenv("SDPGPU_MULTI_FACT")) == 0)) {
        std::vector<double> u1, u2;
        std::vector<int> idx((size_t)nd);
        bool ok = true;
        for (int j = 0; j < nd && ok; ++j) {
          const double2 d = h_dem[(size_t)sp.off[(size_t)t] + j];
          size_t k1 = 0, k2 = 0;
          while (k1 < u1.size() && u1[k1] != d.x) ++k1;
          if (k1 == u1.size()) u1.push_back(d.x);
          while (k2 < u2.size() && u2[k2] != d.y) ++k2;
          if (k2 == u2.size()) u2.push_back(d.y);
          ok = u1.size() <= 4096 && u2.size() <= 4096;
          idx[(size_t)j] = (int)k1 | ((int)k2 << 16);
        }
        const size_t smem_f = fact_lds_bytes(NA, nd, P.qb, u1.size() + u2.size(), P.is_last != 0);
        if (ok && smem_f <= kLdsPerCU) {
          if (d_fact) (void)hipFree(d_fact);
          d_fact = nullptr;
          const size_t bytes = (u1.size() + u2.size()) * 8 + (size_t)nd * 4;
          ML_TRY(hipMalloc((void**)&d_fact, bytes));
          ML_TRY(hipMemcpy(d_fact, u1.data(), u1.size() * 8, hipMemcpyHostToDevice));
          ML_TRY(hipMemcpy(d_fact + u1.size() * 8, u2.data(), u2.size() * 8, hipMemcpyHostToDevice));
          ML_TRY(hipMemcpy(d_fact + (u1.size() + u2.size()) * 8, idx.data(), (size_t)nd * 4, hipMemcpyHostToDevice));
          if (d_vdense) (void)hipFree(d_vdense);
          d_vdense = nullptr;
          if (!P.is_last && d_lat_rank[t]) {
            double cap_gb = 16.0;
            if (const char* e = std::getenv("SDPGPU_MULTI_DENSE_GB")) cap_gb = std::atof(e);
            if ((double)sp.lat.bits * 8.0 <= cap_gb * 1e9 && hipMalloc((void**)&d_vdense, (size_t)sp.lat.bits * 8) == hipSuccess) {
              hipLaunchKernelGGL(dense_scatter_kernel, dim3((unsigned)((n_states[t + 1] + 255) / 256)), dim3(256), 0, 0, sp.lat,
                                 d_states[t + 1], n_states[t + 1], d_vnext, d_vdense);
              ML_TRY(hipGetLastError());
            } else {
              (void)hipGetLastError();  // (no room: ranks and values, as before)
              d_vdense = nullptr;
            }
          }
          FactList F;
          F.u1 = reinterpret_cast<const double*>(d_fact);
          F.u2 = F.u1 + u1.size();
          F.idx = reinterpret_cast<const int*>(F.u2 + u2.size());
          F.nu1 = (int)u1.size();
          F.nu2 = (int)u2.size();
          F.run = fact_run(idx);
#define ML_FACT(MD, LS, LKK, W32)                                                                                               \
  do {                                                                                                                      \
    if (smem_f > 64 * 1024)                                                                                                 \
      ML_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(backward_fact_kernel<MD, LS, LKK, W32>),                              \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_f));                                \
    for (int64_t first = 0; first < n_states[t]; first += (int64_t)1 << 21) {                                               \
      const int64_t nb = std::min<int64_t>((int64_t)1 << 21, n_states[t] - first);                                          \
      hipLaunchKernelGGL((backward_fact_kernel<MD, LS, LKK, W32>), dim3((unsigned)nb), dim3(kFactThreads), smem_f, 0, P, d_states[t], first, \
                         n_states[t], F, d_prob + sp.off[(size_t)t], d_vnext, d_uid[t], d_vcur, d_act, d_cells, sp.lat,     \
                         d_lat_rank[t], d_vdense, (unsigned int*)nullptr, (int*)nullptr);                                   \
      ML_TRY(hipGetLastError());                                                                                            \
    }                                                                                                                       \
  } while (0)
          const int lk = d_vdense ? 2 : (d_lat_rank[t] ? 1 : 0);
          const bool w32 = lat_i32 && lk != 0;
          if (P.model == 1) {
            if (P.is_last) ML_FACT(1, true, 0, false);
            else if (lk == 2 && w32) ML_FACT(1, false, 2, true);
            else if (lk == 2) ML_FACT(1, false, 2, false);
            else if (lk == 1 && w32) ML_FACT(1, false, 1, true);
            else if (lk == 1) ML_FACT(1, false, 1, false);
            else ML_FACT(1, false, 0, false);
          } else {
            if (P.is_last) ML_FACT(2, true, 0, false);
            else if (lk == 2 && w32) ML_FACT(2, false, 2, true);
            else if (lk == 2) ML_FACT(2, false, 2, false);
            else if (lk == 1 && w32) ML_FACT(2, false, 1, true);
            else if (lk == 1) ML_FACT(2, false, 1, false);
            else ML_FACT(2, false, 0, false);
          }
#undef ML_FACT
          fact_done = true;
        }
      }
      // the lead-time family: a wave per state (backward_lead_wave_kernel); SDPGPU_MULTI_WAVE=0: round 3's workgroup per state
      const bool lead_wave = !fact_done && P.model == 0 && NA <= 4096 && !(std::getenv("SDPGPU_MULTI_WAVE") && std::atoi(std::getenv("SDPGPU_MULTI_WAVE")) == 0);
      if (lead_wave) {
        if (!d_oc) {
          ML_TRY(hipMalloc((void**)&d_oc, (size_t)NA * 8));
          hipLaunchKernelGGL(order_cost_kernel, dim3((unsigned)((NA + 255) / 256)), dim3(256), 0, 0, P, d_oc);
          ML_TRY(hipGetLastError());
        }
        const size_t smem_w = (size_t)nd * (4 * sizeof(DemandTerms) + 8);
        for (int64_t first = 0; first < n_states[t]; first += (int64_t)1 << 24) {  // 4M workgroups of four states
          const int64_t ns = std::min<int64_t>((int64_t)1 << 24, n_states[t] - first);
          if (P.is_last)
            hipLaunchKernelGGL(backward_lead_wave_kernel<true>, dim3((unsigned)((ns + 3) / 4)), dim3(256), smem_w, 0, P, d_states[t], first,
                               n_states[t], d_dem + sp.off[(size_t)t], d_prob + sp.off[(size_t)t], d_vnext, d_uid[t], d_vcur, d_act, d_oc);
          else
            hipLaunchKernelGGL(backward_lead_wave_kernel<false>, dim3((unsigned)((ns + 3) / 4)), dim3(256), smem_w, 0, P, d_states[t], first,
                               n_states[t], d_dem + sp.off[(size_t)t], d_prob + sp.off[(size_t)t], d_vnext, d_uid[t], d_vcur, d_act, d_oc);
          ML_TRY(hipGetLastError());
        }
      } else if (!fact_done)
      // a dispatch carries at most 2^32 work-items: batches of 4M workgroups (2^30 lanes)
      for (int64_t first = 0; first < n_states[t]; first += (int64_t)1 << 22) {
        const int64_t nb = std::min<int64_t>((int64_t)1 << 22, n_states[t] - first);
        hipLaunchKernelGGL(backward_kernel, dim3((unsigned)nb), dim3(256), smem, 0, P, d_states[t], first, n_states[t],
                           d_dem + sp.off[(size_t)t], d_prob + sp.off[(size_t)t], d_vnext, d_uid[t], d_vcur, d_act, d_cells,
                           sp.lat, d_lat_rank[t]);
        ML_TRY(hipGetLastError());
      }
      if (P.model != 1) total_cells += n_states[t] * (int64_t)NA * nd;
      if (table_rows_ok) {  // the memo of this period: states, values, actions (hash order; sorted on the host below)
        const size_t n = (size_t)n_states[t];
        const size_t at = (size_t)table_off[(size_t)t];
        checked_resize(h_tuples, n);
        checked_resize(h_acts, n);
        ML_TRY(hipMemcpy(h_tuples.data(), d_states[t], n * sizeof(Tuple), hipMemcpyDeviceToHost));
        ML_TRY(hipMemcpy(g_ml_table->value + at, d_vcur, n * 8, hipMemcpyDeviceToHost));
        ML_TRY(hipMemcpy(h_acts.data(), d_act, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
          const Tuple& u = h_tuples[i];
          g_ml_table->period[at + i] = t + 1;
          g_ml_table->i1[at + i] = u.i1;
          g_ml_table->i2[at + i] = u.i2;
          g_ml_table->q1[at + i] = u.q1;
          g_ml_table->q2[at + i] = u.q2;
          g_ml_table->cash[at + i] = u.cash;
          const int a1 = h_acts[i] / P.qb, a2 = h_acts[i] % P.qb;
          g_ml_table->a1[at + i] = P.model == 2 ? (int)u.i1 + a1 : a1;  // model 2 reports order-up-to levels
          g_ml_table->a2[at + i] = P.model == 2 ? (int)u.i2 + a2 : a2;
        }
      }
      if (d_vnext) (void)hipFree(d_vnext);
      d_vnext = d_vcur;
      d_vcur = nullptr;
    }
    ML_TRY(hipEventRecord(ev1, 0));
    ML_TRY(hipEventSynchronize(ev1));
    double v = 0;
    int act = 0;
    ML_TRY(hipMemcpy(&v, d_vnext, 8, hipMemcpyDeviceToHost));
    ML_TRY(hipMemcpy(&act, d_act, 4, hipMemcpyDeviceToHost));
    if (P.model == 1) {  // only the actions a state is offered were evaluated: counted on the device
      unsigned long long c = 0;
      ML_TRY(hipMemcpy(&c, d_cells, 8, hipMemcpyDeviceToHost));
      total_cells = (int64_t)c;
    }
    if (final_value) *final_value = ini.cash + v;  // MultiProductLeadtime.java:234 / MultiItemCash.java:132
    if (q1) *q1 = act / P.qb;
    if (q2) *q2 = act % P.qb;
    if (states_per_period)
      for (int t = 0; t < T; ++t) states_per_period[t] = n_states[t];
    if (cells) *cells = total_cells;
    if (gpu_ms) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, ev0, ev1);
      *gpu_ms = ms;
    }
    rc = SDPGPU_OK;
  } catch (const std::bad_alloc&) {  // (the device buffers are released below on this path too)
    g_ml_error = "two-product solver: host allocation failed (std::bad_alloc)";
    rc = SDPGPU_ERR_ALLOC;
  } catch (const std::exception& e) {
    g_ml_error = std::string("two-product solver: internal error: ") + e.what();
    rc = SDPGPU_ERR_INTERNAL;
  } catch (...) {
    g_ml_error = "two-product solver: internal error (unknown exception)";
    rc = SDPGPU_ERR_INTERNAL;
  }
fail:
  for (Tuple* p : d_states) if (p) (void)hipFree(p);
  for (int* p : d_uid) if (p) (void)hipFree(p);
  for (uint2* p : d_lat_rank) if (p) (void)hipFree(p);
  if (d_lat_words) (void)hipFree(d_lat_words);
  if (d_lat_prefix) (void)hipFree(d_lat_prefix);
  if (d_lat_counts) (void)hipFree(d_lat_counts);
  if (d_tw) (void)hipFree(d_tw);
  if (d_tc) (void)hipFree(d_tc);
  if (d_tp) (void)hipFree(d_tp);
  if (d_tr_rank) (void)hipFree(d_tr_rank);
  if (d_triples) (void)hipFree(d_triples);
  if (d_oob) (void)hipFree(d_oob);
  if (d_dem) (void)hipFree(d_dem);
  if (d_prob) (void)hipFree(d_prob);
  if (d_hash) (void)hipFree(d_hash);
  if (d_hash2) (void)hipFree(d_hash2);
  if (d_order) (void)hipFree(d_order);
  if (d_order2) (void)hipFree(d_order2);
  if (d_head) (void)hipFree(d_head);
  if (d_rank) (void)hipFree(d_rank);
  if (d_tmp) (void)hipFree(d_tmp);
  if (d_vcur) (void)hipFree(d_vcur);
  if (d_vnext) (void)hipFree(d_vnext);
  if (d_act) (void)hipFree(d_act);
  if (d_cells) (void)hipFree(d_cells);
  if (d_fact) (void)hipFree(d_fact);
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  return rc;
}

}  // namespace

extern "C" {

const char* sdpgpu_multilead_last_error(void) { return g_ml_error.c_str(); }

void sdpgpu_multi_set_table(sdpgpu_multi_table* table) { g_ml_table = table; }

static int multilead_body(const sdpgpu_multilead* k, double* final_value, int32_t* q1, int32_t* q2,
                          int64_t* states_per_period, int64_t* cells, double* gpu_ms) {
  if (!k || k->T < 1 || k->T > 16 || k->n1 < 1 || k->n1 > 16 || k->n2 < 1 || k->n2 > 16 || k->q_bound < 1 ||
      k->q_bound > 256) {
    g_ml_error = "multilead: bad descriptor";
    return SDPGPU_ERR_ARG;
  }
  SparseProblem sp;
  sp.T = k->T;
  const int nd = k->n1 * k->n2;
  for (int t = 0; t <= k->T; ++t) sp.off.push_back(t * nd);
  for (int t = 0; t < k->T; ++t)  // the same product list every period (GetPmfMulti.java:157-172)
    for (int i = 0; i < k->n1; ++i)
      for (int j = 0; j < k->n2; ++j) {
        // `new Demands((int) dAndP[j][0], (int) dAndP[j][1])`, CashRecursionMultiLead.java:74
        sp.dem.push_back(make_double2((double)(int)k->v1[i], (double)(int)k->v2[j]));
        sp.prob.push_back(k->p1[i] * k->p2[j]);
      }
  MLParams& P = sp.P;
  for (int i = 0; i < 2; ++i) {
    P.price[i] = k->price[i];
    P.vari[i] = k->vari_cost[i];
    P.sal[i] = k->sal_value[i];
  }
  P.r0 = k->r0; P.r1 = k->r1; P.r2 = k->r2; P.limit = k->limit; P.interest_free = k->interest_free;
  P.min_inventory = k->min_inventory; P.max_inventory = k->max_inventory;
  P.min_cash = k->min_cash; P.max_cash = k->max_cash; P.discount = k->discount;
  P.qb = k->q_bound;
  P.cash_int_cast = k->cash_int_cast;
  P.model = 0;
  for (int t = 0; t < k->T; ++t) sp.overhead[t] = k->overhead[t];
  sp.ini = Tuple{k->ini_i1, k->ini_i2, 0.0, 0.0, k->ini_cash};  // MultiProductLeadtime.java:232
  return sparse_solve(sp, final_value, q1, q2, states_per_period, cells, gpu_ms);
}

int sdpgpu_multilead_solve(const sdpgpu_multilead* k, double* final_value, int32_t* q1, int32_t* q2,
                           int64_t* states_per_period, int64_t* cells, double* gpu_ms) {
  g_ml_error.clear();
  return ml_guard("multilead", [&] { return multilead_body(k, final_value, q1, q2, states_per_period, cells, gpu_ms); });
}

static int multicash_body(const sdpgpu_multicash* k, int model, double deposit_rate, double* final_value, int32_t* q1,
                          int32_t* q2, int64_t* states_per_period, int64_t* cells, double* gpu_ms) {
  if (!k || k->T < 1 || k->T > 16 || k->q_bound < 1 || k->q_bound > 256 || !k->pmf_off || !k->d1 || !k->d2 || !k->p) {
    g_ml_error = "multicash: bad descriptor";
    return SDPGPU_ERR_ARG;
  }
  if (model == 1 && (k->vari_cost[0] < 0 || k->vari_cost[1] < 0)) {
    g_ml_error = "multicash: negative unit costs are not supported (the offered actions of a row must form a prefix)";
    return SDPGPU_ERR_UNSUPPORTED;
  }
  if (model == 1 && (k->min_cash < 0 || k->ini_cash < 0)) {
    // (0, 0) must always be on offer (0 < cash + 0.1): the candidate lists lean on it
    g_ml_error = "multicash: negative cash is not supported (MultiItemCash.java:51 has minCashState = 0)";
    return SDPGPU_ERR_UNSUPPORTED;
  }
  SparseProblem sp;
  sp.T = k->T;
  for (int t = 0; t <= k->T; ++t) {
    if (k->pmf_off[t] < 0 || (t && k->pmf_off[t] <= k->pmf_off[t - 1]) || k->pmf_off[t] - (t ? k->pmf_off[t - 1] : 0) > 4096) {
      g_ml_error = "multicash: pmf_off must ascend, with 1..4096 demand pairs per period";
      return SDPGPU_ERR_ARG;
    }
    sp.off.push_back(k->pmf_off[t] - k->pmf_off[0]);
  }
  for (int32_t j = k->pmf_off[0]; j < k->pmf_off[k->T]; ++j) {
    // `new Demands((int) dAndP[j][0], (int) dAndP[j][1])`, CashRecursionMulti.java:96; CashRecursionMultiXR keeps
    // the doubles (`new double[] {dAndP[j][0], dAndP[j][1]}`, CashRecursionMultiXR.java:79)
    if (model == 1)
      sp.dem.push_back(make_double2((double)(int)k->d1[j], (double)(int)k->d2[j]));
    else
      sp.dem.push_back(make_double2(k->d1[j], k->d2[j]));
    sp.prob.push_back(k->p[j]);
  }
  MLParams& P = sp.P;
  for (int i = 0; i < 2; ++i) {
    P.price[i] = k->price[i];
    P.vari[i] = k->vari_cost[i];
    P.sal[i] = k->sal_price[i];
  }
  P.min_inventory = k->min_inventory; P.max_inventory = k->max_inventory;
  P.min_cash = k->min_cash; P.max_cash = k->max_cash; P.discount = k->discount;
  P.qb = k->q_bound;
  P.cash_int_cast = 1;
  P.model = model;
  P.one_minus_deposit = 1 - deposit_rate;
  {
    // The box the (int)-cast successors live in: i1 is clamped above by maxInventoryState, i2 can only grow by
    // Qbound - 1 a period (it has no upper clamp, MultiItemCash.java:113), cash is clamped; the XR family's R adds
    // variCost . x to the cash and is an integer when the unit costs are.
    bool ok = true;
    for (int32_t j = k->pmf_off[0]; j < k->pmf_off[k->T]; ++j) ok = ok && k->d1[j] >= 0 && k->d2[j] >= 0;
    const double b1 = std::max(std::floor(std::fabs(k->ini_i1)), std::floor(std::fabs(k->max_inventory)));
    const double b2 = std::max(std::floor(std::fabs(k->ini_i2)) + (double)(k->T - 1) * (k->q_bound - 1) + (double)k->q_bound,
                               std::floor(std::fabs(k->min_inventory)));
    double c_lo = std::floor(k->min_cash) - 1, c_hi = std::ceil(k->max_cash) + 1;
    const bool int_costs = k->vari_cost[0] >= 0 && k->vari_cost[1] >= 0 && k->vari_cost[0] == std::floor(k->vari_cost[0]) &&
                           k->vari_cost[1] == std::floor(k->vari_cost[1]) && k->vari_cost[0] < 1e6 && k->vari_cost[1] < 1e6;
    if (model == 2) ok = ok && int_costs;  // (its R is an integer only then)
    if (int_costs) c_hi += k->vari_cost[0] * b1 + k->vari_cost[1] * b2;  // the range of R = cash + variCost . x
    ok = ok && k->ini_i1 >= 0 && k->ini_i2 >= 0 && b1 < 1e6 && b2 < 1e6 && c_hi - c_lo < 1e9;
    if (ok) {
      Lattice L;
      L.n2 = (long long)b2 + 1;
      L.nr = (long long)(c_hi - c_lo) + 1;
      L.r0 = (long long)c_lo;
      L.skew1 = (model == 1 && int_costs) ? (long long)k->vari_cost[0] : 0;
      L.skew2 = (model == 1 && int_costs) ? (long long)k->vari_cost[1] : 0;
      const double bits = ((double)b1 + 1) * (double)L.n2 * (double)L.nr;
      L.bits = bits < 6.0e10 ? (long long)bits : 0;  // <= 7.5 GB of bitmap; int32 word indices
      sp.lat = L;
      sp.lattice_ok = L.bits > 0;
    }
  }
  // MultiItemCash.java:130; MultiItemCashXR.java:158 hands iniCash over as R
  sp.ini = Tuple{k->ini_i1, k->ini_i2, 0.0, 0.0, k->ini_cash};
  int a1 = 0, a2 = 0;
  const int rc = sparse_solve(sp, final_value, &a1, &a2, states_per_period, cells, gpu_ms);
  if (rc == SDPGPU_OK) {  // model 2 answers with the order-up-to levels themselves (getAction(iniState)[0], [1])
    if (q1) *q1 = model == 2 ? (int)k->ini_i1 + a1 : a1;
    if (q2) *q2 = model == 2 ? (int)k->ini_i2 + a2 : a2;
  }
  return rc;
}

static int multicash_common(const sdpgpu_multicash* k, int model, double deposit_rate, double* final_value, int32_t* q1,
                            int32_t* q2, int64_t* states_per_period, int64_t* cells, double* gpu_ms) {
  g_ml_error.clear();
  return ml_guard(model == 2 ? "multixr" : "multicash",
                  [&] { return multicash_body(k, model, deposit_rate, final_value, q1, q2, states_per_period, cells, gpu_ms); });
}

int sdpgpu_multicash_solve(const sdpgpu_multicash* k, double* final_value, int32_t* q1, int32_t* q2,
                           int64_t* states_per_period, int64_t* cells, double* gpu_ms) {
  return multicash_common(k, 1, 0.0, final_value, q1, q2, states_per_period, cells, gpu_ms);
}

int sdpgpu_multixr_solve(const sdpgpu_multicash* k, double deposit_rate, double* final_value, int32_t* y1, int32_t* y2,
                         int64_t* states_per_period, int64_t* cells, double* gpu_ms) {
  return multicash_common(k, 2, deposit_rate, final_value, y1, y2, states_per_period, cells, gpu_ms);
}

}  // extern "C"

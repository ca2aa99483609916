/*
 * sdpref.c -- CPU ORACLE: line-by-line C restatement of the reference's SDP recursion.
 * TEST INFRASTRUCTURE ONLY (see sdpref.h for the rule and for the parity-pin status).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (strict IEEE fp64, no FMA), see Makefile.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * checkout).  The restatement keeps the reference's operation ORDER: that is what makes the
 * values bit-reproducible and the arg-opt indices exact.
 */
#include "sdpref.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Java arithmetic that differs from the C library's.
 * ---------------------------------------------------------------------------------------- */

/* java.lang.Math.round(double): nearest long, ties toward +infinity (NOT llround, which rounds
 * halves away from zero; NOT floor(x+0.5), which is wrong for 0.49999999999999994).
 * Used at CashConstraint.java:131, CashOverdraft.java:116, SingleProductLeadtime.java:117. */
int64_t sdpref_java_round(double x) {
  if (x != x) return 0;
  if (x >= 9223372036854775807.0) return INT64_MAX;
  if (x <= -9223372036854775808.0) return INT64_MIN;
  double f = floor(x);
  double diff = x - f; /* exact in fp64 */
  return (int64_t)(diff >= 0.5 ? f + 1.0 : f);
}

/* java.lang.Math.max(double,double): NaN-propagating, max(-0.0, 0.0) = 0.0. */
double sdpref_java_max(double a, double b) {
  if (a != a) return a;
  if (a == 0.0 && b == 0.0 && signbit(a)) return b;
  return (a >= b) ? a : b;
}

/* java.lang.Math.min(double,double). */
double sdpref_java_min(double a, double b) {
  if (a != a) return a;
  if (a == 0.0 && b == 0.0 && signbit(b)) return b;
  return (a <= b) ? a : b;
}

/* Java (int) cast of a double: NaN -> 0, saturating, truncation toward zero. */
int32_t sdpref_java_d2i(double x) {
  if (x != x) return 0;
  if (x >= 2147483647.0) return INT32_MAX;
  if (x <= -2147483648.0) return INT32_MIN;
  return (int32_t)x;
}

#define jround sdpref_java_round
#define jmax sdpref_java_max
#define jmin sdpref_java_min
#define jd2i sdpref_java_d2i

/* ------------------------------------------------------------------------------------------
 * State tuple and problem context
 * ---------------------------------------------------------------------------------------- */

/* State.java:12-19, LeadtimeState.java:10-20, CashState.java:12-18, CashLeadtimeState.java:11-17. */
typedef struct st {
  int32_t period;
  double x, cash, preq;
  double preq2; /* lead_time 2 only (a generalisation the reference does not have): the order arriving
                   next period; preq is the one arriving this period */
} st_t;

typedef struct ctx {
  const sdpgpu_desc* d;
  const int32_t* off;
  const double* pd;
  const double* pp;
  const double* oh; /* may be NULL */
  int32_t T;
} ctx_t;

/* User-defined lambdas (the counterpart of sdpgpu_create_custom): the SAME three functions the product
 * compiles with hipRTC, compiled for the host by the test harness (oracle/sdpref.py: compile_custom) and
 * registered here as function pointers.  While registered they replace n_actions / imm_value / transition
 * for every family; desc->family then only selects the state shape and the loop. */
typedef struct sdpref_user_ctx {
  int period;
  int T;
  double step;
  const double* params;
} sdpref_user_ctx;
typedef int (*sdpref_user_count_fn)(const sdpref_user_ctx*, double, double, double);
typedef double (*sdpref_user_imm_fn)(const sdpref_user_ctx*, double, double, double, double, double);
typedef void (*sdpref_user_trans_fn)(const sdpref_user_ctx*, double, double, double, double, double, double*, double*,
                                     double*);
static struct {
  sdpref_user_count_fn count;
  sdpref_user_imm_fn imm;
  sdpref_user_trans_fn trans;
  const double* params;
} g_user = {0, 0, 0, 0};

void sdpref_register_custom(void* count, void* imm, void* trans, const double* params) {
  g_user.count = (sdpref_user_count_fn)count;
  g_user.imm = (sdpref_user_imm_fn)imm;
  g_user.trans = (sdpref_user_trans_fn)trans;
  g_user.params = params;
}

static double overhead_at(const ctx_t* c, int32_t period) {
  return c->oh ? c->oh[period - 1] : c->d->overhead_cost;
}

static int has_cash(int family) {
  return family == SDPGPU_FAMILY_CASH || family == SDPGPU_FAMILY_OVERDRAFT ||
         family == SDPGPU_FAMILY_CASH_LEADTIME || family == SDPGPU_FAMILY_SURVIVAL;
}
static int has_preq(int family) {
  return family == SDPGPU_FAMILY_LEADTIME || family == SDPGPU_FAMILY_CASH_LEADTIME;
}

/* Number of order quantities on the full action grid: `new double[(int)(maxOrderQuantity/stepSize)+1]`
 * (CLSPTesting.java:79, Leadtime.java:51); CLSP.java:252 `limit(maxOrderQuantity + 1)` with step 1. */
static int32_t full_action_count(const sdpgpu_desc* d) {
  return jd2i(d->max_order_quantity / d->step) + 1;
}

/* sdp.cash.CashRecursionXR over cash.singleItem.CashConstraintXR's lambdas: family CASH with cash_formula 2.  The
 * state is (period, x, R) -- st_t.cash holds R, literally what CashStateXR holds (CashStateXR.java:14-22). */
static int is_xr(const sdpgpu_desc* d) { return d->family == SDPGPU_FAMILY_CASH && d->cash_formula == 2; }

/* Caller-supplied action-list lengths (sdpgpu_set_action_counts at the product's boundary): a
 * `Function<State, double[]> getFeasibleAction` (Recursion.java:49,129) whose list is a prefix of the action grid with a
 * length of its own.  The harness registers, per period, the length of every grid state; a state that is not a grid point
 * keeps the family's rule (as in the product). */
static struct {
  const int32_t* counts;     /* concatenated per period */
  const int64_t* off;        /* T + 1 offsets, off[t+1] == off[t]: period t+1 has none */
  sdpref_grid* grids;        /* layout of the problem the table belongs to */
  int32_t T;
} g_counts = {0, 0, 0, 0};

static int64_t index_of(const sdpgpu_desc* d, const sdpref_grid* g, const st_t* s);

/* getFeasibleActions.apply(state).length */
static int32_t n_actions(const ctx_t* c, const st_t* s) {
  const sdpgpu_desc* d = c->d;
  if (g_counts.counts && s->period >= 1 && s->period <= g_counts.T &&
      g_counts.off[s->period] > g_counts.off[s->period - 1]) {
    int64_t idx = index_of(d, &g_counts.grids[s->period - 1], s);
    if (idx >= 0) return g_counts.counts[g_counts.off[s->period - 1] + idx];
  }
  if (g_user.count) {
    sdpref_user_ctx u = {s->period, c->T, d->step, g_user.params};
    return g_user.count(&u, s->x, s->cash, s->preq);
  }
  switch (d->family) {
    case SDPGPU_FAMILY_BACKORDER: /* CLSP.java:251-253, CLSPTesting.java:78-86 */
    case SDPGPU_FAMILY_LEADTIME:  /* Leadtime.java:50-58 */
      return full_action_count(d);
    case SDPGPU_FAMILY_CASH: { /* CashConstraint.java:95-100 */
      if (is_xr(d)) { /* CashConstraintXR.java:84-88 */
        double variCost = d->unit_order_cost;
        double maxY = s->cash / variCost < s->x ? s->x : s->cash / variCost; /* s.getIniR() / variCost */
        int32_t length = jd2i(maxY - s->x) + 1;
        return length;
      }
      double maxQ = (double)jd2i(
          jmin(d->max_order_quantity,
               jmax(0.0, (s->cash - overhead_at(c, s->period) - d->fixed_order_cost) / d->unit_order_cost)));
      return jd2i(maxQ) + 1; /* limit((int) maxQ + 1) */
    }
    case SDPGPU_FAMILY_OVERDRAFT: /* CashOverdraft.java:72-75 */
      return jd2i(d->max_order_quantity) + 1;
    case SDPGPU_FAMILY_SURVIVAL: { /* cashSurvival.java:98-105 (bankruptBefore is always false, RiskState.java:17) */
      double maxQ = jmin(s->cash / d->unit_order_cost, d->max_order_quantity);
      maxQ = jmax(maxQ, 0);
      return jd2i(maxQ) + 1; /* limit((int) maxQ + 1) */
    }
    case SDPGPU_FAMILY_CASH_LEADTIME: { /* SingleProductLeadtime.java:72-77 */
      double maxQ = d->max_order_quantity;
      if (d->zero_order_last_period && s->period == c->T) maxQ = 0;
      return jd2i(maxQ) + 1;
    }
  }
  return 0;
}

/* DoubleStream.iterate(0, i -> i + stepSize): element k.  step is integer-valued so the
 * iterated sum equals k*step exactly.  CashConstraintXR.java:87 iterates from the inventory level instead: the
 * action IS the order-up-to level y (x is an integer there, so x + k * step is the iterated sum too). */
static double action_value(const ctx_t* c, const st_t* s, int32_t k) {
  if (is_xr(c->d)) return s->x + (double)k * c->d->step;
  return (double)k * c->d->step;
}

/* ------------------------------------------------------------------------------------------
 * Statements the overdraft / lost-sales lambdas of the reference share, written ONCE here and used by every family that has
 * them -- F4 (CashOverdraft.java:80-118), F5 (SingleProductLeadtime.java:82-119) and the two-product KAT family
 * (MultiProductLeadtime.java:162-223), whose recorded outputs (MultiProductLeadtime.java:30-50, tests/golden/
 * kat_reference.json) therefore execute the very functions F4 and F5 are checked with.
 * ---------------------------------------------------------------------------------------- */
/* Piecewise interest, CashOverdraft.java:87-95 == SingleProductLeadtime.java:88-96 == MultiProductLeadtime.java:184-192
 * (rates r0 / r2 / r3 there are r0 / r1 / r2 in the two-product file). */
static double interest_piecewise(double r0, double r_overdraft, double r_beyond, double interest_free, double limit,
                                 double cashBalanceBefore) {
  double interest = 0;
  if (cashBalanceBefore >= 0)
    interest = -r0 * cashBalanceBefore;
  else if (cashBalanceBefore >= -interest_free)
    interest = 0;
  else if (cashBalanceBefore >= -limit)
    interest = r_overdraft * (-cashBalanceBefore - interest_free);
  else
    interest = r_beyond * (-cashBalanceBefore - limit) + r_overdraft * (limit - interest_free);
  return interest;
}
static double overdraft_interest(const sdpgpu_desc* d, double cashBalanceBefore) {
  return interest_piecewise(d->r0, d->r2, d->r3, d->interest_free_amount, d->overdraft_limit, cashBalanceBefore);
}
/* `price * Math.min(stock, demand)`: CashOverdraft.java:81, SingleProductLeadtime.java:83, MultiProductLeadtime.java:170-171 */
static double lost_sales_revenue(double price, double stock, double demand) { return price * jmin(stock, demand); }
/* `cash - orderingCosts - overheadCost`: SingleProductLeadtime.java:87, MultiProductLeadtime.java:182 (CashOverdraft.java:86
 * subtracts the fixed cost first: the caller passes cash - fixedCost) */
static double balance_before(double cash, double orderingCosts, double overhead) { return cash - orderingCosts - overhead; }
/* `cashBalanceBefore - interest + revenue`: CashOverdraft.java:96, SingleProductLeadtime.java:97, MultiProductLeadtime.java:193 */
static double balance_after(double before, double interest, double revenue) { return before - interest + revenue; }
/* `Math.max(0, level)`: the lost-sales end inventory of every transition (CashOverdraft.java:108, SingleProductLeadtime.java:108,
 * MultiProductLeadtime.java:205-208) */
static double end_inventory(double level) { return jmax(0, level); }
/* the clamp as the reference writes it: upper ternary, then lower ternary (CashOverdraft.java:110-113 etc.) */
static double clamp_upper(double v, double hi) { return v > hi ? hi : v; }
static double clamp_lower(double v, double lo) { return v < lo ? lo : v; }

/* immediateValue.apply(state, action, randomDemand) */
static double imm_value(const ctx_t* c, const st_t* s, double action, double randomDemand) {
  const sdpgpu_desc* d = c->d;
  if (g_user.imm) {
    sdpref_user_ctx u = {s->period, c->T, d->step, g_user.params};
    return g_user.imm(&u, s->x, s->cash, s->preq, action, randomDemand);
  }
  switch (d->family) {
    case SDPGPU_FAMILY_BACKORDER: { /* CLSP.java:263-272 == CLSPTesting.java:97-106 */
      double fixedCost = action > 0 ? d->fixed_order_cost : 0;
      double variableCost = d->unit_order_cost * action;
      double inventoryLevel = s->x + action - randomDemand;
      double holdingCosts = d->holding_cost * jmax(inventoryLevel, 0);
      double penaltyCosts = d->penalty_cost * jmax(-inventoryLevel, 0);
      double totalCosts = fixedCost + variableCost + holdingCosts + penaltyCosts;
      return totalCosts;
    }
    case SDPGPU_FAMILY_LEADTIME: { /* Leadtime.java:71-81 */
      double fixedCost = action > 0 ? d->fixed_order_cost : 0;
      double variableCost = d->unit_order_cost * action;
      double inventoryLevel = s->x + s->preq - randomDemand;
      double holdingCosts = d->holding_cost * jmax(inventoryLevel, 0);
      double penaltyCosts = d->penalty_cost * jmax(-inventoryLevel, 0);
      double totalCosts = fixedCost + variableCost + holdingCosts + penaltyCosts;
      return totalCosts;
    }
    case SDPGPU_FAMILY_CASH: {
      if (is_xr(d)) { /* CashConstraintXR.java:91-105; `action` is actionY, s->cash is iniR */
        double actionY = action;
        double variCost = d->unit_order_cost;
        double revenue = d->price * jmin(actionY, randomDemand);
        double act = actionY - s->x;
        double fixedCost = actionY > s->x ? d->fixed_order_cost : 0;
        double variableCost = variCost * act;
        double initCash = s->cash - variCost * s->x;
        double deposite = (initCash - fixedCost - variableCost) * (1 + d->deposit_rate);
        double inventoryLevel = actionY - randomDemand;
        double holdCosts = d->holding_cost * jmax(inventoryLevel, 0);
        double cashIncrement = (1 - d->overhead_rate) * revenue + deposite - holdCosts - overhead_at(c, s->period) - initCash;
        double salValue = s->period == c->T ? d->salvage_value * jmax(inventoryLevel, 0) : 0;
        cashIncrement += salValue;
        return cashIncrement;
      }
      double revenue = d->price * jmin(s->x + action, randomDemand);
      double fixedCost = action > 0 ? d->fixed_order_cost : 0;
      double variableCost = d->unit_order_cost * action;
      double inventoryLevel = s->x + action - randomDemand;
      double holdCosts = d->holding_cost * jmax(inventoryLevel, 0);
      double cashIncrement;
      if (d->cash_formula == 0) { /* CashConstraint.java:103-119 */
        double deposite = (s->cash - fixedCost - variableCost) * (1 + d->deposit_rate);
        cashIncrement = (1 - d->overhead_rate) * revenue + deposite - holdCosts - overhead_at(c, s->period) - s->cash;
      } else { /* CashConstraintTesting.java:117-132 */
        cashIncrement = revenue - fixedCost - variableCost - holdCosts - overhead_at(c, s->period);
      }
      double salValue = s->period == c->T ? d->salvage_value * jmax(inventoryLevel, 0) : 0;
      cashIncrement += salValue;
      double endCash = s->cash + cashIncrement;
      if (endCash < 0) {
        cashIncrement += d->penalty_cost * endCash;
      }
      return cashIncrement;
    }
    case SDPGPU_FAMILY_SURVIVAL: { /* cashSurvival.java:112-125 */
      double revenue = d->price * jmin(s->x + action, randomDemand);
      double fixedCost = action > 0 ? d->fixed_order_cost : 0;
      double variableCost = d->unit_order_cost * action;
      double deposite = (s->cash - fixedCost - variableCost) * (1 + d->deposit_rate);
      double inventoryLevel = s->x + action - randomDemand;
      double holdCosts = d->holding_cost * jmax(inventoryLevel, 0);
      double cashIncrement = revenue + deposite - holdCosts - overhead_at(c, s->period) - s->cash;
      double salValue = s->period == c->T ? d->salvage_value * jmax(inventoryLevel, 0) : 0;
      cashIncrement += salValue;
      return cashIncrement;
    }
    case SDPGPU_FAMILY_OVERDRAFT: { /* CashOverdraft.java:80-104 */
      double revenue = lost_sales_revenue(d->price, s->x + action, randomDemand);
      double fixedCost = action > 0 ? d->fixed_order_cost : 0;
      double variableCost = d->unit_order_cost * action;
      double inventoryLevel = s->x + action - randomDemand;
      double cashBalanceBefore = balance_before(s->cash - fixedCost, variableCost, overhead_at(c, s->period));
      double interest = overdraft_interest(d, cashBalanceBefore);
      double cashBalanceAfter = balance_after(cashBalanceBefore, interest, revenue);
      double cashIncrement = cashBalanceAfter - s->cash;
      double salValue = s->period == c->T ? d->salvage_value * jmax(inventoryLevel, 0) : 0;
      cashIncrement += salValue;
      return cashIncrement;
    }
    case SDPGPU_FAMILY_CASH_LEADTIME: { /* SingleProductLeadtime.java:82-104 */
      double revenue = lost_sales_revenue(d->price, s->x + s->preq, randomDemand);
      double variableCost = d->unit_order_cost * action;
      double inventoryLevel = s->x + s->preq - randomDemand;
      double cashBalanceBefore = balance_before(s->cash, variableCost, overhead_at(c, s->period));
      double interest = overdraft_interest(d, cashBalanceBefore);
      double cashBalanceAfter = balance_after(cashBalanceBefore, interest, revenue);
      double cashIncrement = cashBalanceAfter - s->cash;
      double salValue = s->period == c->T ? d->salvage_value * jmax(inventoryLevel, 0) : 0;
      cashIncrement += salValue;
      return cashIncrement;
    }
  }
  return 0;
}

/* `nextCash = Math.round(nextCash * 10) / 10.0` (CashConstraint.java:131), `/ 10` long division
 * (CashOverdraft.java:116), `* 100) / 100.0` (SingleProductLeadtime.java:117),
 * `Math.round(nextCash * 1) / 1` (CashConstraintTesting.java:146: long / int -> long). */
static double round_cash(const sdpgpu_desc* d, double nextCash) {
  int64_t r = jround(nextCash * d->cash_round_mult);
  if (d->cash_round_int_div) return (double)(r / (int64_t)d->cash_round_div); /* truncating */
  return (double)r / d->cash_round_div;
}

/* stateTransition.apply(state, action, randomDemand) */
static void transition(const ctx_t* c, const st_t* s, double action, double randomDemand, st_t* out) {
  const sdpgpu_desc* d = c->d;
  out->period = s->period + 1;
  out->cash = 0;
  out->preq = 0;
  out->preq2 = 0;
  if (g_user.trans) {
    sdpref_user_ctx u = {s->period, c->T, d->step, g_user.params};
    g_user.trans(&u, s->x, s->cash, s->preq, action, randomDemand, &out->x, &out->cash, &out->preq);
    return;
  }
  switch (d->family) {
    case SDPGPU_FAMILY_BACKORDER: { /* CLSP.java:255-260 == CLSPTesting.java:89-94 */
      double nextInventory = s->x + action - randomDemand;
      if (d->clamp_inventory) {
        nextInventory = nextInventory > d->max_inventory ? d->max_inventory : nextInventory;
        nextInventory = nextInventory < d->min_inventory ? d->min_inventory : nextInventory;
      }
      out->x = nextInventory;
      return;
    }
    case SDPGPU_FAMILY_LEADTIME: { /* Leadtime.java:61-68 (clamp lines 65-66 are commented out there) */
      double nextInventory = s->x + s->preq - randomDemand;
      if (d->clamp_inventory) {
        nextInventory = nextInventory > d->max_inventory ? d->max_inventory : nextInventory;
        nextInventory = nextInventory < d->min_inventory ? d->min_inventory : nextInventory;
      }
      out->x = nextInventory;
      if (d->lead_time == 2) { /* synthetic two-stage pipeline (BASELINE configs[3]): the queue shifts by one */
        out->preq = s->preq2;
        out->preq2 = action;
      } else {
        out->preq = action;
      }
      return;
    }
    case SDPGPU_FAMILY_CASH:       /* CashConstraint.java:122-133 */
      if (is_xr(d)) { /* CashConstraintXR.java:108-125; `action` is actionY */
        double variCost = d->unit_order_cost;
        double nextInventory = jmax(0, action - randomDemand);
        double initCash = s->cash - variCost * s->x;
        double nextCash = initCash + imm_value(c, s, action, randomDemand);
        nextCash = nextCash > d->max_cash ? d->max_cash : nextCash;
        nextCash = nextCash < d->min_cash ? d->min_cash : nextCash;
        nextInventory = nextInventory > d->max_inventory ? d->max_inventory : nextInventory;
        nextInventory = nextInventory < d->min_inventory ? d->min_inventory : nextInventory;
        nextCash = round_cash(d, nextCash); /* :120 `Math.round(nextCash * 1) / 1` */
        double nextR = nextCash + variCost * nextInventory;
        out->x = nextInventory;
        out->cash = nextR;
        return;
      }
      /* fall through */
    case SDPGPU_FAMILY_SURVIVAL:   /* cashSurvival.java:128-143 (same statements; `Math.round(nextCash * 1) / 1`) */
    case SDPGPU_FAMILY_OVERDRAFT: { /* CashOverdraft.java:107-118 */
      double nextInventory = end_inventory(s->x + action - randomDemand);
      double nextCash = s->cash + imm_value(c, s, action, randomDemand);
      nextCash = clamp_upper(nextCash, d->max_cash);
      nextCash = clamp_lower(nextCash, d->min_cash);
      nextInventory = clamp_upper(nextInventory, d->max_inventory);
      nextInventory = clamp_lower(nextInventory, d->min_inventory);
      nextCash = round_cash(d, nextCash);
      out->x = nextInventory;
      out->cash = nextCash;
      return;
    }
    case SDPGPU_FAMILY_CASH_LEADTIME: { /* SingleProductLeadtime.java:107-119 */
      double nextInventory = end_inventory(s->x + s->preq - randomDemand);
      double nextCash = s->cash + imm_value(c, s, action, randomDemand);
      double nextPreQ = action;
      nextCash = clamp_upper(nextCash, d->max_cash);
      nextCash = clamp_lower(nextCash, d->min_cash);
      nextInventory = clamp_upper(nextInventory, d->max_inventory);
      nextInventory = clamp_lower(nextInventory, d->min_inventory);
      nextCash = round_cash(d, nextCash);
      out->x = nextInventory;
      out->cash = nextCash;
      out->preq = nextPreQ;
      return;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * The recursion body, Recursion.java:129-161 (== CLSP.java:111-136, LeadtimeRecursion.java:49-73,
 * CashRecursion.java:98-138, CashLeadtimeRecursion.java:50-77).  `vlook` is what
 * getExpectedValue(newState) resolves to: a dense-table read or a recursive call.
 * ---------------------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------------------------
 * THE LOOP TEMPLATE.  Every recursion class of the reference is one loop, copied from class to class:
 *     Recursion.java:129-161 == CLSP.java:111-136 == LeadtimeRecursion.java:49-73 == CashRecursion.java:98-138 ==
 *     CashLeadtimeRecursion.java:50-77 == CashRecursionXR.java:91-121 == CashRecursionMultiLead.java:61-88
 * -- for every feasible action, accumulate over the demand list `q += p * imm; if (period < T) q += p [* discount] * V(next)`
 * in that interleaved order, then keep the first action that improves on the incumbent (`<` for MIN, `> val [+ 0.1]` for
 * MAX).  bellman_loop is that loop, once, for the whole oracle: eval_state (the single-item classes a1-a5 and
 * CashRecursionXR) and ml_value (CashRecursionMultiLead) both run it with their own lambdas plugged in.  ml_value reproduces
 * the outputs the reference itself records (KAT-1 and the others of MultiProductLeadtime.java:30-50,
 * tests/test_oracle_kat.py), so the loop the single-item classes are checked with -- accumulation order, discount
 * association, first-best rule, the +-Double.MAX_VALUE / action-0 initial incumbent -- is PINNED by reference-held
 * numbers; what no reference artefact covers is the arithmetic inside the single-item lambdas (imm / transition).
 *   prob[j]      dAndP[j][1]  (dAndP[j][2] for the two-product classes: GetPmfMulti stores the product)
 *   discount     multiplies the future term as `p * discount * V` (left to right); the classes without a discount factor
 *                pass 1.0: p * 1.0 == p exactly
 *   tol          0.1 for the two-product classes' `thisActionsValue > val + 0.1` (CashRecursionMultiLead.java:82), 0.0 for
 *                the strict `>` of the others (val + 0.0 == val for every val the loop can hold)
 * ---------------------------------------------------------------------------------------- */
typedef void (*bl_begin_fn)(void* env, int32_t action_index); /* `double orderQty = feasibleActions[i]` (Recursion.java:136) */
typedef double (*bl_imm_fn)(void* env, int32_t action_index, int32_t demand_index);
typedef double (*bl_next_fn)(void* env, int32_t action_index, int32_t demand_index); /* getExpectedValue(stateTransition.apply(...)) */

static inline __attribute__((always_inline)) void bellman_loop(int32_t n_actions, int32_t n_demands, const double* prob, double discount, int has_future,
                                int maxdir, double tol, bl_begin_fn begin, bl_imm_fn imm, bl_next_fn next, void* env,
                                double* val_out, int32_t* best_out) {
  double val = maxdir ? -DBL_MAX : DBL_MAX; /* Recursion.java:132-133 */
  int32_t best = 0;                         /* bestOrderQty = 0, :134 (index 0 <-> quantity 0) */
  for (int32_t i = 0; i < n_actions; i++) {
    begin(env, i);
    double thisQValue = 0;
    for (int32_t j = 0; j < n_demands; j++) {
      double thisDValue = imm(env, i, j);
      thisQValue += prob[j] * thisDValue;                             /* Recursion.java:139, CashRecursion.java:117 */
      if (has_future) thisQValue += prob[j] * discount * next(env, i, j); /* :140-143, CashRecursion.java:118-121 */
    }
    if (!maxdir) { /* Recursion.java:146-151 */
      if (thisQValue < val) {
        val = thisQValue;
        best = i;
      }
    } else { /* :152-157; CashRecursionMultiLead.java:82 with tol = 0.1 */
      if (thisQValue > val + tol) {
        val = thisQValue;
        best = i;
      }
    }
  }
  *val_out = val;
  *best_out = best;
}

typedef double (*vlook_fn)(void* env, const st_t* next);

/* eval_state's lambdas for the loop template: action i / demand j of the state being evaluated */
typedef struct es_env {
  const ctx_t* c;
  const st_t* s;
  const double* dem;
  vlook_fn vlook;
  void* venv;
  double orderQty; /* feasibleActions[i] of the action the loop is at */
} es_env;
static inline __attribute__((always_inline)) void es_begin(void* env, int32_t i) {
  es_env* e = (es_env*)env;
  e->orderQty = action_value(e->c, e->s, i);
}
static inline __attribute__((always_inline)) double es_imm(void* env, int32_t i, int32_t j) {
  const es_env* e = (const es_env*)env;
  (void)i;
  return imm_value(e->c, e->s, e->orderQty, e->dem[j]);
}
static inline __attribute__((always_inline)) double es_next(void* env, int32_t i, int32_t j) {
  const es_env* e = (const es_env*)env;
  (void)i;
  st_t newState;
  transition(e->c, e->s, e->orderQty, e->dem[j], &newState);
  return e->vlook(e->venv, &newState);
}

static void eval_state(const ctx_t* c, const st_t* s, vlook_fn vlook, void* env, double* val_out,
                       double* best_out, int32_t* bestk_out, int64_t* cells) {
  const sdpgpu_desc* d = c->d;
  int32_t nA = n_actions(c, s);
  int32_t t = s->period - 1;
  int32_t n = c->off[t + 1] - c->off[t];
  const double* dem = c->pd + c->off[t];
  const double* prob = c->pp + c->off[t];
  int maxdir = d->direction == SDPGPU_MAX;
  int cash_loop = d->family == SDPGPU_FAMILY_CASH || d->family == SDPGPU_FAMILY_OVERDRAFT;
  double val = maxdir ? -DBL_MAX : DBL_MAX; /* Recursion.java:132-133 */
  double bestOrderQty = 0;                  /* Recursion.java:134 */
  int32_t bestk = 0;
  if (d->family == SDPGPU_FAMILY_SURVIVAL) {
    /* RiskRecursion.getSurvProb, RiskRecursion.java:65-108 (== CashRecursion.java:143-194, whose line 174 also
     * multiplies by discountFactor: p * 1.0 == p, so discount_factor = 1 is RiskRecursion exactly). */
    val = -DBL_MAX; /* :70 */
    for (int32_t i = 0; i < nA; i++) {
      double orderQty = action_value(c, s, i);
      double thisQProb = 0;
      for (int32_t j = 0; j < n; j++) {
        double randomDemand = dem[j];
        double dProb = prob[j];
        if (s->period == c->T) { /* :82-86 */
          double thisDFinalCash = s->cash + imm_value(c, s, orderQty, randomDemand);
          double thisDProb = thisDFinalCash >= 0 ? 1 : 0;
          thisQProb += dProb * thisDProb;
        }
        if (s->period < c->T) { /* :87-97 */
          st_t newState;
          transition(c, s, orderQty, dem[j], &newState);
          double thisDProb = 0;
          if (newState.cash < 0) {
            thisDProb = 0;
          } else {
            thisDProb = vlook(env, &newState);
          }
          thisQProb += prob[j] * d->discount_factor * thisDProb;
        }
      }
      if (thisQProb > val) { /* :101-104 */
        val = thisQProb;
        bestOrderQty = orderQty;
        bestk = i;
      }
    }
    if (cells) *cells += (int64_t)nA * n;
    *val_out = val;
    if (best_out) *best_out = bestOrderQty;
    if (bestk_out) *bestk_out = bestk;
    return;
  }
  /* Recursion.java:129-161 / CashRecursion.java:98-138 (p * discountFactor * V; the classes without a discount factor:
   * discount 1.0) through the loop template */
  es_env ev = {c, s, dem, vlook, env, 0};
  bellman_loop(nA, n, prob, cash_loop ? d->discount_factor : 1.0, s->period < c->T, maxdir, 0.0, es_begin, es_imm, es_next,
               &ev, &val, &bestk);
  bestOrderQty = nA > 0 ? action_value(c, s, bestk) : 0; /* (empty list: bestOrderQty stays 0, :134) */
  if (cells) *cells += (int64_t)nA * n;
  *val_out = val;
  if (best_out) *best_out = bestOrderQty;
  if (bestk_out) *bestk_out = bestk;
}

/* ------------------------------------------------------------------------------------------
 * Dense layout
 * ---------------------------------------------------------------------------------------- */
static int64_t cash_key(const sdpgpu_desc* d, double cash) {
  if (d->cash_round_int_div) return (int64_t)cash;
  return jround(cash * d->cash_round_mult);
}
static double cash_of_key(const sdpgpu_desc* d, int64_t k) {
  if (d->cash_round_int_div) return (double)k;
  return (double)k / d->cash_round_div;
}

int sdpref_layout(const sdpgpu_desc* d, const int32_t* pmf_off, const double* pmf_d, sdpref_grid* g) {
  if (!d || d->periods < 1 || !(d->step >= 1) || d->step != floor(d->step)) return 1;
  int32_t T = d->periods;
  int64_t nc = 1, k_lo = 0, nq = 1;
  if (has_cash(d->family)) {
    if (!d->cash_round_int_div && d->cash_round_mult != d->cash_round_div) return 4;
    k_lo = cash_key(d, round_cash(d, d->min_cash));
    int64_t k_hi = cash_key(d, round_cash(d, d->max_cash));
    nc = k_hi - k_lo + 1;
  }
  if (has_preq(d->family)) nq = full_action_count(d);
  int64_t nq1 = nq;
  if (d->lead_time == 2) {
    if (d->family != SDPGPU_FAMILY_LEADTIME || !d->clamp_inventory) return 4;
    nq = nq1 * nq1;
  }
  double lo = d->min_inventory, hi = d->max_inventory;
  if (!d->clamp_inventory) {
    if (d->family != SDPGPU_FAMILY_BACKORDER && d->family != SDPGPU_FAMILY_LEADTIME) return 4;
    lo = hi = d->ini_inventory;
  }
  for (int32_t t = 0; t < T; t++) {
    g[t].x_lo = lo;
    g[t].nx = (int64_t)((hi - lo) / d->step) + 1;
    g[t].nc = nc;
    g[t].nq = nq;
    g[t].nq1 = nq1;
    g[t].k_lo = k_lo;
    if (!d->clamp_inventory) { /* box of period t+2 grown over all actions and demands */
      double dmin = pmf_d[pmf_off[t]], dmax = dmin;
      for (int32_t j = pmf_off[t]; j < pmf_off[t + 1]; j++) {
        if (pmf_d[j] < dmin) dmin = pmf_d[j];
        if (pmf_d[j] > dmax) dmax = pmf_d[j];
      }
      double qmax = (double)(full_action_count(d) - 1) * d->step;
      lo = lo - dmax;
      hi = hi + qmax - dmin;
    }
  }
  return 0;
}

/* NULL counts clears the registration.  The arrays must outlive the solves that use them. */
int sdpref_set_action_counts(const sdpgpu_desc* d, const int32_t* pmf_off, const double* pmf_d, const int32_t* counts,
                             const int64_t* off) {
  free(g_counts.grids);
  g_counts.grids = NULL;
  g_counts.counts = NULL;
  g_counts.off = NULL;
  g_counts.T = 0;
  if (!counts) return 0;
  g_counts.grids = (sdpref_grid*)malloc(sizeof(sdpref_grid) * (size_t)d->periods);
  int rc = sdpref_layout(d, pmf_off, pmf_d, g_counts.grids);
  if (rc) {
    free(g_counts.grids);
    g_counts.grids = NULL;
    return rc;
  }
  g_counts.counts = counts;
  g_counts.off = off;
  g_counts.T = d->periods;
  return 0;
}

static int64_t index_of(const sdpgpu_desc* d, const sdpref_grid* g, const st_t* s) {
  double qx = (s->x - g->x_lo) / d->step;
  int64_t ix = (int64_t)qx;
  if ((double)ix != qx || ix < 0 || ix >= g->nx) return -1;
  int64_t ic = 0, iq = 0;
  if (has_cash(d->family)) {
    double cash = s->cash;
    if (is_xr(d)) { /* key (x, R): the grid cash is the balance the transition rounded before it formed R */
      cash = round_cash(d, s->cash - d->unit_order_cost * s->x);
      if (cash + d->unit_order_cost * s->x != s->cash) return -1;
    }
    int64_t k = cash_key(d, cash);
    if (cash_of_key(d, k) != cash) return -1;
    ic = k - g->k_lo;
    if (ic < 0 || ic >= g->nc) return -1;
  }
  if (has_preq(d->family)) {
    double qq = s->preq / d->step;
    iq = (int64_t)qq;
    if ((double)iq != qq || iq < 0 || iq >= g->nq1) return -1;
  }
  if (d->lead_time == 2) {
    double qq = s->preq2 / d->step;
    int64_t iq2 = (int64_t)qq;
    if ((double)iq2 != qq || iq2 < 0 || iq2 >= g->nq / g->nq1) return -1;
    iq += iq2 * g->nq1;
  }
  return (iq * g->nx + ix) * g->nc + ic;
}

static void state_of(const sdpgpu_desc* d, const sdpref_grid* g, int32_t period, int64_t idx, st_t* s) {
  int64_t ic = idx % g->nc;
  int64_t ix = (idx / g->nc) % g->nx;
  int64_t iq = idx / (g->nc * g->nx);
  s->period = period;
  s->x = g->x_lo + (double)ix * d->step;
  s->cash = has_cash(d->family) ? cash_of_key(d, g->k_lo + ic) : 0;
  if (is_xr(d)) s->cash = s->cash + d->unit_order_cost * s->x; /* R = nextCash + variCost * nextInventory */
  s->preq = has_preq(d->family) ? (double)(iq % g->nq1) * d->step : 0;
  s->preq2 = (double)(iq / g->nq1) * d->step;
}

typedef struct dense_env {
  const sdpgpu_desc* d;
  const sdpref_grid* gnext;
  const double* vnext;
  int err;
} dense_env;

static double dense_look(void* env, const st_t* next) {
  dense_env* e = (dense_env*)env;
  int64_t idx = index_of(e->d, e->gnext, next);
  if (idx < 0) {
    e->err = 1;
    return 0;
  }
  return e->vnext[idx];
}

typedef struct job {
  const ctx_t* c;
  const sdpref_grid* gcur;
  const sdpref_grid* gnext;
  int32_t period;
  const double* vnext;
  double* vcur;
  int32_t* pol;
  int64_t lo, hi;
  int64_t cells;
  int err;
} job_t;

static void* period_worker(void* arg) {
  job_t* jb = (job_t*)arg;
  dense_env env = {jb->c->d, jb->gnext, jb->vnext, 0};
  for (int64_t idx = jb->lo; idx < jb->hi; idx++) {
    st_t s;
    state_of(jb->c->d, jb->gcur, jb->period, idx, &s);
    double val;
    int32_t bestk;
    eval_state(jb->c, &s, dense_look, &env, &val, NULL, &bestk, &jb->cells);
    jb->vcur[idx] = val;
    jb->pol[idx] = bestk;
  }
  jb->err = env.err;
  return NULL;
}

static int run_period(const ctx_t* c, const sdpref_grid* grids, int32_t period, const double* vnext,
                      double* vcur, int32_t* pol, int64_t lo, int64_t hi, int32_t nthreads, int64_t* cells) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  int64_t n = hi - lo;
  if (n < nthreads) nthreads = n > 0 ? (int32_t)n : 1;
  job_t jobs[256];
  pthread_t th[256];
  for (int32_t i = 0; i < nthreads; i++) {
    jobs[i].c = c;
    jobs[i].gcur = &grids[period - 1];
    jobs[i].gnext = period < c->T ? &grids[period] : NULL;
    jobs[i].period = period;
    jobs[i].vnext = vnext;
    jobs[i].vcur = vcur;
    jobs[i].pol = pol;
    jobs[i].lo = lo + n * i / nthreads;
    jobs[i].hi = lo + n * (i + 1) / nthreads;
    jobs[i].cells = 0;
    jobs[i].err = 0;
  }
  if (nthreads == 1) {
    period_worker(&jobs[0]);
  } else {
    for (int32_t i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, period_worker, &jobs[i]);
    for (int32_t i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  }
  int err = 0;
  for (int32_t i = 0; i < nthreads; i++) {
    if (cells) *cells += jobs[i].cells;
    err |= jobs[i].err;
  }
  return err ? 5 : 0;
}

int sdpref_period(const sdpgpu_desc* d, const int32_t* pmf_off, const double* pmf_d, const double* pmf_p,
                  const double* overhead, int32_t period, const double* v_next, double* v_cur,
                  int32_t* pol_cur, int64_t lo, int64_t hi, int32_t nthreads, int64_t* cells_out) {
  if (period < 1 || period > d->periods) return 1;
  sdpref_grid* grids = (sdpref_grid*)malloc(sizeof(sdpref_grid) * (size_t)d->periods);
  int rc = sdpref_layout(d, pmf_off, pmf_d, grids);
  if (rc == 0) {
    ctx_t c = {d, pmf_off, pmf_d, pmf_p, overhead, d->periods};
    int64_t cells = 0;
    rc = run_period(&c, grids, period, v_next, v_cur, pol_cur, lo, hi, nthreads, &cells);
    if (cells_out) *cells_out = cells;
  }
  free(grids);
  return rc;
}

int sdpref_solve(const sdpgpu_desc* d, const int32_t* pmf_off, const double* pmf_d, const double* pmf_p,
                 const double* overhead, double* values, int32_t* policy, const int64_t* values_off,
                 int32_t nthreads, int64_t* cells_out) {
  sdpref_grid* grids = (sdpref_grid*)malloc(sizeof(sdpref_grid) * (size_t)d->periods);
  int rc = sdpref_layout(d, pmf_off, pmf_d, grids);
  int64_t cells = 0;
  if (rc == 0) {
    ctx_t c = {d, pmf_off, pmf_d, pmf_p, overhead, d->periods};
    for (int32_t period = d->periods; period >= 1 && rc == 0; period--) {
      const sdpref_grid* g = &grids[period - 1];
      int64_t S = g->nx * g->nc * g->nq;
      const double* vnext = period < d->periods ? values + values_off[period] : NULL;
      rc = run_period(&c, grids, period, vnext, values + values_off[period - 1], policy + values_off[period - 1],
                      0, S, nthreads, &cells);
    }
  }
  if (cells_out) *cells_out = cells;
  free(grids);
  return rc;
}

int sdpref_eval_states(const sdpgpu_desc* d, const int32_t* pmf_off, const double* pmf_d, const double* pmf_p,
                       const double* overhead, int32_t period, const double* v_next, int64_t n,
                       const double* x, const double* cash, const double* preq, const double* preq2,
                       double* out_value, int32_t* out_action) {
  if (period < 1 || period > d->periods) return 1;
  sdpref_grid* grids = (sdpref_grid*)malloc(sizeof(sdpref_grid) * (size_t)d->periods);
  int rc = sdpref_layout(d, pmf_off, pmf_d, grids);
  if (rc == 0) {
    ctx_t c = {d, pmf_off, pmf_d, pmf_p, overhead, d->periods};
    dense_env env = {d, period < d->periods ? &grids[period] : NULL, v_next, 0};
    for (int64_t i = 0; i < n; i++) {
      st_t s = {period, x[i], cash ? cash[i] : 0, preq ? preq[i] : 0, (preq2 && d->lead_time == 2) ? preq2[i] : 0};
      double val;
      int32_t bestk;
      eval_state(&c, &s, dense_look, &env, &val, NULL, &bestk, NULL);
      out_value[i] = val;
      out_action[i] = bestk;
    }
    if (env.err) rc = 5;
  }
  free(grids);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * Forward rollout of a computed policy along demand paths: the inner loops of
 * Simulation.simulateSDPGivenSamplNum (Simulation.java:59-69) and CashSimulation (:101-112).
 * `policy` holds action INDICES per period (values_off as in sdpref_solve); getAction on a state the
 * tables do not hold (an off-grid period-1 state) re-enters the recursion, Simulation.java:62.
 * ---------------------------------------------------------------------------------------- */
int sdpref_simulate(const sdpgpu_desc* d, const int32_t* pmf_off, const double* pmf_d, const double* pmf_p,
                    const double* overhead, const double* values, const int32_t* policy, const int64_t* values_off,
                    int64_t n_paths, const double* demand, const double* discount, double ini_x, double ini_cash,
                    double ini_preq, double* out_sum, uint8_t* out_valid) {
  sdpref_grid* grids = (sdpref_grid*)malloc(sizeof(sdpref_grid) * (size_t)d->periods);
  int rc = sdpref_layout(d, pmf_off, pmf_d, grids);
  if (rc) {
    free(grids);
    return rc;
  }
  ctx_t c = {d, pmf_off, pmf_d, pmf_p, overhead, d->periods};
  int32_t T = d->periods;
  if (d->family == SDPGPU_FAMILY_SURVIVAL) {
    /* RiskSimulation.simulateLostSale, RiskSimulation.java:213-234: out_sum[i] = simuValues[i] (1 when the path
     * held negative cash at some point), out_valid[i] = 1 | 2 * (a demand was lost at some point). */
    for (int64_t i = 0; i < n_paths; i++) {
      st_t state = {1, ini_x, ini_cash, 0, 0};
      int countBefore = 0, countBeforeBankrupt = 0, valid = 1;
      double simuValue = 0;
      for (int32_t t = 0; t < T; t++) {
        int64_t idx = index_of(d, &grids[t], &state);
        double optQ;
        if (idx >= 0) {
          optQ = action_value(&c, &state, policy[values_off[t] + idx]);
        } else if (t == 0) { /* recursion.getSurvProb(state); recursion.getAction(state) */
          dense_env env = {d, T > 1 ? &grids[1] : NULL, T > 1 ? values + values_off[1] : NULL, 0};
          double val;
          eval_state(&c, &state, dense_look, &env, &val, &optQ, NULL, NULL);
        } else {
          valid = 0;
          break;
        }
        if (state.cash < 0) optQ = 0; /* :221-222 */
        double randomDemand = demand[i * T + t];
        if (state.x + optQ < randomDemand && !countBefore) countBefore = 1; /* :224-227 */
        double thisValue = state.cash + imm_value(&c, &state, optQ, randomDemand);
        st_t next;
        transition(&c, &state, optQ, randomDemand, &next);
        state = next;
        if (thisValue < 0 && !countBeforeBankrupt) { /* :230-233 */
          simuValue = 1;
          countBeforeBankrupt = 1;
        }
      }
      out_sum[i] = simuValue;
      out_valid[i] = (uint8_t)(valid | (countBefore << 1));
    }
    free(grids);
    return 0;
  }
  for (int64_t i = 0; i < n_paths; i++) {
    double sum = 0;
    st_t state = {1, ini_x, has_cash(d->family) ? ini_cash : 0, has_preq(d->family) ? ini_preq : 0,
                  d->lead_time == 2 ? d->ini_preq2 : 0};
    int valid = 1;
    for (int32_t t = 0; t < T && valid; t++) {
      int64_t idx = index_of(d, &grids[t], &state);
      double optQ;
      if (idx >= 0) {
        optQ = action_value(&c, &state, policy[values_off[t] + idx]);
      } else if (t == 0) { /* recursion.getExpectedValue(state); recursion.getAction(state) */
        dense_env env = {d, T > 1 ? &grids[1] : NULL, T > 1 ? values + values_off[1] : NULL, 0};
        double val;
        eval_state(&c, &state, dense_look, &env, &val, &optQ, NULL, NULL);
      } else {
        valid = 0;
        break;
      }
      double randomDemand = demand[i * T + t];
      double thisValue = imm_value(&c, &state, optQ, randomDemand);
      sum += discount[t] * thisValue;
      if (t + 1 < T) {
        st_t next;
        transition(&c, &state, optQ, randomDemand, &next);
        if (index_of(d, &grids[t + 1], &next) < 0) valid = 0;
        state = next;
      }
    }
    out_sum[i] = sum;
    out_valid[i] = (uint8_t)valid;
  }
  free(grids);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Forward reachable set: the key set of cacheActions after getExpectedValue(initialState).
 * ---------------------------------------------------------------------------------------- */
int sdpref_reachable(const sdpgpu_desc* d, const int32_t* pmf_off, const double* pmf_d, const double* pmf_p,
                     const double* overhead, uint8_t* mask, const int64_t* values_off) {
  sdpref_grid* grids = (sdpref_grid*)malloc(sizeof(sdpref_grid) * (size_t)d->periods);
  int rc = sdpref_layout(d, pmf_off, pmf_d, grids);
  if (rc) {
    free(grids);
    return rc;
  }
  ctx_t c = {d, pmf_off, pmf_d, pmf_p, overhead, d->periods};
  int32_t T = d->periods;
  for (int32_t t = 0; t < T; t++) {
    const sdpref_grid* g = &grids[t];
    memset(mask + values_off[t], 0, (size_t)(g->nx * g->nc * g->nq));
  }
  st_t ini = {1, d->ini_inventory, has_cash(d->family) ? d->ini_cash : 0, has_preq(d->family) ? d->ini_preq : 0,
              d->lead_time == 2 ? d->ini_preq2 : 0};
  int64_t i0 = index_of(d, &grids[0], &ini);
  for (int32_t period = 1; period <= T; period++) {
    const sdpref_grid* g = &grids[period - 1];
    int64_t S = g->nx * g->nc * g->nq;
    /* period 1: the initial state alone (it may lie off the grid) */
    for (int64_t idx = (period == 1 ? -1 : 0); idx < (period == 1 ? 0 : S); idx++) {
      st_t s;
      if (period == 1) {
        s = ini;
        if (i0 >= 0) mask[values_off[0] + i0] = 1;
      } else {
        if (!mask[values_off[period - 1] + idx]) continue;
        state_of(d, g, period, idx, &s);
      }
      if (period == T) continue;
      int32_t nA = n_actions(&c, &s);
      int32_t n = pmf_off[period] - pmf_off[period - 1];
      for (int32_t i = 0; i < nA; i++)
        for (int32_t j = 0; j < n; j++) {
          st_t nx;
          transition(&c, &s, action_value(&c, &s, i), pmf_d[pmf_off[period - 1] + j], &nx);
          if (d->family == SDPGPU_FAMILY_SURVIVAL && nx.cash < 0) continue; /* RiskRecursion.java:90-92: not visited */
          int64_t ni = index_of(d, &grids[period], &nx);
          if (ni < 0) {
            free(grids);
            return 5;
          }
          mask[values_off[period] + ni] = 1;
        }
    }
  }
  free(grids);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Literal memoised recursion (the reference's actual control flow): cacheValues.computeIfAbsent
 * (Recursion.java:90) with a hash map in place of the skip list (keys compare by exact ==,
 * State.java:66-71).
 * ---------------------------------------------------------------------------------------- */
typedef struct mentry {
  st_t key;
  double value, action;
  int used;
} mentry;

typedef struct memo {
  const ctx_t* c;
  mentry* tab;
  int64_t cap, n;
  int64_t cells;
} memo_t;

static uint64_t hash_st(const st_t* s) {
  uint64_t h = 1469598103934665603ull;
  uint64_t w[5];
  double x = s->x + 0.0, ca = s->cash + 0.0, pq = s->preq + 0.0, pq2 = s->preq2 + 0.0; /* -0.0 -> +0.0 */
  w[0] = (uint64_t)s->period;
  memcpy(&w[1], &x, 8);
  memcpy(&w[2], &ca, 8);
  memcpy(&w[3], &pq, 8);
  memcpy(&w[4], &pq2, 8);
  for (int i = 0; i < 5; i++) {
    h ^= w[i];
    h *= 1099511628211ull;
    h ^= h >> 29;
  }
  return h;
}
static int eq_st(const st_t* a, const st_t* b) {
  return a->period == b->period && a->x == b->x && a->cash == b->cash && a->preq == b->preq && a->preq2 == b->preq2;
}

static mentry* memo_find(memo_t* m, const st_t* s) {
  uint64_t i = hash_st(s) & (uint64_t)(m->cap - 1);
  while (m->tab[i].used) {
    if (eq_st(&m->tab[i].key, s)) return &m->tab[i];
    i = (i + 1) & (uint64_t)(m->cap - 1);
  }
  return &m->tab[i];
}

static void memo_grow(memo_t* m) {
  mentry* old = m->tab;
  int64_t ocap = m->cap;
  m->cap *= 2;
  m->tab = (mentry*)calloc((size_t)m->cap, sizeof(mentry));
  for (int64_t i = 0; i < ocap; i++)
    if (old[i].used) *memo_find(m, &old[i].key) = old[i];
  free(old);
}

static double memo_value(void* env, const st_t* s) {
  memo_t* m = (memo_t*)env;
  mentry* e = memo_find(m, s);
  if (e->used) return e->value;
  double val, best;
  eval_state(m->c, s, memo_value, m, &val, &best, NULL, &m->cells);
  if ((m->n + 1) * 2 > m->cap) memo_grow(m);
  e = memo_find(m, s); /* the table may have been rebuilt by nested inserts */
  e->key = *s;
  e->value = val;
  e->action = best;
  e->used = 1;
  m->n++;
  return val;
}

int sdpref_memo(const sdpgpu_desc* d, const int32_t* pmf_off, const double* pmf_d, const double* pmf_p,
                const double* overhead, double* root_value, double* root_action, int64_t cap,
                int32_t* out_period, double* out_x, double* out_cash, double* out_preq, double* out_preq2,
                double* out_value, double* out_action, int64_t* n_out) {
  ctx_t c = {d, pmf_off, pmf_d, pmf_p, overhead, d->periods};
  memo_t m;
  m.c = &c;
  m.cap = 1 << 12;
  m.n = 0;
  m.cells = 0;
  m.tab = (mentry*)calloc((size_t)m.cap, sizeof(mentry));
  st_t ini = {1, d->ini_inventory, has_cash(d->family) ? d->ini_cash : 0, has_preq(d->family) ? d->ini_preq : 0,
              d->lead_time == 2 ? d->ini_preq2 : 0};
  double v = memo_value(&m, &ini);
  if (root_value) *root_value = v;
  if (root_action) *root_action = memo_find(&m, &ini)->action;
  int rc = 0;
  if (n_out) *n_out = m.n;
  if (out_period) {
    if (m.n > cap) {
      rc = 6;
    } else {
      int64_t k = 0;
      for (int64_t i = 0; i < m.cap; i++)
        if (m.tab[i].used) {
          out_period[k] = m.tab[i].key.period;
          out_x[k] = m.tab[i].key.x;
          out_cash[k] = m.tab[i].key.cash;
          out_preq[k] = m.tab[i].key.preq;
          if (out_preq2) out_preq2[k] = m.tab[i].key.preq2;
          out_value[k] = m.tab[i].value;
          out_action[k] = m.tab[i].action;
          k++;
        }
    }
  }
  free(m.tab);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * KAT family: CashRecursionMultiLead.java:54-90 + MultiProductLeadtime.java:150-223.
 * ---------------------------------------------------------------------------------------- */
typedef struct mst {
  int32_t period;
  double i1, i2, q1, q2, cash;
} mst_t;

typedef struct mlentry {
  mst_t key;
  double value;
  int32_t a1, a2;
  int used;
} mlentry;

typedef struct mlmemo {
  const sdpref_multilead* k;
  mlentry* tab;
  int64_t cap, n, cells;
} mlmemo;

static uint64_t hash_mst(const mst_t* s) {
  uint64_t h = 1469598103934665603ull, w[6];
  double v[5] = {s->i1 + 0.0, s->i2 + 0.0, s->q1 + 0.0, s->q2 + 0.0, s->cash + 0.0};
  w[0] = (uint64_t)s->period;
  memcpy(&w[1], v, 40);
  for (int i = 0; i < 6; i++) {
    h ^= w[i];
    h *= 1099511628211ull;
    h ^= h >> 29;
  }
  return h;
}
static mlentry* ml_find(mlmemo* m, const mst_t* s) {
  uint64_t i = hash_mst(s) & (uint64_t)(m->cap - 1);
  while (m->tab[i].used) {
    const mst_t* k = &m->tab[i].key;
    if (k->period == s->period && k->i1 == s->i1 && k->i2 == s->i2 && k->q1 == s->q1 && k->q2 == s->q2 &&
        k->cash == s->cash)
      return &m->tab[i];
    i = (i + 1) & (uint64_t)(m->cap - 1);
  }
  return &m->tab[i];
}
static void ml_grow(mlmemo* m) {
  mlentry* old = m->tab;
  int64_t ocap = m->cap;
  m->cap *= 2;
  m->tab = (mlentry*)calloc((size_t)m->cap, sizeof(mlentry));
  for (int64_t i = 0; i < ocap; i++)
    if (old[i].used) *ml_find(m, &old[i].key) = old[i];
  free(old);
}

/* read-out of a finished memo (same row layout as the product's sdpgpu_multi_table) */
static sdpgpu_multi_table* g_ref_table = NULL;
void sdpref_multi_set_table(sdpgpu_multi_table* t) { g_ref_table = t; }
static void dump_memo(const mlentry* tab, int64_t cap, int64_t n) {
  sdpgpu_multi_table* t = g_ref_table;
  if (!t) return;
  t->rows = n;
  if (n > t->capacity) return;
  int64_t r = 0;
  for (int64_t i = 0; i < cap; i++)
    if (tab[i].used) {
      t->period[r] = tab[i].key.period;
      t->i1[r] = tab[i].key.i1;
      t->i2[r] = tab[i].key.i2;
      t->q1[r] = tab[i].key.q1;
      t->q2[r] = tab[i].key.q2;
      t->cash[r] = tab[i].key.cash;
      t->value[r] = tab[i].value;
      t->a1[r] = tab[i].a1;
      t->a2[r] = tab[i].a2;
      r++;
    }
}

/* MultiProductLeadtime.java:162-199 */
static double ml_imm(const sdpref_multilead* k, const mst_t* s, int32_t a1, int32_t a2, int32_t dm1, int32_t dm2) {
  double action1 = a1, action2 = a2, demand1 = dm1, demand2 = dm2;
  double preQ1 = s->q1, preQ2 = s->q2;
  double endInventory1 = end_inventory(s->i1 + preQ1 - demand1);
  double endInventory2 = end_inventory(s->i2 + preQ2 - demand2);
  double revenue1 = lost_sales_revenue(k->price[0], demand1, s->i1 + preQ1); /* :170 writes Math.min(demand1, stock) */
  double revenue2 = lost_sales_revenue(k->price[1], s->i2 + preQ2, demand2);
  double revenue = revenue1 + revenue2;
  double orderingCost1 = k->vari_cost[0] * action1;
  double orderingCost2 = k->vari_cost[1] * action2;
  double orderingCosts = orderingCost1 + orderingCost2;
  double salValue = 0;
  if (s->period == k->T) salValue = k->sal_value[0] * endInventory1 + k->sal_value[1] * endInventory2;
  int t = s->period - 1;
  double cashBalanceBefore = balance_before(s->cash, orderingCosts, k->overhead[t]);
  double interest = interest_piecewise(k->r0, k->r1, k->r2, k->interest_free, k->limit, cashBalanceBefore);
  double cashBalanceAfter = balance_after(cashBalanceBefore, interest, revenue) + salValue;
  double cashIncrement = cashBalanceAfter - s->cash;
  return cashIncrement;
}

/* MultiProductLeadtime.java:203-223 (the one-sided clamps at :217-218 are the reference's). */
static void ml_trans(const sdpref_multilead* k, const mst_t* s, int32_t a1, int32_t a2, int32_t dm1, int32_t dm2,
                     mst_t* out) {
  double nextPreQ1 = a1, nextPreQ2 = a2;
  double endInventory1 = s->i1 + s->q1 - (double)dm1;
  endInventory1 = end_inventory(endInventory1);
  double endInventory2 = s->i2 + s->q2 - (double)dm2;
  endInventory2 = end_inventory(endInventory2);
  double nextCash = s->cash + ml_imm(k, s, a1, a2, dm1, dm2);
  nextCash = clamp_upper(nextCash, k->max_cash);
  nextCash = clamp_lower(nextCash, k->min_cash);
  endInventory1 = clamp_upper(endInventory1, k->max_inventory);
  endInventory2 = clamp_lower(endInventory2, k->min_inventory);
  if (k->cash_int_cast) nextCash = (double)jd2i(nextCash); /* :219, commented out in the file as it stands */
  endInventory1 = (double)jd2i(endInventory1);
  endInventory2 = (double)jd2i(endInventory2);
  out->period = s->period + 1;
  out->i1 = endInventory1;
  out->i2 = endInventory2;
  out->q1 = nextPreQ1;
  out->q2 = nextPreQ2;
  out->cash = nextCash;
}

/* CashRecursionMultiLead.java:54-90 through the loop template (bellman_loop, above eval_state): action index
 * i = ai * Qbound + aj enumerates buildActionList's double loop (MultiProductLeadtime.java:150-158), demand index
 * j = di * n2 + dj the rows of GetPmfMulti.getPmf (GetPmfMulti.java:157-172), whose third column is the product of the two
 * probabilities. */
/* `thisActionsValue > val + 0.1` (CashRecursionMultiLead.java:80).  Tests may set the slack to 0 to turn the family's
 * arg-max into a true one (the bridge to the single-product family, tests/test_oracle_kat.py); NOT thread-safe. */
static double g_ml_tolerance = 0.1;
void sdpref_kat_set_tolerance(double tol) { g_ml_tolerance = tol; }

static double ml_value(mlmemo* m, const mst_t* s);
typedef struct ml_env {
  mlmemo* m;
  const mst_t* s;
} ml_env;
static void ml_cb_begin(void* env, int32_t i) {
  (void)env;
  (void)i;
}
static double ml_cb_imm(void* env, int32_t i, int32_t j) {
  const ml_env* e = (const ml_env*)env;
  const sdpref_multilead* k = e->m->k;
  e->m->cells++;
  return ml_imm(k, e->s, i / k->q_bound, i % k->q_bound, jd2i(k->v1[j / k->n2]), jd2i(k->v2[j % k->n2]));
}
static double ml_cb_next(void* env, int32_t i, int32_t j) {
  const ml_env* e = (const ml_env*)env;
  const sdpref_multilead* k = e->m->k;
  mst_t ns;
  ml_trans(k, e->s, i / k->q_bound, i % k->q_bound, jd2i(k->v1[j / k->n2]), jd2i(k->v2[j % k->n2]), &ns);
  return ml_value(e->m, &ns);
}

static double ml_value(mlmemo* m, const mst_t* s) {
  mlentry* e = ml_find(m, s);
  if (e->used) return e->value;
  const sdpref_multilead* k = m->k;
  double prob[256]; /* dAndP[j][2] = p1 * p2 (n1, n2 <= 16) */
  for (int32_t i = 0; i < k->n1; i++)
    for (int32_t j = 0; j < k->n2; j++) prob[i * k->n2 + j] = k->p1[i] * k->p2[j];
  double val;
  int32_t best;
  ml_env ev = {m, s};
  bellman_loop(k->q_bound * k->q_bound, k->n1 * k->n2, prob, k->discount, s->period < k->T, 1, g_ml_tolerance, ml_cb_begin, ml_cb_imm,
               ml_cb_next, &ev, &val, &best);
  int32_t b1 = best / k->q_bound, b2 = best % k->q_bound;
  if ((m->n + 1) * 2 > m->cap) ml_grow(m);
  e = ml_find(m, s);
  e->key = *s;
  e->value = val;
  e->a1 = b1;
  e->a2 = b2;
  e->used = 1;
  m->n++;
  return val;
}

int sdpref_kat_multilead(const sdpref_multilead* k, double* final_value, int32_t* q1, int32_t* q2,
                         int64_t* states_visited, int64_t* cells) {
  if (!k || k->T < 1 || k->T > 16 || k->n1 < 1 || k->n1 > 16 || k->n2 < 1 || k->n2 > 16) return 1;
  mlmemo m;
  m.k = k;
  m.cap = 1 << 14;
  m.n = 0;
  m.cells = 0;
  m.tab = (mlentry*)calloc((size_t)m.cap, sizeof(mlentry));
  mst_t ini = {1, k->ini_i1, k->ini_i2, 0, 0, k->ini_cash}; /* MultiProductLeadtime.java:232 */
  double v = ml_value(&m, &ini);
  mlentry* e = ml_find(&m, &ini);
  if (final_value) *final_value = k->ini_cash + v; /* :234 */
  if (q1) *q1 = e->a1;
  if (q2) *q2 = e->a2;
  if (states_visited) *states_visited = m.n;
  if (cells) *cells = m.cells;
  dump_memo(m.tab, m.cap, m.n);
  free(m.tab);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * sdp.cash.multiItem.CashRecursionMulti.getExpectedValue (CashRecursionMulti.java:82-116) over the lambdas of
 * cash.multiItem.MultiItemCash (MultiItemCash.java:66-118): the literal memoised recursion.  Parity unpinned by
 * the reference (its main has the solve commented out and records no output).
 * ---------------------------------------------------------------------------------------- */
typedef struct mcmemo {
  const sdpref_multicash* k;
  mlentry* tab;
  int64_t cap, n, cells;
  int64_t per_period[17];
} mcmemo;

/* MultiItemCash.java:79-99 */
static double mc_imm(const sdpref_multicash* k, const mst_t* s, int32_t a1, int32_t a2, int32_t dm1, int32_t dm2) {
  double action1 = a1, action2 = a2, demand1 = dm1, demand2 = dm2;
  double endInventory1 = jmax(0, s->i1 + action1 - demand1);
  double endInventory2 = jmax(0, s->i2 + action2 - demand2);
  double revenue1 = k->price[0] * (s->i1 + action1 - endInventory1);
  double revenue2 = k->price[1] * (s->i2 + action2 - endInventory2);
  double revenue = revenue1 + revenue2;
  double orderingCost1 = k->vari_cost[0] * action1;
  double orderingCost2 = k->vari_cost[1] * action2;
  double orderingCosts = orderingCost1 + orderingCost2;
  double salValue = 0;
  if (s->period == k->T) salValue = k->sal_price[0] * endInventory1 + k->sal_price[1] * endInventory2;
  return revenue - orderingCosts + salValue;
}

/* MultiItemCash.java:103-118 (one-sided clamps and (int) casts as written there) */
static void mc_trans(const sdpref_multicash* k, const mst_t* s, int32_t a1, int32_t a2, int32_t dm1, int32_t dm2,
                     mst_t* out) {
  double endInventory1 = s->i1 + (double)a1 - (double)dm1;
  endInventory1 = jmax(0, endInventory1);
  double endInventory2 = s->i2 + (double)a2 - (double)dm2;
  endInventory2 = jmax(0, endInventory2);
  double nextCash = s->cash + mc_imm(k, s, a1, a2, dm1, dm2);
  nextCash = nextCash > k->max_cash ? k->max_cash : nextCash;
  nextCash = nextCash < k->min_cash ? k->min_cash : nextCash;
  endInventory1 = endInventory1 > k->max_inventory ? k->max_inventory : endInventory1;
  endInventory2 = endInventory2 < k->min_inventory ? k->min_inventory : endInventory2;
  nextCash = (double)jd2i(nextCash);
  endInventory1 = (double)jd2i(endInventory1);
  endInventory2 = (double)jd2i(endInventory2);
  out->period = s->period + 1;
  out->i1 = endInventory1;
  out->i2 = endInventory2;
  out->q1 = 0;
  out->q2 = 0;
  out->cash = nextCash;
}

static mlentry* mc_find(mcmemo* m, const mst_t* s) {
  mlmemo view = {NULL, m->tab, m->cap, m->n, 0};
  return ml_find(&view, s);
}

static double mc_value(mcmemo* m, const mst_t* s) {
  mlentry* e = mc_find(m, s);
  if (e->used) return e->value;
  const sdpref_multicash* k = m->k;
  const int32_t t = s->period - 1;
  double val = -DBL_MAX;
  int32_t b1 = 0, b2 = 0; /* new Actions(0, 0) */
  for (int32_t ai = 0; ai < k->q_bound; ai++)
    for (int32_t aj = 0; aj < k->q_bound; aj++) {
      if (!(k->vari_cost[0] * ai + k->vari_cost[1] * aj < s->cash + 0.1)) continue; /* buildActionList, :66-76 */
      double thisActionsValue = 0;
      for (int32_t j = k->pmf_off[t]; j < k->pmf_off[t + 1]; j++) {
        int32_t dm1 = jd2i(k->d1[j]), dm2 = jd2i(k->d2[j]); /* new Demands((int) .., (int) ..) */
        thisActionsValue += k->p[j] * mc_imm(k, s, ai, aj, dm1, dm2);
        if (s->period < k->T) {
          mst_t ns;
          mc_trans(k, s, ai, aj, dm1, dm2, &ns);
          thisActionsValue += k->p[j] * k->discount * mc_value(m, &ns);
        }
        m->cells++;
      }
      if (thisActionsValue > val + 0.1) { /* CashRecursionMulti.java:108 */
        val = thisActionsValue;
        b1 = ai;
        b2 = aj;
      }
    }
  if ((m->n + 1) * 2 > m->cap) {
    mlmemo view = {NULL, m->tab, m->cap, m->n, 0};
    ml_grow(&view);
    m->tab = view.tab;
    m->cap = view.cap;
  }
  e = mc_find(m, s);
  e->key = *s;
  e->value = val;
  e->a1 = b1;
  e->a2 = b2;
  e->used = 1;
  m->n++;
  m->per_period[s->period]++;
  return val;
}

int sdpref_multicash_memo(const sdpref_multicash* k, double* final_value, int32_t* q1, int32_t* q2,
                          int64_t* states_per_period, int64_t* cells) {
  if (!k || k->T < 1 || k->T > 16 || !k->pmf_off || !k->d1 || !k->d2 || !k->p) return 1;
  mcmemo m;
  memset(&m, 0, sizeof m);
  m.k = k;
  m.cap = 1 << 14;
  m.tab = (mlentry*)calloc((size_t)m.cap, sizeof(mlentry));
  mst_t ini = {1, k->ini_i1, k->ini_i2, 0, 0, k->ini_cash}; /* MultiItemCash.java:130 */
  double v = mc_value(&m, &ini);
  mlentry* e = mc_find(&m, &ini);
  if (final_value) *final_value = k->ini_cash + v; /* :132 */
  if (q1) *q1 = e->a1;
  if (q2) *q2 = e->a2;
  if (states_per_period)
    for (int32_t t = 0; t < k->T; t++) states_per_period[t] = m.per_period[t + 1];
  if (cells) *cells = m.cells;
  dump_memo(m.tab, m.cap, m.n);
  free(m.tab);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * sdp.cash.multiItem.CashRecursionMultiXR.getExpectedValue (CashRecursionMultiXR.java:60-96) over the lambdas of
 * cash.multiItem.MultiItemCashXR (MultiItemCashXR.java:92-148): state (x1, x2, R), R kept in mst_t.cash; actions are
 * order-up-to levels.  Parity unpinned (the reference records no output of it).
 * ---------------------------------------------------------------------------------------- */
typedef struct xrmemo {
  const sdpref_multicash* k;
  double deposit_rate;
  mlentry* tab;
  int64_t cap, n, cells;
  int64_t per_period[17];
} xrmemo;

/* MultiItemCashXR.java:108-128 */
static double xr_imm(const xrmemo* m, const mst_t* s, double action1, double action2, double demand1, double demand2) {
  const sdpref_multicash* k = m->k;
  double endInventory1 = jmax(0, action1 - demand1);
  double endInventory2 = jmax(0, action2 - demand2);
  double revenue1 = k->price[0] * (action1 - endInventory1);
  double revenue2 = k->price[1] * (action2 - endInventory2);
  double revenue = revenue1 + revenue2;
  double initialCash = s->cash - k->vari_cost[0] * s->i1 - k->vari_cost[1] * s->i2;
  double orderingCostY1 = k->vari_cost[0] * action1;
  double orderingCostY2 = k->vari_cost[1] * action2;
  double orderingCostsY = orderingCostY1 + orderingCostY2;
  double salValue = 0;
  if (s->period == k->T) salValue = k->sal_price[0] * endInventory1 + k->sal_price[1] * endInventory2;
  return revenue + (1 - m->deposit_rate) * (s->cash - orderingCostsY) + salValue - initialCash;
}

/* MultiItemCashXR.java:132-148 */
static void xr_trans(const xrmemo* m, const mst_t* s, double action1, double action2, double demand1, double demand2,
                     mst_t* out) {
  const sdpref_multicash* k = m->k;
  double endInventory1 = action1 - demand1;
  endInventory1 = jmax(0, endInventory1);
  double endInventory2 = action2 - demand2;
  endInventory2 = jmax(0, endInventory2);
  double initialCash = s->cash - k->vari_cost[0] * s->i1 - k->vari_cost[1] * s->i2;
  double nextCash = initialCash + xr_imm(m, s, action1, action2, demand1, demand2);
  nextCash = nextCash > k->max_cash ? k->max_cash : nextCash;
  nextCash = nextCash < k->min_cash ? k->min_cash : nextCash;
  endInventory1 = endInventory1 > k->max_inventory ? k->max_inventory : endInventory1;
  endInventory2 = endInventory2 < k->min_inventory ? k->min_inventory : endInventory2;
  nextCash = (double)jd2i(nextCash);
  endInventory1 = (double)jd2i(endInventory1);
  endInventory2 = (double)jd2i(endInventory2);
  double nextR = jd2i(nextCash) + k->vari_cost[0] * endInventory1 + k->vari_cost[1] * endInventory2;
  out->period = s->period + 1;
  out->i1 = endInventory1;
  out->i2 = endInventory2;
  out->q1 = 0;
  out->q2 = 0;
  out->cash = nextR;
}

static mlentry* xr_find(xrmemo* m, const mst_t* s) {
  mlmemo view = {NULL, m->tab, m->cap, m->n, 0};
  return ml_find(&view, s);
}

static double xr_value(xrmemo* m, const mst_t* s) {
  mlentry* e = xr_find(m, s);
  if (e->used) return e->value;
  const sdpref_multicash* k = m->k;
  const int32_t t = s->period - 1;
  double val = -DBL_MAX;
  int32_t b1 = 0, b2 = 0; /* bestYs = {0, 0} */
  int32_t miny1 = jd2i(s->i1), miny2 = jd2i(s->i2);
  for (int32_t yi = miny1; yi < miny1 + k->q_bound; yi++)
    for (int32_t yj = miny2; yj < miny2 + k->q_bound; yj++) { /* buildActionList, :92-105 */
      double thisActionsValue = 0;
      for (int32_t j = k->pmf_off[t]; j < k->pmf_off[t + 1]; j++) {
        thisActionsValue += k->p[j] * xr_imm(m, s, yi, yj, k->d1[j], k->d2[j]);
        if (s->period < k->T) {
          mst_t ns;
          xr_trans(m, s, yi, yj, k->d1[j], k->d2[j], &ns);
          thisActionsValue += k->p[j] * k->discount * xr_value(m, &ns);
        }
        m->cells++;
      }
      if (thisActionsValue > val + 0.1) { /* CashRecursionMultiXR.java:89 */
        val = thisActionsValue;
        b1 = yi;
        b2 = yj;
      }
    }
  if ((m->n + 1) * 2 > m->cap) {
    mlmemo view = {NULL, m->tab, m->cap, m->n, 0};
    ml_grow(&view);
    m->tab = view.tab;
    m->cap = view.cap;
  }
  e = xr_find(m, s);
  e->key = *s;
  e->value = val;
  e->a1 = b1;
  e->a2 = b2;
  e->used = 1;
  m->n++;
  m->per_period[s->period]++;
  return val;
}

/* T == 2 only: the period-2 states (no recursion below them) evaluated by several threads before the root is; the
 * arithmetic and its order per state are xr_value's, only WHO computes a state changes.  Lets the full-size
 * MultiItemCashXR.main (6.3e10 cells) be checked in about a minute. */
typedef struct xr_par_job {
  xrmemo* m;
  mst_t* states;
  double* val;
  int32_t *b1, *b2;
  int64_t lo, hi, cells;
} xr_par_job;

static void* xr_par_worker(void* arg) {
  xr_par_job* j = (xr_par_job*)arg;
  const sdpref_multicash* k = j->m->k;
  for (int64_t i = j->lo; i < j->hi; i++) {
    const mst_t* s = &j->states[i];
    const int32_t t = s->period - 1;
    double val = -DBL_MAX;
    int32_t b1 = 0, b2 = 0;
    int32_t miny1 = jd2i(s->i1), miny2 = jd2i(s->i2);
    for (int32_t yi = miny1; yi < miny1 + k->q_bound; yi++)
      for (int32_t yj = miny2; yj < miny2 + k->q_bound; yj++) {
        double thisActionsValue = 0;
        for (int32_t d = k->pmf_off[t]; d < k->pmf_off[t + 1]; d++) {
          thisActionsValue += k->p[d] * xr_imm(j->m, s, yi, yj, k->d1[d], k->d2[d]);
          j->cells++;
        }
        if (thisActionsValue > val + 0.1) {
          val = thisActionsValue;
          b1 = yi;
          b2 = yj;
        }
      }
    j->val[i] = val;
    j->b1[i] = b1;
    j->b2[i] = b2;
  }
  return NULL;
}

static int xr_prefill_last_period(xrmemo* m, const mst_t* ini, int32_t nthreads) {
  const sdpref_multicash* k = m->k;
  /* distinct successors of the root (period 2 == T) */
  xrmemo seen;
  memset(&seen, 0, sizeof seen);
  seen.k = k;
  seen.cap = 1 << 16;
  seen.tab = (mlentry*)calloc((size_t)seen.cap, sizeof(mlentry));
  int64_t n = 0, cap = 1 << 16;
  mst_t* states = (mst_t*)malloc((size_t)cap * sizeof(mst_t));
  int32_t miny1 = jd2i(ini->i1), miny2 = jd2i(ini->i2);
  for (int32_t yi = miny1; yi < miny1 + k->q_bound; yi++)
    for (int32_t yj = miny2; yj < miny2 + k->q_bound; yj++)
      for (int32_t d = k->pmf_off[0]; d < k->pmf_off[1]; d++) {
        mst_t ns;
        xr_trans(m, ini, yi, yj, k->d1[d], k->d2[d], &ns);
        mlentry* e = xr_find(&seen, &ns);
        if (e->used) continue;
        if ((seen.n + 1) * 2 > seen.cap) {
          mlmemo view = {NULL, seen.tab, seen.cap, seen.n, 0};
          ml_grow(&view);
          seen.tab = view.tab;
          seen.cap = view.cap;
          e = xr_find(&seen, &ns);
        }
        e->key = ns;
        e->used = 1;
        seen.n++;
        if (n == cap) {
          cap *= 2;
          states = (mst_t*)realloc(states, (size_t)cap * sizeof(mst_t));
        }
        states[n++] = ns;
      }
  free(seen.tab);
  double* val = (double*)malloc((size_t)n * sizeof(double));
  int32_t* b1 = (int32_t*)malloc((size_t)n * sizeof(int32_t));
  int32_t* b2 = (int32_t*)malloc((size_t)n * sizeof(int32_t));
  if (nthreads > 64) nthreads = 64;
  pthread_t th[64];
  xr_par_job jobs[64];
  for (int32_t w = 0; w < nthreads; w++) {
    jobs[w].m = m;
    jobs[w].states = states;
    jobs[w].val = val;
    jobs[w].b1 = b1;
    jobs[w].b2 = b2;
    jobs[w].lo = n * w / nthreads;
    jobs[w].hi = n * (w + 1) / nthreads;
    jobs[w].cells = 0;
    pthread_create(&th[w], NULL, xr_par_worker, &jobs[w]);
  }
  for (int32_t w = 0; w < nthreads; w++) {
    pthread_join(th[w], NULL);
    m->cells += jobs[w].cells;
  }
  for (int64_t i = 0; i < n; i++) { /* into the memo, as xr_value would have left them */
    if ((m->n + 1) * 2 > m->cap) {
      mlmemo view = {NULL, m->tab, m->cap, m->n, 0};
      ml_grow(&view);
      m->tab = view.tab;
      m->cap = view.cap;
    }
    mlentry* e = xr_find(m, &states[i]);
    e->key = states[i];
    e->value = val[i];
    e->a1 = b1[i];
    e->a2 = b2[i];
    e->used = 1;
    m->n++;
    m->per_period[states[i].period]++;
  }
  free(states);
  free(val);
  free(b1);
  free(b2);
  return 0;
}

static int32_t g_xr_threads = 1;
void sdpref_multixr_set_threads(int32_t n) { g_xr_threads = n < 1 ? 1 : n; }

int sdpref_multixr_memo(const sdpref_multicash* k, double deposit_rate, double* final_value, int32_t* y1, int32_t* y2,
                        int64_t* states_per_period, int64_t* cells) {
  if (!k || k->T < 1 || k->T > 16 || !k->pmf_off || !k->d1 || !k->d2 || !k->p) return 1;
  xrmemo m;
  memset(&m, 0, sizeof m);
  m.k = k;
  m.deposit_rate = deposit_rate;
  m.cap = 1 << 14;
  m.tab = (mlentry*)calloc((size_t)m.cap, sizeof(mlentry));
  mst_t ini = {1, k->ini_i1, k->ini_i2, 0, 0, k->ini_cash}; /* MultiItemCashXR.java:158: iniCash handed over as R */
  if (k->T == 2 && g_xr_threads > 1) xr_prefill_last_period(&m, &ini, g_xr_threads);
  double v = xr_value(&m, &ini);
  mlentry* e = xr_find(&m, &ini);
  if (final_value) *final_value = k->ini_cash + v; /* :160 */
  if (y1) *y1 = e->a1;
  if (y2) *y2 = e->a2;
  if (states_per_period)
    for (int32_t t = 0; t < k->T; t++) states_per_period[t] = m.per_period[t + 1];
  if (cells) *cells = m.cells;
  dump_memo(m.tab, m.cap, m.n);
  free(m.tab);
  return 0;
}

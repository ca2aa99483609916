/*
 * staffref.c -- CPU ORACLE for workforce.StaffRecursion.  TEST INFRASTRUCTURE ONLY (see sdpref.h: nothing in the
 * product may import, link, call or execute anything under oracle/).
 *
 * What it restates: src/workforce/StaffRecursion.java:81-118 -- getExpectedValue(StaffState), the recursion the
 * workforce drivers solve (WorkforcePlanning.java:104-112, WorkforceTesting.java:116-124): the pmf of a period
 * depends on the hire-up-to level y = iniStaffNum + orderQty (pmfs[t][min(y, pmfs[t].length - 1)], :92-95); the
 * accumulation interleaves p * imm and p * V(next) per realisation (:101-106); strict `<` scan from
 * Double.MAX_VALUE with bestHireQty = 0 (:88-90, :110-113).  Lambdas: WorkforcePlanning.java:84-101 (clamped
 * transition) and WorkforceTesting.java:91-107 (unclamped).  Parity pin status: PARITY UNPINNED by the reference
 * (it records no outputs for this class and cannot be run here); protected by dense-sweep == literal memoised
 * recursion (staffref_memo below), an independent pure-Python restatement (tests/pyref.py) and a hand-computed case.
 *
 * Compile with -ffp-contract=off (oracle/Makefile).
 */
#include "staffref.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define JAVA_DOUBLE_MAX 1.7976931348623157e308

static int32_t row_of(const staffref_problem* p, int32_t y) { return y >= p->n_rows - 1 ? p->n_rows - 1 : y; }
static int32_t len_of(const staffref_problem* p, int32_t row) { return p->row_len ? p->row_len[row] : row + 1; }
static const double* probs_of(const staffref_problem* p, int32_t period, int32_t row) {
  return p->prob + ((size_t)(period - 1) * (size_t)p->n_rows + (size_t)row) * (size_t)p->row_stride;
}

/* WorkforcePlanning.java:92-101 == WorkforceTesting.java:97-107 */
static double immediate_value(const staffref_problem* p, int32_t period, int32_t iniStaffNum, int32_t action,
                              int32_t randomDemand) {
  double fixHireCost = action > 0 ? p->fix_cost : 0;
  double variHireCost = p->unit_vari_cost * action;
  int32_t nextStaffNum = iniStaffNum + action - randomDemand;
  double salaryCost = p->salary * nextStaffNum;
  int32_t t = period - 1;
  double penaltyCost = nextStaffNum > p->min_staff[t] ? 0 : p->unit_penalty * (p->min_staff[t] - nextStaffNum);
  double totalCosts = fixHireCost + variHireCost + salaryCost + penaltyCost;
  return totalCosts;
}

/* WorkforcePlanning.java:84-89 (clamp: upper then lower) / WorkforceTesting.java:91-94 (no clamp) */
static int32_t state_transition(const staffref_problem* p, int32_t iniStaffNum, int32_t action, int32_t randomDemand) {
  int32_t nextStaffNum = iniStaffNum + action - randomDemand;
  if (p->clamp) {
    nextStaffNum = nextStaffNum > p->max_x ? p->max_x : nextStaffNum;
    nextStaffNum = nextStaffNum < p->min_x ? p->min_x : nextStaffNum;
  }
  return nextStaffNum;
}

typedef double (*vlook_fn)(void* env, int32_t period, int32_t x);

/* StaffRecursion.java:83-116 for one state */
static void eval_state(const staffref_problem* p, int32_t period, int32_t iniStaffNum, vlook_fn vlook, void* env,
                       double* val_out, int32_t* act_out, int64_t* cells) {
  int32_t bestHireQty = 0;
  double val = JAVA_DOUBLE_MAX;
  for (int32_t orderQty = 0; orderQty <= p->max_hire; ++orderQty) { /* getFeasibleAction: 0, 1, ..., maxHireNum */
    int32_t hireUpTo = row_of(p, iniStaffNum + orderQty);
    const double* pmf = probs_of(p, period, hireUpTo);
    int32_t n = len_of(p, hireUpTo);
    double thisQValue = 0;
    for (int32_t j = 0; j < n; ++j) {
      int32_t demand = j;
      double thisValue = immediate_value(p, period, iniStaffNum, orderQty, demand);
      thisQValue += pmf[j] * thisValue;
      if (period < p->T) {
        int32_t next = state_transition(p, iniStaffNum, orderQty, demand);
        thisQValue += pmf[j] * vlook(env, period + 1, next);
      }
    }
    *cells += n;
    if (thisQValue < val) {
      val = thisQValue;
      bestHireQty = orderQty;
    }
  }
  *val_out = val;
  *act_out = bestHireQty;
}

int staffref_layout(const staffref_problem* p, int32_t* x_lo, int32_t* nx) {
  if (!p || p->T < 1 || p->n_rows < 1 || p->max_hire < 0) return 1;
  int32_t lo = p->clamp ? p->min_x : p->ini_x, hi = p->clamp ? p->max_x : p->ini_x;
  if (lo < 0 || hi < lo) return 1;
  int32_t dmax = 0;
  for (int32_t r = 0; r < p->n_rows; ++r) {
    int32_t n = len_of(p, r);
    if (n < 1 || n > r + 1 || n > p->row_stride) return 1; /* a realisation never exceeds the staff it hits */
    if (n - 1 > dmax) dmax = n - 1;
  }
  for (int32_t t = 0; t < p->T; ++t) {
    x_lo[t] = lo;
    nx[t] = hi - lo + 1;
    if (!p->clamp) {
      lo = lo - dmax < 0 ? 0 : lo - dmax;
      hi = hi + p->max_hire;
    }
  }
  return 0;
}

typedef struct dense_env {
  const double* v_next;
  int32_t next_lo;
} dense_env;

static double dense_look(void* env, int32_t period, int32_t x) {
  const dense_env* e = (const dense_env*)env;
  (void)period;
  return e->v_next[x - e->next_lo];
}

typedef struct work {
  const staffref_problem* p;
  int32_t period, x_lo;
  dense_env env;
  double* v_cur;
  int32_t* pol;
  int64_t lo, hi, cells;
} work;

static void* worker(void* arg) {
  work* w = (work*)arg;
  for (int64_t i = w->lo; i < w->hi; ++i)
    eval_state(w->p, w->period, w->x_lo + (int32_t)i, dense_look, &w->env, &w->v_cur[i], &w->pol[i], &w->cells);
  return NULL;
}

int staffref_period(const staffref_problem* p, int32_t period, const double* v_next, double* v_cur, int32_t* pol,
                    int64_t lo, int64_t hi, int32_t nthreads, int64_t* cells_out) {
  int32_t x_lo[4096], nx[4096];
  if (p->T > 4096 || staffref_layout(p, x_lo, nx)) return 1;
  if (period < 1 || period > p->T || lo < 0 || hi > nx[period - 1]) return 1;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  work w[64];
  pthread_t th[64];
  int64_t n = hi - lo, cells = 0;
  for (int32_t k = 0; k < nthreads; ++k) {
    w[k].p = p;
    w[k].period = period;
    w[k].x_lo = x_lo[period - 1];
    w[k].env.v_next = v_next;
    w[k].env.next_lo = period < p->T ? x_lo[period] : 0;
    w[k].v_cur = v_cur;
    w[k].pol = pol;
    w[k].lo = lo + n * k / nthreads;
    w[k].hi = lo + n * (k + 1) / nthreads;
    w[k].cells = 0;
  }
  if (nthreads == 1) {
    worker(&w[0]);
  } else {
    for (int32_t k = 0; k < nthreads; ++k) pthread_create(&th[k], NULL, worker, &w[k]);
    for (int32_t k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
  }
  for (int32_t k = 0; k < nthreads; ++k) cells += w[k].cells;
  if (cells_out) *cells_out += cells;
  return 0;
}

int staffref_solve(const staffref_problem* p, double* values, int32_t* policy, const int64_t* off, int32_t nthreads,
                   int64_t* cells_out) {
  int32_t x_lo[4096], nx[4096];
  if (p->T > 4096 || staffref_layout(p, x_lo, nx)) return 1;
  if (cells_out) *cells_out = 0;
  for (int32_t period = p->T; period >= 1; --period) {
    const double* v_next = period < p->T ? values + off[period] : NULL;
    int rc = staffref_period(p, period, v_next, values + off[period - 1], policy + off[period - 1], 0, nx[period - 1],
                             nthreads, cells_out);
    if (rc) return rc;
  }
  return 0;
}

/* ---- the literal recursion: cacheValues.computeIfAbsent(state, ...) from the initial state ------------------- */
typedef struct memo_env {
  const staffref_problem* p;
  int32_t width; /* staff numbers 0 .. width-1 */
  double* val;
  int32_t* act;
  uint8_t* seen;
  int64_t cells;
  int bad;
} memo_env;

static double memo_look(void* env, int32_t period, int32_t x) {
  memo_env* m = (memo_env*)env;
  if (x < 0 || x >= m->width) {
    m->bad = 1;
    return 0;
  }
  size_t at = (size_t)(period - 1) * (size_t)m->width + (size_t)x;
  if (!m->seen[at]) {
    double v;
    int32_t a;
    eval_state(m->p, period, x, memo_look, m, &v, &a, &m->cells);
    m->val[at] = v;
    m->act[at] = a;
    m->seen[at] = 1;
  }
  return m->val[at];
}

int staffref_memo(const staffref_problem* p, double* root_value, int32_t* root_action, int32_t width, double* val,
                  int32_t* act, uint8_t* seen, int64_t* cells_out) {
  if (!p || p->T < 1 || width < 1 || p->ini_x < 0 || p->ini_x >= width) return 1;
  memo_env m = {p, width, val, act, seen, 0, 0};
  memset(seen, 0, (size_t)p->T * (size_t)width);
  *root_value = memo_look(&m, 1, p->ini_x);
  *root_action = act[p->ini_x];
  if (cells_out) *cells_out = m.cells;
  return m.bad ? 2 : 0;
}

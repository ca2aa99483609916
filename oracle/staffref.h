/*
 * staffref.h -- CPU ORACLE for workforce.StaffRecursion (StaffRecursion.java:81-118).  TEST INFRASTRUCTURE ONLY:
 * same rules as sdpref.h.  Parity unpinned by the reference (no recorded outputs, no JDK here); see staffref.c.
 */
#ifndef STAFFREF_H
#define STAFFREF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct staffref_problem {
  int32_t T;
  int32_t min_x, max_x; /* minX / maxX of the clamped transition (WorkforcePlanning.java:84-89) */
  int32_t clamp;        /* 0: WorkforceTesting.java:91-94 */
  int32_t ini_x;        /* iniStaffNum of the period-1 state */
  int32_t max_hire;     /* maxHireNum: actions 0 .. max_hire */
  double fix_cost, unit_vari_cost, salary, unit_penalty;
  const int32_t* min_staff; /* minStaffNum[t], T entries */
  int32_t n_rows;           /* pmfs[t].length */
  int32_t row_stride;
  const double* prob;     /* prob[((t * n_rows) + y) * row_stride + j] = pmfs[t][y][j][1]; pmfs[t][y][j][0] == j */
  const int32_t* row_len; /* pmfs[t][y].length, n_rows entries; NULL: y + 1 (WorkforcePlanning.java:57-68) */
} staffref_problem;

/* per-period staff-number boxes (x_lo[t], nx[t]), t = 0..T-1 */
int staffref_layout(const staffref_problem* p, int32_t* x_lo, int32_t* nx);

/* states [lo, hi) of one period against a dense v_next (NULL for period T); *cells_out is incremented */
int staffref_period(const staffref_problem* p, int32_t period, const double* v_next, double* v_cur, int32_t* pol,
                    int64_t lo, int64_t hi, int32_t nthreads, int64_t* cells_out);

/* dense backward sweep; off[t] = offset of period t+1 inside values / policy */
int staffref_solve(const staffref_problem* p, double* values, int32_t* policy, const int64_t* off, int32_t nthreads,
                   int64_t* cells_out);

/* the literal memoised recursion from (1, ini_x); val/act/seen are [T][width] (staff numbers 0..width-1) */
int staffref_memo(const staffref_problem* p, double* root_value, int32_t* root_action, int32_t width, double* val,
                  int32_t* act, uint8_t* seen, int64_t* cells_out);

#ifdef __cplusplus
}
#endif
#endif

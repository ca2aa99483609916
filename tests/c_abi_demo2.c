/* c_abi_demo2.c -- plain-C callers of the two entry-point groups added after the single-item families:
 *   staff      sdpgpu_create (family STAFF) + sdpgpu_set_level_pmf + sdpgpu_set_overhead: workforce.StaffRecursion
 *   multicash  sdpgpu_multicash_solve with a memo read-out (sdpgpu_multi_set_table): CashRecursionMulti
 * The input file is written by tests/test_gpu_c_abi.py; the numbers printed are checked there against the oracles. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdpgpu.h"

static int staff(FILE* f) {
  int T, rows, max_hire, max_x;
  if (fscanf(f, "%d %d %d %d", &T, &rows, &max_hire, &max_x) != 4) return 1;
  sdpgpu_desc d;
  sdpgpu_desc_init(&d);
  d.family = SDPGPU_FAMILY_STAFF;
  d.direction = SDPGPU_MIN;
  d.periods = T;
  d.min_inventory = 0;  /* minX */
  d.max_inventory = max_x;
  d.max_order_quantity = max_hire;
  d.fixed_order_cost = 100; /* fixCost,      WorkforcePlanning.java:37 */
  d.unit_order_cost = 10;   /* unitVariCost, :38 */
  d.holding_cost = 20;      /* salary,       :39 */
  d.penalty_cost = 80;      /* unitPenalty,  :40 */
  d.ini_inventory = 0;
  sdpgpu_handle* h = NULL;
  if (sdpgpu_create(&d, &h) != 0) {
    fprintf(stderr, "create: %s\n", sdpgpu_last_error(NULL));
    return 2;
  }
  double* prob = calloc((size_t)rows * rows, sizeof(double));
  for (int t = 0; t < T; t++) {
    for (int y = 0; y < rows; y++)
      for (int j = 0; j <= y; j++)
        if (fscanf(f, "%lf", &prob[(size_t)y * rows + j]) != 1) return 1;
    if (sdpgpu_set_level_pmf(h, t, prob, NULL, rows, rows) != 0 || sdpgpu_set_overhead(h, t, 8.0) != 0) {
      fprintf(stderr, "pmf: %s\n", sdpgpu_last_error(h));
      return 2;
    }
  }
  if (sdpgpu_solve(h, 1) != 0) {
    fprintf(stderr, "solve: %s\n", sdpgpu_last_error(h));
    return 2;
  }
  int64_t n = sdpgpu_num_states(h, 1);
  double* v = malloc(sizeof(double) * (size_t)n);
  int32_t* pol = malloc(sizeof(int32_t) * (size_t)n);
  if (sdpgpu_values(h, 1, v, n) != 0 || sdpgpu_policy(h, 1, pol, 0, n) != 0) return 2;
  int64_t i0 = sdpgpu_state_index(h, 1, 0.0, 0.0, 0.0);
  printf("final optimal expected cost is: %.17g\n", v[i0]);
  printf("optimal hiring number in the first priod is : %d\n", pol[i0]);
  sdpgpu_destroy(h);
  free(prob);
  free(v);
  free(pol);
  return 0;
}

static int multicash(FILE* f) {
  sdpgpu_multicash k;
  memset(&k, 0, sizeof k);
  int n_all = 0;
  int32_t off[17];
  if (fscanf(f, "%d %d", &k.T, &k.q_bound) != 2 || k.T > 16) return 1;
  if (fscanf(f, "%lf %lf %lf %lf %lf %lf", &k.price[0], &k.price[1], &k.vari_cost[0], &k.vari_cost[1], &k.sal_price[0],
             &k.sal_price[1]) != 6)
    return 1;
  if (fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %lf", &k.ini_cash, &k.ini_i1, &k.ini_i2, &k.min_inventory, &k.max_inventory,
             &k.min_cash, &k.max_cash, &k.discount) != 8)
    return 1;
  off[0] = 0;
  double *d1 = NULL, *d2 = NULL, *p = NULL;
  for (int t = 0; t < k.T; t++) {
    int n;
    if (fscanf(f, "%d", &n) != 1) return 1;
    d1 = realloc(d1, sizeof(double) * (size_t)(n_all + n));
    d2 = realloc(d2, sizeof(double) * (size_t)(n_all + n));
    p = realloc(p, sizeof(double) * (size_t)(n_all + n));
    for (int j = 0; j < n; j++)
      if (fscanf(f, "%lf %lf %lf", &d1[n_all + j], &d2[n_all + j], &p[n_all + j]) != 3) return 1;
    n_all += n;
    off[t + 1] = n_all;
  }
  k.pmf_off = off;
  k.d1 = d1;
  k.d2 = d2;
  k.p = p;
  double final_value, ms;
  int32_t q1, q2;
  int64_t states[16], cells;
  if (sdpgpu_multicash_solve(&k, &final_value, &q1, &q2, states, &cells, &ms) != 0) {
    fprintf(stderr, "multicash: %s\n", sdpgpu_multilead_last_error());
    return 2;
  }
  int64_t rows = 0;
  for (int t = 0; t < k.T; t++) rows += states[t];
  sdpgpu_multi_table tab;
  memset(&tab, 0, sizeof tab);
  tab.capacity = rows;
  tab.period = malloc(sizeof(int32_t) * (size_t)rows);
  tab.a1 = malloc(sizeof(int32_t) * (size_t)rows);
  tab.a2 = malloc(sizeof(int32_t) * (size_t)rows);
  tab.i1 = malloc(sizeof(double) * (size_t)rows);
  tab.i2 = malloc(sizeof(double) * (size_t)rows);
  tab.q1 = malloc(sizeof(double) * (size_t)rows);
  tab.q2 = malloc(sizeof(double) * (size_t)rows);
  tab.cash = malloc(sizeof(double) * (size_t)rows);
  tab.value = malloc(sizeof(double) * (size_t)rows);
  sdpgpu_multi_set_table(&tab);
  int rc = sdpgpu_multicash_solve(&k, &final_value, &q1, &q2, states, &cells, &ms);
  sdpgpu_multi_set_table(NULL);
  if (rc != 0 || tab.rows != rows) return 2;
  double vsum = 0;
  for (int64_t i = 0; i < rows; i++) vsum += tab.value[i];
  printf("final optimal cash  is %.17g\n", final_value);
  printf("optimal order quantity in the first priod is :  Q1 = %d, Q2 = %d\n", q1, q2);
  printf("visited states %lld, cells %lld, sum of their values %.17g\n", (long long)rows, (long long)cells, vsum);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* f = fopen(argv[2], "r");
  if (!f) return 1;
  int rc = strcmp(argv[1], "staff") == 0 ? staff(f) : multicash(f);
  fclose(f);
  return rc;
}

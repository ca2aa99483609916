/* c_abi_demo.c -- a plain-C caller of include/sdpgpu.h shaped like capacitated.CLSP.main
 * (CLSP.java:196-286): build the problem, solve, print "final optimal expected value" and the
 * first-period order quantity.  Compiled with gcc and linked against libsdpgpu.so by
 * tests/test_gpu_c_abi.py; it shows that nothing but C types crosses the boundary. */
#include <stdio.h>
#include <stdlib.h>

#include "sdpgpu.h"

#define CHECK(h, call)                                                        \
  do {                                                                        \
    int rc_ = (call);                                                         \
    if (rc_ != 0) {                                                           \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sdpgpu_last_error(h));    \
      return 2;                                                               \
    }                                                                         \
  } while (0)

int main(int argc, char** argv) {
  /* pmf file: T, then per period n followed by n (demand, prob) pairs */
  if (argc < 2) return 1;
  FILE* f = fopen(argv[1], "r");
  if (!f) return 1;
  int T;
  if (fscanf(f, "%d", &T) != 1) return 1;

  sdpgpu_desc d;
  sdpgpu_desc_init(&d);
  d.family = SDPGPU_FAMILY_BACKORDER;
  d.direction = SDPGPU_MIN;
  d.periods = T;
  d.step = 1;                 /* stepSize,             CLSP.java:202 */
  d.min_inventory = -300;     /* minState,             :203 */
  d.max_inventory = 300;      /* maxState,             :204 */
  d.fixed_order_cost = 500;   /* fixedOrderingCost,    :207 */
  d.unit_order_cost = 0;      /* proportionalOrderingCost */
  d.holding_cost = 2;
  d.penalty_cost = 10;
  d.max_order_quantity = 60;  /* maxOrderQuantity,     :211 */
  d.ini_inventory = 1;        /* initialInventory,     :197 */

  sdpgpu_handle* h = NULL;
  CHECK(NULL, sdpgpu_create(&d, &h));
  for (int t = 0; t < T; ++t) {
    int n;
    if (fscanf(f, "%d", &n) != 1) return 1;
    double* dem = malloc(sizeof(double) * n);
    double* pr = malloc(sizeof(double) * n);
    for (int j = 0; j < n; ++j)
      if (fscanf(f, "%lf %lf", &dem[j], &pr[j]) != 2) return 1;
    CHECK(h, sdpgpu_set_pmf(h, t, dem, pr, n));
    free(dem);
    free(pr);
  }
  fclose(f);

  CHECK(h, sdpgpu_solve(h, 1));
  int64_t idx = sdpgpu_state_index(h, 1, d.ini_inventory, 0, 0);
  int64_t S = sdpgpu_num_states(h, 1);
  double* v = malloc(sizeof(double) * S);
  int32_t* pol = malloc(sizeof(int32_t) * S);
  CHECK(h, sdpgpu_values(h, 1, v, S));
  CHECK(h, sdpgpu_policy(h, 1, pol, 0, S));
  printf("planning horizon is %d periods\n", T);
  printf("final optimal expected value is: %.17g\n", v[idx]);
  printf("optimal order quantity in the first priod is : %.17g\n", pol[idx] * d.step);
  sdpgpu_stats st;
  CHECK(h, sdpgpu_stats_get(h, &st));
  printf("cells %lld\n", (long long)st.cells_evaluated);
  free(v);
  free(pol);
  sdpgpu_destroy(h);
  return 0;
}

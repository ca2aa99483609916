/* c_abi_sharded.c -- a plain-C caller of the multi-GPU half of include/sdpgpu.h.  It solves the SAME problem
 * (CLSP.main's shape on a wider grid, CLSP.java:196-286) four ways and checks that every table is bit-identical to
 * the single-handle sdpgpu_solve:
 *   1. one rank of one, through an RCCL communicator created with sdpgpu_comm_unique_id + sdpgpu_comm_init and the
 *      per-period all-gathers of sdpgpu_solve_sharded (blocking, then overlapped), V_1 gathered too: the collective
 *      path with nobody to talk to -- what one rank of eight executes;
 *   2. sdpgpu_solve_multi with ONE handle (ncclCommInitAll over one device, grouped all-gather);
 *   3. sdpgpu_solve_multi with `nranks` handles that share this device (slabs exchanged by device copies);
 *   4. sdpgpu_run_period + sdpgpu_exchange stepped by the caller.
 * Usage: c_abi_sharded <states> <actions> <demands> <periods> <nranks>; prints "ok <cells>" or a diagnosis. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdpgpu.h"

#define CHECK(h, call)                                                     \
  do {                                                                     \
    int rc_ = (call);                                                      \
    if (rc_ != 0) {                                                        \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sdpgpu_last_error(h)); \
      return 2;                                                            \
    }                                                                      \
  } while (0)

static int S, A, D, T;

static int make(sdpgpu_handle** out, int rank, int world) {
  sdpgpu_desc d;
  sdpgpu_desc_init(&d);
  d.family = SDPGPU_FAMILY_BACKORDER;
  d.direction = SDPGPU_MIN;
  d.periods = T;
  d.min_inventory = -(S / 2);
  d.max_inventory = S - S / 2 - 1;
  d.fixed_order_cost = 500;
  d.unit_order_cost = 1;
  d.holding_cost = 2;
  d.penalty_cost = 10;
  d.max_order_quantity = A - 1;
  d.rank = rank;
  d.world_size = world;
  d.device = 0;
  CHECK(NULL, sdpgpu_create(&d, out));
  double* dem = malloc(sizeof(double) * D);
  double* pr = malloc(sizeof(double) * D);
  for (int t = 0; t < T; ++t) {
    /* triangular pmf whose peak moves with the period: dyadic-free, every period different */
    double tot = 0;
    for (int j = 0; j < D; ++j) {
      dem[j] = j;
      pr[j] = 1.0 + fmin((double)j, (double)((D - 1 - j) + t % 3));
      tot += pr[j];
    }
    for (int j = 0; j < D; ++j) pr[j] /= tot;
    CHECK(*out, sdpgpu_set_pmf(*out, t, dem, pr, D));
  }
  free(dem);
  free(pr);
  return 0;
}

/* whole V_t of every period + the policy slab [lo, hi) of `h`, against the single-handle tables */
static int same_tables(sdpgpu_handle* h, double** v_ref, int32_t** p_ref, int first_full, const char* what) {
  double* v = malloc(sizeof(double) * S);
  int32_t* p = malloc(sizeof(int32_t) * S);
  for (int t = 1; t <= T; ++t) {
    int64_t pad, lo, hi;
    CHECK(h, sdpgpu_slab(h, t, &pad, &lo, &hi));
    CHECK(h, sdpgpu_values(h, t, v, S));
    CHECK(h, sdpgpu_policy(h, t, p, lo, hi - lo));
    int64_t a = t >= first_full ? 0 : lo, b = t >= first_full ? S : hi;
    if (memcmp(v + a, v_ref[t] + a, sizeof(double) * (size_t)(b - a)) != 0) {
      fprintf(stderr, "%s: V_%d differs from the single-handle solve\n", what, t);
      return 3;
    }
    if (memcmp(p, p_ref[t] + lo, sizeof(int32_t) * (size_t)(hi - lo)) != 0) {
      fprintf(stderr, "%s: policy of period %d differs\n", what, t);
      return 3;
    }
  }
  free(v);
  free(p);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 6) return 1;
  S = atoi(argv[1]), A = atoi(argv[2]), D = atoi(argv[3]), T = atoi(argv[4]);
  int nranks = atoi(argv[5]);
  int rc;

  sdpgpu_handle* one = NULL;
  if ((rc = make(&one, 0, 1))) return rc;
  CHECK(one, sdpgpu_solve(one, 1));
  double** v_ref = malloc(sizeof(double*) * (T + 1));
  int32_t** p_ref = malloc(sizeof(int32_t*) * (T + 1));
  for (int t = 1; t <= T; ++t) {
    v_ref[t] = malloc(sizeof(double) * S);
    p_ref[t] = malloc(sizeof(int32_t) * S);
    CHECK(one, sdpgpu_values(one, t, v_ref[t], S));
    CHECK(one, sdpgpu_policy(one, t, p_ref[t], 0, S));
  }
  sdpgpu_stats st;
  CHECK(one, sdpgpu_stats_get(one, &st));

  /* 1. RCCL communicator of one rank, blocking then overlapped */
  unsigned char id[SDPGPU_UNIQUE_ID_BYTES];
  CHECK(NULL, sdpgpu_comm_unique_id(id));
  sdpgpu_handle* r0 = NULL;
  if ((rc = make(&r0, 0, 1))) return rc;
  CHECK(r0, sdpgpu_comm_init(r0, id, 0, 1));
  CHECK(r0, sdpgpu_solve_sharded(r0, SDPGPU_SHARDED_SYNC | SDPGPU_SHARDED_GATHER_FIRST));
  if ((rc = same_tables(r0, v_ref, p_ref, 1, "solve_sharded (blocking)"))) return rc;
  CHECK(r0, sdpgpu_solve_sharded(r0, SDPGPU_SHARDED_SYNC | SDPGPU_SHARDED_OVERLAP | SDPGPU_SHARDED_GATHER_FIRST));
  if ((rc = same_tables(r0, v_ref, p_ref, 1, "solve_sharded (overlapped)"))) return rc;
  /* 4. stepped by the caller */
  for (int t = T; t >= 1; --t) {
    CHECK(r0, sdpgpu_run_period(r0, t));
    CHECK(r0, sdpgpu_exchange(r0, t));
  }
  CHECK(r0, sdpgpu_synchronize(r0));
  if ((rc = same_tables(r0, v_ref, p_ref, 1, "run_period + exchange"))) return rc;
  CHECK(r0, sdpgpu_comm_destroy(r0));
  sdpgpu_destroy(r0);

  /* 2. one process, one device: ncclCommInitAll */
  sdpgpu_handle* m1 = NULL;
  if ((rc = make(&m1, 0, 1))) return rc;
  CHECK(m1, sdpgpu_solve_multi(&m1, 1, SDPGPU_SHARDED_SYNC | SDPGPU_SHARDED_GATHER_FIRST));
  if ((rc = same_tables(m1, v_ref, p_ref, 1, "solve_multi (1 handle, RCCL)"))) return rc;
  sdpgpu_destroy(m1);

  /* 3. nranks handles on this device: slabs exchanged by copies; run twice (communicator state is kept) */
  sdpgpu_handle** hs = malloc(sizeof(sdpgpu_handle*) * nranks);
  for (int r = 0; r < nranks; ++r)
    if ((rc = make(&hs[r], r, nranks))) return rc;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hs[0], sdpgpu_solve_multi(hs, nranks, SDPGPU_SHARDED_SYNC));
    long long cells = 0;
    for (int r = 0; r < nranks; ++r) {
      if ((rc = same_tables(hs[r], v_ref, p_ref, 2, "solve_multi (shared device)"))) return rc;
      sdpgpu_stats sr;
      CHECK(hs[r], sdpgpu_stats_get(hs[r], &sr));
      cells += sr.cells_evaluated;
    }
    if (cells != st.cells_evaluated) {
      fprintf(stderr, "cells of the slabs %lld != %lld\n", cells, (long long)st.cells_evaluated);
      return 3;
    }
  }
  for (int r = 0; r < nranks; ++r) sdpgpu_destroy(hs[r]);
  sdpgpu_destroy(one);
  printf("ok %lld\n", (long long)st.cells_evaluated);
  return 0;
}

/*
 * sdpgpu_jni.c -- JNI shim between sdp.gpu.SdpGpu and the C ABI of include/sdpgpu.h.
 * SOURCE ONLY: jni.h does not exist in the authoring image; build recipe in SdpGpu.java.
 * Rules: never retain a Java reference, never call back into the JVM from a HIP callback,
 * turn every non-zero status into IllegalStateException(sdpgpu_last_error()).
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sdpgpu.h"

static void throw_state(JNIEnv* env, const char* msg) {
  jclass c = (*env)->FindClass(env, "java/lang/IllegalStateException");
  if (c) (*env)->ThrowNew(env, c, msg);
}
#define H(h) ((sdpgpu_handle*)(intptr_t)(h))
/* after a throw no further JNI call may be made with the exception pending: CHECK is always the last JNI-touching
 * statement of a void function; functions that go on to allocate use CHECK_RET */
#define CHECK(env, h, rc) do { if ((rc) != 0) throw_state(env, sdpgpu_last_error(H(h))); } while (0)
#define CHECK_RET(env, h, rc, ret) do { if ((rc) != 0) { throw_state(env, sdpgpu_last_error(H(h))); return ret; } } while (0)

static int fill_desc(JNIEnv* env, jintArray ints, jdoubleArray dbls, sdpgpu_desc* out);

/* sdpgpu_create_custom: the driver's lambdas as HIP device text + the constants they close over */
JNIEXPORT jlong JNICALL Java_sdp_gpu_SdpGpu_createCustom(JNIEnv* env, jclass cls, jintArray ints, jdoubleArray dbls,
                                                         jstring source, jdoubleArray params) {
  sdpgpu_desc d;
  if (!source || fill_desc(env, ints, dbls, &d)) return 0;
  const char* text = (*env)->GetStringUTFChars(env, source, NULL);
  jsize n = params ? (*env)->GetArrayLength(env, params) : 0;
  if (!text) return 0; /* OutOfMemoryError pending */
  jdouble* p = n ? (*env)->GetDoubleArrayElements(env, params, NULL) : NULL;
  if (n && !p) {
    (*env)->ReleaseStringUTFChars(env, source, text);
    return 0;
  }
  sdpgpu_handle* h = NULL;
  int rc = sdpgpu_create_custom(&d, text, p, n, &h);
  if (p) (*env)->ReleaseDoubleArrayElements(env, params, p, JNI_ABORT);
  (*env)->ReleaseStringUTFChars(env, source, text);
  if (rc != 0) {
    throw_state(env, sdpgpu_last_error(NULL));
    return 0;
  }
  return (jlong)(intptr_t)h;
}

JNIEXPORT jlong JNICALL Java_sdp_gpu_SdpGpu_create(JNIEnv* env, jclass cls, jintArray ints, jdoubleArray dbls) {
  sdpgpu_desc d;
  if (fill_desc(env, ints, dbls, &d)) return 0;
  sdpgpu_handle* h = NULL;
  if (sdpgpu_create(&d, &h) != 0) {
    throw_state(env, sdpgpu_last_error(NULL));
    return 0;
  }
  return (jlong)(intptr_t)h;
}

/* 0 = ok; 1 = an exception is pending (arrays too short) */
static int fill_desc(JNIEnv* env, jintArray ints, jdoubleArray dbls, sdpgpu_desc* out) {
  sdpgpu_desc d;
  sdpgpu_desc_init(&d);
  jint i[12];
  jdouble v[28];
  if (!ints || !dbls || (*env)->GetArrayLength(env, ints) < 12 || (*env)->GetArrayLength(env, dbls) < 28) {
    throw_state(env, "descriptor arrays must hold 12 ints and 28 doubles");
    return 1;
  }
  (*env)->GetIntArrayRegion(env, ints, 0, 12, i);
  (*env)->GetDoubleArrayRegion(env, dbls, 0, 28, v);
  d.family = i[0]; d.direction = i[1]; d.periods = i[2]; d.clamp_inventory = i[3];
  d.zero_order_last_period = i[4]; d.cash_round_int_div = i[5]; d.cash_formula = i[6]; d.kernel = i[7];
  d.device = i[8]; d.rank = i[9]; d.world_size = i[10]; d.store_all_values = i[11];
  d.step = v[0]; d.min_inventory = v[1]; d.max_inventory = v[2]; d.max_order_quantity = v[3];
  d.ini_inventory = v[4]; d.ini_cash = v[5]; d.ini_preq = v[6];
  d.fixed_order_cost = v[7]; d.unit_order_cost = v[8]; d.holding_cost = v[9]; d.penalty_cost = v[10];
  d.price = v[11]; d.salvage_value = v[12]; d.deposit_rate = v[13]; d.overhead_cost = v[14];
  d.overhead_rate = v[15]; d.discount_factor = v[16]; d.min_cash = v[17]; d.max_cash = v[18];
  d.cash_round_mult = v[19]; d.cash_round_div = v[20];
  d.r0 = v[21]; d.r2 = v[22]; d.r3 = v[23]; d.overdraft_limit = v[24]; d.interest_free_amount = v[25];
  *out = d;
  return 0;
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_destroy(JNIEnv* env, jclass cls, jlong h) { sdpgpu_destroy(H(h)); }

JNIEXPORT jstring JNICALL Java_sdp_gpu_SdpGpu_buildId(JNIEnv* env, jclass cls) { return (*env)->NewStringUTF(env, sdpgpu_build_id()); }

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_setPmf(JNIEnv* env, jclass cls, jlong h, jint t, jdoubleArray dem, jdoubleArray prob) {
  if (!dem || !prob) {
    throw_state(env, "setPmf: null array");
    return;
  }
  jsize n = (*env)->GetArrayLength(env, dem);
  if ((*env)->GetArrayLength(env, prob) < n) { /* pmf[t][j] = {demand, prob}: one probability per demand */
    throw_state(env, "setPmf: fewer probabilities than demands");
    return;
  }
  jdouble* d = (*env)->GetPrimitiveArrayCritical(env, dem, NULL);
  if (!d) return; /* OutOfMemoryError pending */
  jdouble* p = (*env)->GetPrimitiveArrayCritical(env, prob, NULL);
  if (!p) {
    (*env)->ReleasePrimitiveArrayCritical(env, dem, d, JNI_ABORT);
    return;
  }
  int rc = sdpgpu_set_pmf(H(h), t, d, p, n); /* copies; no JNI call between Get and Release */
  (*env)->ReleasePrimitiveArrayCritical(env, prob, p, JNI_ABORT);
  (*env)->ReleasePrimitiveArrayCritical(env, dem, d, JNI_ABORT);
  CHECK(env, h, rc);
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_setLevelPmf(JNIEnv* env, jclass cls, jlong h, jint t, jdoubleArray prob,
                                                        jintArray rowLen, jint nRows, jint rowStride) {
  if ((jlong)(*env)->GetArrayLength(env, prob) < (jlong)nRows * rowStride || (*env)->GetArrayLength(env, rowLen) < nRows) {
    throw_state(env, "setLevelPmf: arrays shorter than nRows * rowStride / nRows");
    return;
  }
  jdouble* p = (*env)->GetPrimitiveArrayCritical(env, prob, NULL);
  if (!p) return;
  jint* len = (*env)->GetPrimitiveArrayCritical(env, rowLen, NULL);
  if (!len) {
    (*env)->ReleasePrimitiveArrayCritical(env, prob, p, JNI_ABORT);
    return;
  }
  int rc = sdpgpu_set_level_pmf(H(h), t, p, (const int32_t*)len, nRows, rowStride); /* copies */
  (*env)->ReleasePrimitiveArrayCritical(env, rowLen, len, JNI_ABORT);
  (*env)->ReleasePrimitiveArrayCritical(env, prob, p, JNI_ABORT);
  CHECK(env, h, rc);
}

/* sdpgpu_getpmf: sizing call, then the tile of period t as rows {demand, probability} */
JNIEXPORT jobjectArray JNICALL Java_sdp_gpu_SdpGpu_getPmf(JNIEnv* env, jclass cls, jintArray kinds, jdoubleArray a, jdoubleArray b,
                                                         jdouble q, jdouble step, jint variant, jint t) {
  jsize T = kinds ? (*env)->GetArrayLength(env, kinds) : 0;
  if (T < 1 || T > 4096 || !a || !b || (*env)->GetArrayLength(env, a) < T || (*env)->GetArrayLength(env, b) < T) {
    throw_state(env, "getPmf: kinds / a / b must have one entry per period");
    return NULL;
  }
  sdpgpu_dist_spec* spec = malloc(sizeof(sdpgpu_dist_spec) * (size_t)T);
  jint* ki = malloc(sizeof(jint) * (size_t)T);
  jdouble* av = malloc(sizeof(jdouble) * (size_t)T);
  jdouble* bv = malloc(sizeof(jdouble) * (size_t)T);
  double *dem = NULL, *pr = NULL;
  jobjectArray out = NULL;
  if (!spec || !ki || !av || !bv) goto oom;
  (*env)->GetIntArrayRegion(env, kinds, 0, T, ki);
  (*env)->GetDoubleArrayRegion(env, a, 0, T, av);
  (*env)->GetDoubleArrayRegion(env, b, 0, T, bv);
  for (jsize i = 0; i < T; i++) {
    spec[i].kind = ki[i];
    spec[i].reserved = 0;
    spec[i].a = av[i];
    spec[i].b = bv[i];
  }
  int32_t n = 0;
  if (sdpgpu_getpmf(spec, T, q, step, variant, t, NULL, NULL, 0, &n) != 0) goto fail;
  dem = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  pr = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  if (!dem || !pr) goto oom;
  if (sdpgpu_getpmf(spec, T, q, step, variant, t, dem, pr, n, &n) != 0) goto fail;
  {
    jclass rowCls = (*env)->FindClass(env, "[D");
    out = rowCls ? (*env)->NewObjectArray(env, n, rowCls, NULL) : NULL;
    for (int32_t j = 0; j < n && out; j++) {
      jdouble r[2] = {dem[j], pr[j]};
      jdoubleArray row = (*env)->NewDoubleArray(env, 2);
      if (!row) {
        out = NULL;
        break;
      }
      (*env)->SetDoubleArrayRegion(env, row, 0, 2, r);
      (*env)->SetObjectArrayElement(env, out, j, row);
      (*env)->DeleteLocalRef(env, row);
    }
  }
  goto done;
fail:
  throw_state(env, sdpgpu_last_error(NULL));
  goto done;
oom : {
  jclass c = (*env)->FindClass(env, "java/lang/OutOfMemoryError");
  if (c) (*env)->ThrowNew(env, c, "getPmf");
}
done:
  free(spec); free(ki); free(av); free(bv); free(dem); free(pr);
  return out;
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_setOverhead(JNIEnv* env, jclass cls, jlong h, jint t, jdouble oh) {
  CHECK(env, h, sdpgpu_set_overhead(H(h), t, oh));
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_solve(JNIEnv* env, jclass cls, jlong h) { CHECK(env, h, sdpgpu_solve(H(h), 1)); }

/* ---- multi-GPU (include/sdpgpu.h, "multi-GPU" section) -------------------------------------------------------
 * A JVM owns every GPU of the node: one handle per device (desc.rank = r, desc.world_size = n, desc.device = r) and ONE
 * call, the way `getExpectedValue(initialState)` is one call (Recursion.java:89).  RCCL communicators are created
 * inside the library (ncclCommInitAll) at the first call. */
JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_solveMulti(JNIEnv* env, jclass cls, jlongArray handles, jboolean gatherFirst) {
  jsize n = handles ? (*env)->GetArrayLength(env, handles) : 0;
  if (n < 1 || n > 64) {
    throw_state(env, "solveMulti: 1..64 handles");
    return;
  }
  jlong hv[64];
  sdpgpu_handle* hs[64];
  (*env)->GetLongArrayRegion(env, handles, 0, n, hv);
  for (jsize r = 0; r < n; r++) hs[r] = H(hv[r]);
  int rc = sdpgpu_solve_multi(hs, n, SDPGPU_SHARDED_SYNC | (gatherFirst ? SDPGPU_SHARDED_GATHER_FIRST : 0));
  CHECK(env, hv[0], rc);
}

/* One JVM (or thread) per GPU: rank 0 draws the id, the host ships its 128 bytes to the other ranks by any channel. */
JNIEXPORT jbyteArray JNICALL Java_sdp_gpu_SdpGpu_commUniqueId(JNIEnv* env, jclass cls) {
  jbyte id[SDPGPU_UNIQUE_ID_BYTES];
  if (sdpgpu_comm_unique_id(id) != 0) {
    throw_state(env, sdpgpu_last_error(NULL));
    return NULL;
  }
  jbyteArray out = (*env)->NewByteArray(env, SDPGPU_UNIQUE_ID_BYTES);
  if (out) (*env)->SetByteArrayRegion(env, out, 0, SDPGPU_UNIQUE_ID_BYTES, id);
  return out;
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_commInit(JNIEnv* env, jclass cls, jlong h, jbyteArray id, jint rank, jint world) {
  jbyte buf[SDPGPU_UNIQUE_ID_BYTES];
  if (!id || (*env)->GetArrayLength(env, id) != SDPGPU_UNIQUE_ID_BYTES) {
    throw_state(env, "commInit: the unique id is 128 bytes");
    return;
  }
  (*env)->GetByteArrayRegion(env, id, 0, SDPGPU_UNIQUE_ID_BYTES, buf);
  CHECK(env, h, sdpgpu_comm_init(H(h), buf, rank, world));
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_solveSharded(JNIEnv* env, jclass cls, jlong h, jboolean overlap, jboolean gatherFirst) {
  CHECK(env, h, sdpgpu_solve_sharded(H(h), SDPGPU_SHARDED_SYNC | (overlap ? SDPGPU_SHARDED_OVERLAP : 0) |
                                               (gatherFirst ? SDPGPU_SHARDED_GATHER_FIRST : 0)));
}

/* this rank's slab [lo, hi) of period t and the padded row length: {padded, lo, hi} */
JNIEXPORT jlongArray JNICALL Java_sdp_gpu_SdpGpu_slab(JNIEnv* env, jclass cls, jlong h, jint period) {
  int64_t pad, lo, hi;
  int rc = sdpgpu_slab(H(h), period, &pad, &lo, &hi);
  CHECK_RET(env, h, rc, NULL);
  jlong v[3] = {pad, lo, hi};
  jlongArray out = (*env)->NewLongArray(env, 3);
  if (out) (*env)->SetLongArrayRegion(env, out, 0, 3, v);
  return out;
}

/* policy of this rank's slab: out.length = hi - lo */
JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_policySlab(JNIEnv* env, jclass cls, jlong h, jint period, jlong lo, jintArray out) {
  if (!out) {
    throw_state(env, "policySlab: null array");
    return;
  }
  jsize n = (*env)->GetArrayLength(env, out);
  jint* o = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
  if (!o) return;
  int rc = sdpgpu_policy(H(h), period, (int32_t*)o, lo, n);
  (*env)->ReleasePrimitiveArrayCritical(env, out, o, 0);
  CHECK(env, h, rc);
}

JNIEXPORT jlong JNICALL Java_sdp_gpu_SdpGpu_numStates(JNIEnv* env, jclass cls, jlong h, jint period) {
  return sdpgpu_num_states(H(h), period);
}

JNIEXPORT jdoubleArray JNICALL Java_sdp_gpu_SdpGpu_grid(JNIEnv* env, jclass cls, jlong h, jint period) {
  double x_lo; int64_t nx, nc, nq;
  int rc = sdpgpu_grid(H(h), period, &x_lo, &nx, &nc, &nq);
  CHECK_RET(env, h, rc, NULL); /* no NewDoubleArray with an exception pending */
  jdouble v[4] = {x_lo, (double)nx, (double)nc, (double)nq};
  jdoubleArray out = (*env)->NewDoubleArray(env, 4);
  if (out) (*env)->SetDoubleArrayRegion(env, out, 0, 4, v);
  return out;
}

JNIEXPORT jdouble JNICALL Java_sdp_gpu_SdpGpu_cashValue(JNIEnv* env, jclass cls, jlong h, jlong ic) { return sdpgpu_cash_value(H(h), ic); }

JNIEXPORT jlong JNICALL Java_sdp_gpu_SdpGpu_stateIndex(JNIEnv* env, jclass cls, jlong h, jint period, jdouble x, jdouble cash, jdouble preq) {
  return sdpgpu_state_index(H(h), period, x, cash, preq);
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_values(JNIEnv* env, jclass cls, jlong h, jint period, jdoubleArray out) {
  jsize n = (*env)->GetArrayLength(env, out);
  jdouble* o = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
  if (!o) return;
  int rc = sdpgpu_values(H(h), period, o, n);
  (*env)->ReleasePrimitiveArrayCritical(env, out, o, 0);
  CHECK(env, h, rc);
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_policy(JNIEnv* env, jclass cls, jlong h, jint period, jintArray out) {
  jsize n = (*env)->GetArrayLength(env, out);
  jint* o = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
  if (!o) return;
  int rc = sdpgpu_policy(H(h), period, (int32_t*)o, 0, n);
  (*env)->ReleasePrimitiveArrayCritical(env, out, o, 0);
  CHECK(env, h, rc);
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_evalStates(JNIEnv* env, jclass cls, jlong h, jint period, jdoubleArray x, jdoubleArray cash,
                                                      jdoubleArray preq, jdoubleArray outv, jintArray outa) {
  if (!x || !outv || !outa) {
    throw_state(env, "evalStates: null array");
    return;
  }
  jsize n = (*env)->GetArrayLength(env, x);
  if ((cash && (*env)->GetArrayLength(env, cash) < n) || (preq && (*env)->GetArrayLength(env, preq) < n) ||
      (*env)->GetArrayLength(env, outv) < n || (*env)->GetArrayLength(env, outa) < n) {
    throw_state(env, "evalStates: arrays shorter than x");
    return;
  }
  jdouble* px = (*env)->GetDoubleArrayElements(env, x, NULL);
  jdouble* pc = cash ? (*env)->GetDoubleArrayElements(env, cash, NULL) : NULL;
  jdouble* pq = preq ? (*env)->GetDoubleArrayElements(env, preq, NULL) : NULL;
  jdouble* ov = (*env)->GetDoubleArrayElements(env, outv, NULL);
  jint* oa = (*env)->GetIntArrayElements(env, outa, NULL);
  int rc = 0;
  const int got_all = px && ov && oa && (!cash || pc) && (!preq || pq); /* else OutOfMemoryError is pending */
  if (got_all) rc = sdpgpu_eval_states(H(h), period, n, px, pc, pq, ov, (int32_t*)oa);
  if (oa) (*env)->ReleaseIntArrayElements(env, outa, oa, 0);
  if (ov) (*env)->ReleaseDoubleArrayElements(env, outv, ov, 0);
  if (pq) (*env)->ReleaseDoubleArrayElements(env, preq, pq, JNI_ABORT);
  if (pc) (*env)->ReleaseDoubleArrayElements(env, cash, pc, JNI_ABORT);
  if (px) (*env)->ReleaseDoubleArrayElements(env, x, px, JNI_ABORT);
  if (!got_all) return;
  CHECK(env, h, rc);
}

JNIEXPORT void JNICALL Java_sdp_gpu_SdpGpu_reachable(JNIEnv* env, jclass cls, jlong h, jint period, jbyteArray out) {
  jsize n = (*env)->GetArrayLength(env, out);
  jbyte* o = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
  if (!o) return;
  int rc = sdpgpu_reachable(H(h), period, (uint8_t*)o, n);
  (*env)->ReleasePrimitiveArrayCritical(env, out, o, 0);
  CHECK(env, h, rc);
}


/* sdpgpu_multicash_solve / sdpgpu_multixr_solve + sdpgpu_multi_set_table: two calls, the first sizes the table */
JNIEXPORT jobjectArray JNICALL Java_sdp_gpu_SdpGpu_multiSolve(JNIEnv* env, jclass cls, jint model, jint T, jint qBound,
                                                              jdoubleArray scalars, jintArray pmfOff, jdoubleArray d1,
                                                              jdoubleArray d2, jdoubleArray p) {
  if ((*env)->GetArrayLength(env, scalars) < 15 || (*env)->GetArrayLength(env, pmfOff) < T + 1 || T < 1 || T > 16) {
    throw_state(env, "multiSolve: bad arguments");
    return NULL;
  }
  jdouble v[15];
  (*env)->GetDoubleArrayRegion(env, scalars, 0, 15, v);
  jint* off = (*env)->GetIntArrayElements(env, pmfOff, NULL);
  jdouble* a1 = (*env)->GetDoubleArrayElements(env, d1, NULL);
  jdouble* a2 = (*env)->GetDoubleArrayElements(env, d2, NULL);
  jdouble* ap = (*env)->GetDoubleArrayElements(env, p, NULL);
  if (!off || !a1 || !a2 || !ap) { /* OutOfMemoryError pending: release what was pinned and leave */
    if (ap) (*env)->ReleaseDoubleArrayElements(env, p, ap, JNI_ABORT);
    if (a2) (*env)->ReleaseDoubleArrayElements(env, d2, a2, JNI_ABORT);
    if (a1) (*env)->ReleaseDoubleArrayElements(env, d1, a1, JNI_ABORT);
    if (off) (*env)->ReleaseIntArrayElements(env, pmfOff, off, JNI_ABORT);
    return NULL;
  }
  {
    const jsize need = off[T];
    if (need < 0 || (*env)->GetArrayLength(env, d1) < need || (*env)->GetArrayLength(env, d2) < need ||
        (*env)->GetArrayLength(env, p) < need) {
      (*env)->ReleaseDoubleArrayElements(env, p, ap, JNI_ABORT);
      (*env)->ReleaseDoubleArrayElements(env, d2, a2, JNI_ABORT);
      (*env)->ReleaseDoubleArrayElements(env, d1, a1, JNI_ABORT);
      (*env)->ReleaseIntArrayElements(env, pmfOff, off, JNI_ABORT);
      throw_state(env, "multiSolve: demand arrays shorter than pmfOff[T]");
      return NULL;
    }
  }
  sdpgpu_multicash k;
  memset(&k, 0, sizeof k);
  k.T = T; k.q_bound = qBound;
  k.price[0] = v[0]; k.price[1] = v[1]; k.vari_cost[0] = v[2]; k.vari_cost[1] = v[3]; k.sal_price[0] = v[4]; k.sal_price[1] = v[5];
  k.ini_cash = v[6]; k.ini_i1 = v[7]; k.ini_i2 = v[8]; k.min_inventory = v[9]; k.max_inventory = v[10];
  k.min_cash = v[11]; k.max_cash = v[12]; k.discount = v[13];
  k.pmf_off = (const int32_t*)off; k.d1 = a1; k.d2 = a2; k.p = ap;
  double fv, ms;
  int32_t q1, q2;
  int64_t states[16], cells, rows = 0;
  sdpgpu_multi_table tab;
  memset(&tab, 0, sizeof tab);
  jobjectArray out = NULL;
  int oom = 0;
  int rc = model == 2 ? sdpgpu_multixr_solve(&k, v[14], &fv, &q1, &q2, states, &cells, &ms)
                      : sdpgpu_multicash_solve(&k, &fv, &q1, &q2, states, &cells, &ms);
  if (rc == 0) {
    for (int t = 0; t < T; t++) rows += states[t];
    tab.capacity = rows;
    tab.period = malloc(sizeof(int32_t) * (size_t)rows); tab.a1 = malloc(sizeof(int32_t) * (size_t)rows);
    tab.a2 = malloc(sizeof(int32_t) * (size_t)rows); tab.i1 = malloc(sizeof(double) * (size_t)rows);
    tab.i2 = malloc(sizeof(double) * (size_t)rows); tab.q1 = malloc(sizeof(double) * (size_t)rows);
    tab.q2 = malloc(sizeof(double) * (size_t)rows); tab.cash = malloc(sizeof(double) * (size_t)rows);
    tab.value = malloc(sizeof(double) * (size_t)rows);
    if (!tab.period || !tab.a1 || !tab.a2 || !tab.i1 || !tab.i2 || !tab.q1 || !tab.q2 || !tab.cash || !tab.value) {
      oom = 1;
    } else {
      sdpgpu_multi_set_table(&tab);
      rc = model == 2 ? sdpgpu_multixr_solve(&k, v[14], &fv, &q1, &q2, states, &cells, &ms)
                      : sdpgpu_multicash_solve(&k, &fv, &q1, &q2, states, &cells, &ms);
      sdpgpu_multi_set_table(NULL);
    }
  }
  (*env)->ReleaseDoubleArrayElements(env, p, ap, JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, d2, a2, JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, d1, a1, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, pmfOff, off, JNI_ABORT);
  if (oom) {
    jclass c = (*env)->FindClass(env, "java/lang/OutOfMemoryError");
    if (c) (*env)->ThrowNew(env, c, "multiSolve: memo table");
  } else if (rc != 0) {
    throw_state(env, sdpgpu_multilead_last_error());
  } else {
    jclass rowCls = (*env)->FindClass(env, "[D");
    out = rowCls ? (*env)->NewObjectArray(env, (jsize)rows, rowCls, NULL) : NULL;
    for (int64_t i = 0; i < rows && out; i++) {
      jdouble r[7] = {tab.period[i], tab.i1[i], tab.i2[i], tab.cash[i], tab.value[i], tab.a1[i], tab.a2[i]};
      jdoubleArray row = (*env)->NewDoubleArray(env, 7);
      if (!row) { /* OutOfMemoryError pending */
        out = NULL;
        break;
      }
      (*env)->SetDoubleArrayRegion(env, row, 0, 7, r);
      (*env)->SetObjectArrayElement(env, out, (jsize)i, row);
      (*env)->DeleteLocalRef(env, row);
    }
  }
  free(tab.period); free(tab.a1); free(tab.a2); free(tab.i1); free(tab.i2); free(tab.q1); free(tab.q2); free(tab.cash); free(tab.value);
  return out;
}

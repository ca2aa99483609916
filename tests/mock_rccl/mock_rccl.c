/*
 * mock_rccl.c -- TEST DOUBLE for librccl.so.1 (never shipped, never linked by the product): the RCCL entry points
 * libsdpgpu.so resolves with dlsym (csrc/sdpgpu_comm.hip), implemented over POSIX shared memory and blocking HIP copies, so
 * that the multi-PROCESS path of the library -- sdpgpu_comm_prepare / sdpgpu_comm_init with world > 1 in separate processes,
 * sdpgpu_solve_sharded, bench.py --exchange native -- can run with several ranks on ONE GPU (RCCL itself refuses two ranks on
 * one device).  What it does not stand in for is RCCL's transport between devices.
 *
 * Loaded through SDPGPU_RCCL_LIB.  An all-gather: wait for the stream, copy this rank's slab to its slot of the shared
 * segment, barrier, copy the other ranks' slabs into place, barrier.  Synchronous where RCCL is asynchronous -- the
 * ordering the library relies on (the collective behind the kernel that produced the slab, ahead of whatever is enqueued
 * next on that stream) is kept.
 *
 * Build: hipcc -shared -fPIC -o libmock_rccl.so mock_rccl.c   (tests/test_gpu_native_mock_rccl.py does it)
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclDataType_t;

typedef struct {
  _Atomic int arrived;
  _Atomic int generation;
  _Atomic int attached;
  _Atomic int aborted;  /* ncclCommAbort on any rank: everybody waiting in a barrier of this communicator leaves it with an error */
  int world;
  size_t slot_bytes;
} header_t;

typedef struct mock_comm {
  int rank, world;
  char name[64];  /* "" = a process-local segment (ncclCommInitAll) */
  header_t* hd;
  char* slots;
  size_t map_bytes;
} mock_comm;

/* ncclGroupStart / ncclGroupEnd: ONE thread issuing the all-gathers of several ranks (sdpgpu_solve_multi's default) cannot
 * block in the first of them; the calls of a group are queued and carried out together at the closing ncclGroupEnd. */
typedef struct { const void* send; void* recv; size_t bytes; mock_comm* c; hipStream_t stream; } queued_t;
static __thread int g_depth = 0;
static __thread int g_n = 0;
static __thread queued_t g_q[64];

static size_t slot_bytes_env(void) {
  const char* e = getenv("MOCK_RCCL_SLOT_MB");
  size_t mb = e ? (size_t)atol(e) : 64;
  return mb << 20;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof id->internal, "/sdpmock_%d_%ld", (int)getpid(), (long)(ts.tv_nsec ^ ts.tv_sec));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
  mock_comm* c = (mock_comm*)calloc(1, sizeof *c);
  c->rank = rank;
  c->world = world;
  memcpy(c->name, id.internal, sizeof c->name - 1);
  const size_t slot = slot_bytes_env();
  c->map_bytes = 4096 + slot * (size_t)world;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;  /* (every rank sets the same size) */
  void* p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->hd = (header_t*)p;
  c->slots = (char*)p + 4096;
  c->hd->world = world;
  c->hd->slot_bytes = slot;
  atomic_fetch_add(&c->hd->attached, 1);
  /* collective: wait until every rank has attached (RCCL's ncclCommInitRank blocks the same way) */
  for (long spins = 0; atomic_load(&c->hd->attached) < world; ++spins) {
    sched_yield();
    if (spins > 2000000000L) return ncclInternalError;
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(void** comms, int n, const int* devs) {
  (void)devs;  /* one process owning every rank: the segment is ordinary memory of this process */
  if (!comms || n < 1 || n > 64) return ncclInvalidArgument;
  const size_t slot = slot_bytes_env();
  const size_t bytes = 4096 + slot * (size_t)n;
  char* p = (char*)calloc(1, bytes);
  if (!p) return ncclSystemError;
  header_t* hd = (header_t*)p;
  hd->world = n;
  hd->slot_bytes = slot;
  atomic_store(&hd->attached, n);
  for (int r = 0; r < n; ++r) {
    mock_comm* c = (mock_comm*)calloc(1, sizeof *c);
    c->rank = r;
    c->world = n;
    c->hd = hd;
    c->slots = p + 4096;
    c->map_bytes = bytes;
    comms[r] = c;
  }
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(void* comm) {
  mock_comm* c = (mock_comm*)comm;
  if (!c) return ncclSuccess;
  const int last = atomic_fetch_sub(&c->hd->attached, 1) == 1;
  if (c->name[0]) {
    if (last) shm_unlink(c->name);
    munmap((void*)c->hd, c->map_bytes);
  } else if (last) {
    free((void*)c->hd);
  }
  free(c);
  return ncclSuccess;
}

/* 0 = everybody arrived; 1 = the communicator was aborted while waiting */
static int barrier(mock_comm* c) {
  const int gen = atomic_load(&c->hd->generation);
  if (atomic_load(&c->hd->aborted)) return 1;
  if (atomic_fetch_add(&c->hd->arrived, 1) == c->world - 1) {
    atomic_store(&c->hd->arrived, 0);
    atomic_fetch_add(&c->hd->generation, 1);
  } else {
    while (atomic_load(&c->hd->generation) == gen) {
      if (atomic_load(&c->hd->aborted)) return 1;
      sched_yield();
    }
  }
  return 0;
}

/* releases every rank blocked in a collective of this communicator (they return an error) and frees the handle, as
 * ncclCommAbort does; the shared header stays until its last holder is gone */
ncclResult_t ncclCommAbort(void* comm) {
  mock_comm* c = (mock_comm*)comm;
  if (!c) return ncclSuccess;
  atomic_store(&c->hd->aborted, 1);
  return ncclSuccess;  /* (the handle itself is leaked on purpose: a peer thread may still be inside a call on it) */
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, void* comm, hipStream_t stream) {
  (void)type;  /* the library sends 8-byte elements (ncclUint64) */
  mock_comm* c = (mock_comm*)comm;
  const size_t bytes = count * 8;
  if (bytes > c->hd->slot_bytes) {
    fprintf(stderr, "[mock_rccl] slab of %zu bytes exceeds the slot (MOCK_RCCL_SLOT_MB)\n", bytes);
    return ncclInvalidArgument;
  }
  if (g_depth > 0) {  /* inside a group: carried out at ncclGroupEnd */
    if (g_n >= 64) return ncclInternalError;
    g_q[g_n++] = (queued_t){send, recv, bytes, c, stream};
    return ncclSuccess;
  }
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slots + (size_t)c->rank * c->hd->slot_bytes, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (barrier(c)) return ncclSystemError;
  for (int r = 0; r < c->world; ++r) {
    char* dst = (char*)recv + (size_t)r * bytes;
    if (r == c->rank) {
      if (dst != (const char*)send && hipMemcpy(dst, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
      continue;
    }
    if (hipMemcpy(dst, c->slots + (size_t)r * c->hd->slot_bytes, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  if (barrier(c)) return ncclSystemError;  /* nobody overwrites its slot before everybody has read it */
  return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) {
  ++g_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd(void) {
  if (g_depth <= 0) return ncclInvalidArgument;
  if (--g_depth > 0) return ncclSuccess;
  /* every rank of the communicator is in the group (sdpgpu_solve_multi): all slabs out, then all slabs in */
  ncclResult_t rc = ncclSuccess;
  for (int i = 0; i < g_n && rc == ncclSuccess; ++i) {
    const queued_t* q = &g_q[i];
    if (hipStreamSynchronize(q->stream) != hipSuccess ||
        hipMemcpy(q->c->slots + (size_t)q->c->rank * q->c->hd->slot_bytes, q->send, q->bytes, hipMemcpyDeviceToHost) != hipSuccess)
      rc = ncclUnhandledCudaError;
  }
  for (int i = 0; i < g_n && rc == ncclSuccess; ++i) {
    const queued_t* q = &g_q[i];
    for (int r = 0; r < q->c->world; ++r) {
      char* dst = (char*)q->recv + (size_t)r * q->bytes;
      if (r == q->c->rank) {
        if (dst != (const char*)q->send && hipMemcpy(dst, q->send, q->bytes, hipMemcpyDeviceToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
      } else if (hipMemcpy(dst, q->c->slots + (size_t)r * q->c->hd->slot_bytes, q->bytes, hipMemcpyHostToDevice) != hipSuccess) {
        rc = ncclUnhandledCudaError;
      }
    }
  }
  g_n = 0;
  return rc;
}
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error (mock_rccl)";
    case ncclUnhandledCudaError: return "HIP error inside mock_rccl";
    case ncclSystemError: return "shared memory error inside mock_rccl";
    case ncclInvalidArgument: return "invalid argument (mock_rccl)";
    default: return "internal error (mock_rccl)";
  }
}

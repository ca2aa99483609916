// sdpgpu_comm.hip -- the multi-GPU half of the C ABI: the state axis of a period is cut into world_size contiguous
// slabs (layout(), sdpgpu.hip), each rank computes its slab reading the FULL V_{t+1}, and ONE all-gather per period
// rebuilds V_t on every rank, in place in the row the next period reads.  The reference's
// contract is one call that solves everything (Recursion.java:89 `getExpectedValue(initialState)`): here that call is
// sdpgpu_solve_sharded (one rank per process / thread) or sdpgpu_solve_multi (one process that owns all devices).
//
// RCCL is not linked: librccl.so.1 is opened on first use, so a single-GPU caller never loads it.  xGMI is
// point-to-point (7 links per GPU); the message is S/world * 8 B per rank per period (1 MB at the 1e6-state grid,
// 100 MB at 1e8 states) against milliseconds to seconds of compute, so one ring/direct all-gather per period on the
// compute stream is the whole protocol; the overlapped schedule (second stream, interior tiles first) exists for the
// families with a bounded footprint and small slabs.
#include "sdpgpu_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <mutex>
#include <thread>

using namespace sdpgpu_detail;

namespace {

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: the failure path of the thread-per-rank sweep
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

// nullptr + message when the library or one of its symbols is missing
RcclApi* rccl() {
  std::call_once(g_rccl_once, [] {
    static_assert(SDPGPU_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    const char* names[] = {"librccl.so.1", "librccl.so"};
    // SDPGPU_RCCL_LIB: the collective library to open instead (a site's own RCCL build; tests/mock_rccl.c -- a stand-in that
    // moves the slabs through host shared memory, so that the multi-PROCESS path can run with several ranks on one GPU)
    if (const char* path = std::getenv("SDPGPU_RCCL_LIB")) {
      g_rccl.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
      if (!g_rccl.lib) {
        const char* e = dlerror();
        g_rccl.err = std::string("cannot load SDPGPU_RCCL_LIB=") + path + ": " + (e ? e : "?");
        return;
      }
    }
    for (const char* n : names) {
      if (g_rccl.lib) break;
      g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // a copy the process already holds (e.g. PyTorch's)
    }
    for (int i = 0; !g_rccl.lib && i < 2; ++i) g_rccl.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.lib) {
      const char* e = dlerror();
      g_rccl.err = std::string("cannot load librccl.so.1: ") + (e ? e : "?");
      return;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
      void* p = dlsym(g_rccl.lib, name);
      if (!p) {
        ok = false;
        g_rccl.err = std::string("librccl lacks ") + name;
      }
      return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(g_rccl.lib, "ncclCommAbort");  // (absent: peers are not released early)
    if (!ok) g_rccl.lib = nullptr;
  });
  return g_rccl.lib ? &g_rccl : nullptr;
}

#define NCCL_TRY(h, api, expr)                                                                             \
  do {                                                                                                     \
    ncclResult_t r_ = (expr);                                                                              \
    if (r_ != ncclSuccess) return fail(h, SDPGPU_ERR_DEVICE, "%s: %s", #expr, (api)->GetErrorString(r_)); \
  } while (0)

// the row of `period` that travels, as sdpgpu_exchange_ptr reports it: 8-byte elements, S_pad of them
void* exchange_row(sdpgpu_handle* h, int period) {
  if (h->pending_chunks[period - 1] > 0) return h->d_keys + (size_t)(period - 1) * h->key_stride;
  return h->d_values + h->per[period - 1].v_off;
}

int ensure_comm_stream(sdpgpu_handle* h) {
  if (h->comm_stream) return SDPGPU_OK;
  HIP_TRY(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  HIP_TRY(h, hipEventCreateWithFlags(&h->ev_comp, hipEventDisableTiming));
  HIP_TRY(h, hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
  return SDPGPU_OK;
}

// all-gather of this rank's slab of the row of `period`, in place, on stream `st`
int enqueue_allgather(sdpgpu_handle* h, RcclApi* api, int period, hipStream_t st) {
  const PeriodInfo& p = h->per[period - 1];
  const size_t n = (size_t)(p.S_pad / h->d.world_size);
  if (n == 0) return SDPGPU_OK;
  char* row = (char*)exchange_row(h, period);
  // bytes travel untouched (an all-gather does no arithmetic): the keys and the fp64 values are both sent as u64
  NCCL_TRY(h, api, api->AllGather(row + (size_t)h->d.rank * n * 8, row, n, ncclUint64, (ncclComm_t)h->comm, st));
  return SDPGPU_OK;
}

// One rank's sweep.  `exchange(period, overlapped)` enqueues the collective for the row of `period`.
template <class Exchange>
int sweep_rank(sdpgpu_handle* h, int flags, Exchange&& exchange) {
  int rc = allocate(h);
  if (rc) return rc;
  rc = ensure_device(h);
  if (rc) return rc;
  const bool overlap = (flags & SDPGPU_SHARDED_OVERLAP) != 0;
  const int last = (flags & SDPGPU_SHARDED_GATHER_FIRST) ? 1 : 2;  // lowest period whose row is exchanged
  if (overlap) {
    rc = ensure_comm_stream(h);
    if (rc) return rc;
  }
  std::fill(h->period_done.begin(), h->period_done.end(), 0);
  HIP_TRY(h, hipEventRecord(h->ev_solve0, h->stream));
  bool in_flight = false;  // the exchange of period + 1 is running on the second stream
  for (int period = h->T; period >= 1; --period) {
    if (in_flight) {
      // interior tiles read only this rank's slab of V_{period+1}: they run beside the all-gather
      rc = run_period_impl(h, period, SDPGPU_PART_INTERIOR);
      if (rc) return rc;
      HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_comm, 0));
      rc = run_period_impl(h, period, SDPGPU_PART_BOUNDARY);
      in_flight = false;
    } else {
      rc = run_period_impl(h, period);
    }
    if (rc) return rc;
    count_cells(h, period);
    if (period >= last) {
      if (overlap && period > 1) {
        HIP_TRY(h, hipEventRecord(h->ev_comp, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->comm_stream, h->ev_comp, 0));
        rc = exchange(period, true);
        if (rc) return rc;
        HIP_TRY(h, hipEventRecord(h->ev_comm, h->comm_stream));
        in_flight = true;
      } else {
        rc = exchange(period, false);
        if (rc) return rc;
      }
    }
  }
  if (in_flight) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_comm, 0));
  rc = flush_api(h);  // deferred read-out: every V_t row (decoded over the whole row) and this rank's policy rows
  if (rc) return rc;
  HIP_TRY(h, hipEventRecord(h->ev_solve1, h->stream));
  h->solve_timed = true;
  if (flags & SDPGPU_SHARDED_SYNC) HIP_TRY(h, hipStreamSynchronize(h->stream));
  return SDPGPU_OK;
}

}  // namespace

namespace sdpgpu_detail {

void comm_release(sdpgpu_handle* h) {
  if (h->comm) {
    if (RcclApi* api = rccl()) (void)api->CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
  }
  if (h->ev_comp) (void)hipEventDestroy(h->ev_comp);
  if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  h->ev_comp = h->ev_comm = nullptr;
  h->comm_stream = nullptr;
  // the siblings of a multi-handle solve must not touch this handle any more
  for (sdpgpu_handle* s : h->siblings)
    if (s && s != h) s->siblings.clear();
  h->siblings.clear();
}

}  // namespace sdpgpu_detail

extern "C" {

int sdpgpu_comm_unique_id(void* out_id) {
  g_create_error.clear();
  if (!out_id) return fail(nullptr, SDPGPU_ERR_ARG, "comm_unique_id: null buffer");
  RcclApi* api = rccl();
  if (!api) return fail(nullptr, SDPGPU_ERR_DEVICE, "%s", g_rccl.err.c_str());
  ncclUniqueId id;
  NCCL_TRY(nullptr, api, api->GetUniqueId(&id));
  std::memcpy(out_id, &id, sizeof id);
  return SDPGPU_OK;
}

// Everything of sdpgpu_comm_init that can fail on ONE rank alone -- RCCL does not load, the device tables do not fit, no
// device -- without entering a collective: a rank that fails here has not left its peers waiting inside
// ncclCommInitRank.  Idempotent.
int sdpgpu_comm_prepare(sdpgpu_handle* h) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  try {
    RcclApi* api = rccl();  // (a load check: no ncclGetUniqueId, which would start a bootstrap thread on every rank)
    if (!api) return fail(h, SDPGPU_ERR_DEVICE, "%s", g_rccl.err.c_str());
    int rc = allocate(h);  // needs the device: no CPU path
    if (rc) return rc;
    rc = ensure_device(h);
    if (rc) return rc;
    if (h->device < 0) HIP_TRY(h, hipGetDevice(&h->device));  // the communicator is bound to a device: pin the handle to it
    h->comm_prepared = true;
    return SDPGPU_OK;
  } catch (const std::exception& e) {
    return fail(h, SDPGPU_ERR_ARG, "exception: %s", e.what());
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "unknown exception");
  }
}

int sdpgpu_comm_init(sdpgpu_handle* h, const void* unique_id, int32_t rank, int32_t world) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (!unique_id) return fail(h, SDPGPU_ERR_ARG, "comm_init: null unique id");
  if (rank != h->d.rank || world != h->d.world_size)
    return fail(h, SDPGPU_ERR_ARG, "comm_init: rank %d of %d, but the handle was created as rank %d of %d", rank, world, h->d.rank, h->d.world_size);
  if (h->comm) return fail(h, SDPGPU_ERR_STATE, "comm_init: the handle already has a communicator");
  try {
    if (!h->comm_prepared) {  // (a caller that skipped sdpgpu_comm_prepare: same work, but its failure is seen only here)
      int rc = sdpgpu_comm_prepare(h);
      if (rc) return rc;
    }
    RcclApi* api = rccl();
    if (!api) return fail(h, SDPGPU_ERR_DEVICE, "%s", g_rccl.err.c_str());
    int rc = ensure_device(h);
    if (rc) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    NCCL_TRY(h, api, api->CommInitRank(&comm, world, id, rank));
    h->comm = comm;
    h->multi_copy = false;
    return SDPGPU_OK;
  } catch (const std::exception& e) {
    return fail(h, SDPGPU_ERR_ARG, "exception: %s", e.what());
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "unknown exception");
  }
}

int sdpgpu_comm_destroy(sdpgpu_handle* h) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (h->allocated) {
    int rc = ensure_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIP_TRY(h, hipStreamSynchronize(h->comm_stream));
  }
  comm_release(h);
  return SDPGPU_OK;
}

int sdpgpu_exchange(sdpgpu_handle* h, int32_t period) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (period < 1 || period > h->T) return fail(h, SDPGPU_ERR_ARG, "exchange: period %d out of 1..%d", period, h->T);
  if (!h->comm) return fail(h, SDPGPU_ERR_STATE, "exchange: no communicator (sdpgpu_comm_init first)");
  if (!h->allocated || !h->period_done[period - 1]) return fail(h, SDPGPU_ERR_STATE, "exchange: period %d has not been run", period);
  try {
    RcclApi* api = rccl();
    if (!api) return fail(h, SDPGPU_ERR_DEVICE, "%s", g_rccl.err.c_str());
    int rc = ensure_device(h);
    if (rc) return rc;
    return enqueue_allgather(h, api, period, h->stream);
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "unknown exception");
  }
}

int sdpgpu_solve_sharded(sdpgpu_handle* h, int32_t flags) {
  if (!h) return SDPGPU_ERR_ARG;
  h->err.clear();
  if (!h->comm) return fail(h, SDPGPU_ERR_STATE, "solve_sharded: no communicator (sdpgpu_comm_init first)");
  try {
    RcclApi* api = rccl();
    if (!api) return fail(h, SDPGPU_ERR_DEVICE, "%s", g_rccl.err.c_str());
    return sweep_rank(h, flags, [&](int period, bool overlapped) {
      return enqueue_allgather(h, api, period, overlapped ? h->comm_stream : h->stream);
    });
  } catch (const std::exception& e) {
    return fail(h, SDPGPU_ERR_ARG, "exception: %s", e.what());
  } catch (...) {
    return fail(h, SDPGPU_ERR_ARG, "unknown exception");
  }
}

// sdpgpu_solve_multi.  Default: ONE host thread drives every rank -- per period the kernels of all ranks are enqueued
// first (each on its own device and stream), then the exchange of that period for all ranks: one RCCL group, or
// device-to-device copies when ranks share a device.  Streams are asynchronous, so the devices run concurrently; the
// host cost is n x (launch + collective) issued serially per period, which is nothing against periods of milliseconds
// (the BASELINE grids from configs[2] up) and is the path every one-GPU rehearsal exercises.
// SDPGPU_SHARDED_THREADS: one host thread per rank instead, each running the very sweep of sdpgpu_solve_sharded on its
// own device (its own communicator, no RCCL group; with shared devices the copies are ordered by a per-period
// rendezvous of the threads) -- for periods of tens of microseconds (configs[1]-sized slabs), where eight devices'
// launches issued from one thread would queue behind each other on the host.
namespace {

struct Rendezvous {  // a reusable barrier of n threads; `broken` releases everybody once a rank has failed
  std::mutex m;
  std::condition_variable cv;
  int n = 0, waiting = 0;
  unsigned long generation = 0;
  bool broken = false;
  int culprit = -1;  // the rank whose failure broke the rendezvous (the others only report "another rank failed")
  bool arrive() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const unsigned long g = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    cv.wait(lk, [&] { return generation != g || broken; });
    return !broken;
  }
  void abandon(int rank) {
    std::lock_guard<std::mutex> lk(m);
    if (!broken) culprit = rank;
    broken = true;
    cv.notify_all();
  }
};

// rank q pulls slab r of the row of `period` from every other rank r, behind r's kernel (ev_comp of r)
int enqueue_copies(sdpgpu_handle** hs, int n, int q, int period, sdpgpu_handle* herr) {
  const size_t cnt = (size_t)(hs[0]->per[period - 1].S_pad / n);
  if (!cnt) return SDPGPU_OK;
  HIP_TRY(herr, hipSetDevice(hs[q]->device));
  char* dst = (char*)exchange_row(hs[q], period);
  for (int r = 0; r < n; ++r) {
    if (r == q) continue;
    const char* src = (const char*)exchange_row(hs[r], period);
    HIP_TRY(herr, hipStreamWaitEvent(hs[q]->stream, hs[r]->ev_comp, 0));
    HIP_TRY(herr, hipMemcpyAsync(dst + (size_t)r * cnt * 8, src + (size_t)r * cnt * 8, cnt * 8, hipMemcpyDeviceToDevice, hs[q]->stream));
  }
  return SDPGPU_OK;
}

}  // namespace

int sdpgpu_solve_multi(sdpgpu_handle** hs, int32_t n, int32_t flags) {
  if (!hs || n < 1 || !hs[0]) return SDPGPU_ERR_ARG;
  sdpgpu_handle* h0 = hs[0];
  h0->err.clear();
  int caller_device = -1;
  (void)hipGetDevice(&caller_device);  // (no device at all: the calls below report it)
  struct RestoreDevice {
    int dev;
    ~RestoreDevice() {
      if (dev >= 0) (void)hipSetDevice(dev);
    }
  } restore{caller_device};
  try {
    for (int r = 0; r < n; ++r) {
      if (!hs[r]) return fail(h0, SDPGPU_ERR_ARG, "solve_multi: handle %d is null", r);
      if (hs[r]->d.rank != r || hs[r]->d.world_size != n)
        return fail(h0, SDPGPU_ERR_ARG, "solve_multi: handle %d was created as rank %d of %d (want rank %d of %d)", r, hs[r]->d.rank, hs[r]->d.world_size, r, n);
      if (hs[r]->T != h0->T) return fail(h0, SDPGPU_ERR_ARG, "solve_multi: handles describe different horizons");
      for (int q = 0; q < r; ++q)
        if (hs[q] == hs[r]) return fail(h0, SDPGPU_ERR_ARG, "solve_multi: handle %d given twice", r);
    }
    // devices: distinct -> RCCL (ncclCommInitAll), shared -> copies
    std::vector<int> dev((size_t)n);
    bool distinct = true;
    for (int r = 0; r < n; ++r) {
      sdpgpu_handle* h = hs[r];
      int rc = allocate(h);
      if (rc) return h == h0 ? rc : fail(h0, rc, "rank %d: %s", r, h->err.c_str());
      rc = ensure_device(h);
      if (rc) return h == h0 ? rc : fail(h0, rc, "rank %d: %s", r, h->err.c_str());
      if (h->device < 0) HIP_TRY(h0, hipGetDevice(&h->device));
      dev[(size_t)r] = h->device;
      for (int q = 0; q < r; ++q) distinct = distinct && dev[(size_t)q] != dev[(size_t)r];
    }
    // the ranks must describe ONE problem: the same family and, per period, the same grid and padded row (the exchange
    // moves S_pad / n elements per rank into every handle's row)
    for (int r = 1; r < n; ++r) {
      const sdpgpu_handle* h = hs[r];
      bool same = h->d.family == h0->d.family && h->d.store_all_values == h0->d.store_all_values && h->custom == h0->custom;
      for (int t = 0; same && t < h0->T; ++t) {
        const PeriodInfo &a = h0->per[t], &b = h->per[t];
        same = a.S == b.S && a.S_pad == b.S_pad && a.nD == b.nD && a.g.nx == b.g.nx && a.g.nc == b.g.nc && a.g.nq == b.g.nq &&
               a.g.x_lo == b.g.x_lo && a.g.k_lo == b.g.k_lo;
      }
      if (!same) return fail(h0, SDPGPU_ERR_ARG, "solve_multi: handle %d describes another problem than handle 0 (family, grid or pmf sizes differ)", r);
    }
    // SDPGPU_MULTI_EXCHANGE = "copy": device-to-device copies even on distinct devices; "rccl": the communicator branch
    // (ncclCommInitAll + one group of all-gathers per period) even when ranks share a device -- RCCL itself refuses that, the
    // tests' stand-in (SDPGPU_RCCL_LIB=tests/mock_rccl) does not, which is how this branch runs on a one-GPU box
    const char* force = std::getenv("SDPGPU_MULTI_EXCHANGE");
    const bool force_rccl = force && std::strcmp(force, "rccl") == 0;
    const bool want_copy = (!distinct && !force_rccl) || (force && std::strcmp(force, "copy") == 0);
    bool same_group = true, any_comm = false;
    for (int r = 0; r < n; ++r) {
      same_group = same_group && hs[r]->siblings.size() == (size_t)n && std::equal(hs, hs + n, hs[r]->siblings.begin());
      any_comm = any_comm || hs[r]->comm != nullptr;
    }
    RcclApi* api = nullptr;
    if (!want_copy || any_comm) {  // (also when only an earlier communicator has to be destroyed)
      api = rccl();
      if (!api && !want_copy) return fail(h0, SDPGPU_ERR_DEVICE, "%s", g_rccl.err.c_str());
    }
    if (!same_group || hs[0]->multi_copy != want_copy || (!want_copy && !hs[0]->comm)) {
      for (int r = 0; r < n; ++r) {
        if (hs[r]->comm && api) {
          HIP_TRY(h0, hipSetDevice(hs[r]->device));
          (void)api->CommDestroy((ncclComm_t)hs[r]->comm);
        }
        hs[r]->comm = nullptr;
      }
      if (!want_copy) {
        std::vector<ncclComm_t> comms((size_t)n, nullptr);
        NCCL_TRY(h0, api, api->CommInitAll(comms.data(), n, dev.data()));
        for (int r = 0; r < n; ++r) hs[r]->comm = comms[(size_t)r];
      }
      for (int r = 0; r < n; ++r) {
        hs[r]->siblings.assign(hs, hs + n);
        hs[r]->multi_copy = want_copy;
      }
    }
    if (want_copy)
      for (int r = 0; r < n; ++r) {
        int rc = ensure_device(hs[r]);
        if (!rc) rc = ensure_comm_stream(hs[r]);  // (for its events)
        if (rc) return hs[r] == h0 ? rc : fail(h0, rc, "rank %d: %s", r, hs[r]->err.c_str());
      }
    const int last = (flags & SDPGPU_SHARDED_GATHER_FIRST) ? 1 : 2;

    if ((flags & SDPGPU_SHARDED_THREADS) && n > 1) {
      // one host thread per rank: the sweep of sdpgpu_solve_sharded on its own device
      Rendezvous meet;
      meet.n = n;
      std::vector<int> rcs((size_t)n, SDPGPU_OK);
      const int inner = flags & ~(SDPGPU_SHARDED_THREADS | (want_copy ? SDPGPU_SHARDED_OVERLAP : 0));
      // (tests: "kernel:R:P" = rank R fails where its kernel of period P would be launched, "collective:R:P" = where its
      // all-gather of that period would be enqueued -- neither may leave the other ranks inside a collective)
      int inject_kind = 0, inject_rank = -1, inject_period = -1;
      if (const char* inj = std::getenv("SDPGPU_TEST_FAIL_RANK")) {
        char kind[16] = {0};
        if (std::sscanf(inj, "%15[a-z]:%d:%d", kind, &inject_rank, &inject_period) == 3)
          inject_kind = std::strcmp(kind, "kernel") == 0 ? 1 : std::strcmp(kind, "collective") == 0 ? 2 : 0;
      }
      // A rank that fails after its peers have enqueued the all-gather of a period would leave them waiting for a
      // participant that never comes (for ever under SDPGPU_SHARDED_SYNC): the failing thread then aborts EVERY
      // communicator of the call (ncclCommAbort, which is what releases a blocked collective), once; the communicators
      // are gone afterwards and the next call builds new ones.
      std::once_flag abort_once;
      bool aborted = false;
      auto abort_all = [&] {
        std::call_once(abort_once, [&] {
          aborted = true;
          if (!api || !api->CommAbort) return;
          for (int q = 0; q < n; ++q)
            if (hs[q]->comm) (void)api->CommAbort((ncclComm_t)hs[q]->comm);
        });
      };
      auto body = [&](int r) {
        sdpgpu_handle* h = hs[r];
        h->err.clear();
        int rc;
        try {
          rc = sweep_rank(h, inner, [&](int period, bool overlapped) -> int {
            if (inject_kind == 1 && inject_rank == r && inject_period == period)
              return fail(h, SDPGPU_ERR_DEVICE, "injected failure of the kernel launch of period %d (SDPGPU_TEST_FAIL_RANK)", period);
            if (!want_copy) {
              // every rank confirms that its kernel of this period is launched BEFORE any rank enters the collective: a
              // rank that failed in run_period_impl has abandoned the rendezvous and nobody enqueues an all-gather it
              // would never join
              if (!meet.arrive()) return fail(h, SDPGPU_ERR_STATE, "another rank's sweep failed");
              if (inject_kind == 2 && inject_rank == r && inject_period == period)
                return fail(h, SDPGPU_ERR_DEVICE, "injected failure of the all-gather of period %d (SDPGPU_TEST_FAIL_RANK)", period);
              return enqueue_allgather(h, api, period, overlapped ? h->comm_stream : h->stream);
            }
            // shared device: publish "my kernel of this period is enqueued", wait for everybody's, then pull
            HIP_TRY(h, hipEventRecord(h->ev_comp, h->stream));
            if (!meet.arrive()) return fail(h, SDPGPU_ERR_STATE, "another rank's sweep failed");
            int rc2 = enqueue_copies(hs, n, r, period, h);
            // (nobody re-records its event for the next period before every rank has enqueued its waits on this one)
            if (!meet.arrive() && !rc2) rc2 = fail(h, SDPGPU_ERR_STATE, "another rank's sweep failed");
            return rc2;
          });
        } catch (...) {
          rc = fail(h, SDPGPU_ERR_ARG, "exception in the rank's thread");
        }
        if (rc) {
          const std::string why = h->err;  // (kept: the abort below must not replace the reason)
          meet.abandon(r);
          if (!want_copy) abort_all();
          h->err = why;
        }
        rcs[(size_t)r] = rc;
      };
      std::vector<std::thread> th;
      th.reserve((size_t)n);
      for (int r = 0; r < n; ++r) th.emplace_back(body, r);
      for (auto& t : th) t.join();
      if (aborted)  // (ncclCommAbort frees a communicator as ncclCommDestroy does; without it the handles keep theirs)
        for (int r = 0; r < n && api && api->CommAbort; ++r) hs[r]->comm = nullptr;
      for (int pass = 0; pass < 2; ++pass)  // the rank that failed first, then anybody else
        for (int r = 0; r < n; ++r) {
          if (!rcs[(size_t)r] || (pass == 0 && r != meet.culprit)) continue;
          const std::string why = hs[r]->err;
          return fail(h0, rcs[(size_t)r], "rank %d: %s", r, why.c_str());
        }
      return SDPGPU_OK;
    }

    for (int r = 0; r < n; ++r) {
      sdpgpu_handle* h = hs[r];
      HIP_TRY(h0, hipSetDevice(h->device));
      std::fill(h->period_done.begin(), h->period_done.end(), 0);
      HIP_TRY(h0, hipEventRecord(h->ev_solve0, h->stream));
    }
    for (int period = h0->T; period >= 1; --period) {
      for (int r = 0; r < n; ++r) {
        sdpgpu_handle* h = hs[r];
        h->err.clear();
        int rc = run_period_impl(h, period);
        if (rc) return h == h0 ? rc : fail(h0, rc, "rank %d: %s", r, h->err.c_str());
        count_cells(h, period);
      }
      if (period < last) continue;
      if (!want_copy) {
        NCCL_TRY(h0, api, api->GroupStart());
        for (int r = 0; r < n; ++r) {
          sdpgpu_handle* h = hs[r];
          int rc = enqueue_allgather(h, api, period, h->stream);
          if (rc) {
            (void)api->GroupEnd();
            return h == h0 ? rc : fail(h0, rc, "rank %d: %s", r, h->err.c_str());
          }
        }
        NCCL_TRY(h0, api, api->GroupEnd());
      } else {
        for (int r = 0; r < n; ++r) {
          HIP_TRY(h0, hipSetDevice(hs[r]->device));
          HIP_TRY(h0, hipEventRecord(hs[r]->ev_comp, hs[r]->stream));
        }
        for (int q = 0; q < n; ++q) {
          int rc = enqueue_copies(hs, n, q, period, h0);
          if (rc) return rc;
        }
      }
    }
    for (int r = 0; r < n; ++r) {
      sdpgpu_handle* h = hs[r];
      int rc = ensure_device(h);
      if (!rc) rc = flush_api(h);
      if (rc) return h == h0 ? rc : fail(h0, rc, "rank %d: %s", r, h->err.c_str());
      HIP_TRY(h0, hipEventRecord(h->ev_solve1, h->stream));
      h->solve_timed = true;
    }
    if (flags & SDPGPU_SHARDED_SYNC)
      for (int r = 0; r < n; ++r) {
        HIP_TRY(h0, hipSetDevice(hs[r]->device));
        HIP_TRY(h0, hipStreamSynchronize(hs[r]->stream));
      }
    return SDPGPU_OK;
  } catch (const std::exception& e) {
    return fail(h0, SDPGPU_ERR_ARG, "exception: %s", e.what());
  } catch (...) {
    return fail(h0, SDPGPU_ERR_ARG, "unknown exception");
  }
}

}  // extern "C"

// sdpgpu_window.hip -- host side of the F1 window kernel (plan, chunk rows and key rows, deferred finalize)
// and of the F2 row-window kernel (sdp_window.hpp).
#include "sdpgpu_internal.hpp"
#include "sdp_window.hpp"

namespace sdpgpu_detail {

// ---- window kernel (F1) -----------------------------------------------------------------------


// F1 / F2 on a unit-stride demand grid d_j = d_0 + j*step (supports with gaps are laid out on one, see PeriodInfo).
bool window_eligible(const sdpgpu_handle* h, int period) {
  if (h->custom && !h->level_shape) return false;  // (user lambdas of the level shape run on the F1 window kernel from tables)
  if (!h->counts[(size_t)period - 1].empty()) return false;  // caller-supplied action counts: generic kernel
  if (h->d.family != SDPGPU_FAMILY_BACKORDER && h->d.family != SDPGPU_FAMILY_LEADTIME) return false;
  const PeriodInfo& p = h->per[period - 1];
  if (p.nD_win <= 0) return false;
  if (p.S >= 2147483647LL - 4096) return false;
  if (h->n_actions_full + p.nD_win > 3500) return false;
  return true;
}

// One task = one wave = (tile of 64*S states, run of R-blocks).  The measured timeline of a SIMD is task
// after task, so a launch costs  rounds x task time  with rounds = ceil(tasks / 1024 SIMDs): pick the
// register block R, the states per lane S and the number of chunks per tile that minimise it (fewest chunks
// on ties: fewer chunk rows, less staging).  More states per lane = fewer fp64 operations per cell
// ((5 + 4(S-1)) / S, see window_f1_kernel) but bigger, fewer tasks: small grids keep S low.
static WinPlan plan_window_search(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi, std::string* why);

WinPlan plan_window(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi, std::string* why) {
  WinPlanCache& c = h->per[period - 1].win_plan;
  const bool may_chunk = h->fuse_combine && h->d.store_all_values;
  if (!(c.valid && c.lo == lo && c.hi == hi && c.win_r == h->win_r && c.win_s == h->win_s && c.win_nch == h->win_nch &&
        c.may_chunk == may_chunk)) {
    c.why.clear();
    c.plan = plan_window_search(h, period, lo, hi, &c.why);
    c.lo = lo;
    c.hi = hi;
    c.win_r = h->win_r;
    c.win_s = h->win_s;
    c.win_nch = h->win_nch;
    c.may_chunk = may_chunk;
    c.valid = true;
  }
  if (why && !c.plan.R) *why = c.why;
  return c.plan;
}

static WinPlan plan_window_search(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi, std::string* why) {
  const PeriodInfo& p = h->per[period - 1];
  const int A = h->n_actions_full, D = p.nD_win;
  WinPlan best;
  double best_cost = -1;
  // The plan is chosen from the NOMINAL slab S_pad / world_size, which is the same on every rank: ranks
  // must agree on whether a period's row is exchanged as keys or as fp64 values, whatever their own
  // (possibly clipped or empty) slab looks like.
  const int64_t nominal = p.S_pad / std::max(1, h->d.world_size);
  auto rup = [](int v, int r) { return (v + r - 1) / r * r; };
  struct Cand {
    int r, s, occupancy;  // occupancy: waves a SIMD can hold within the register budget
  };
  // (VGPRs: 107 / 62 / 54 for S = 1, R = 8 / 5 / 4; 130 / 116 / 73 for S = 2; 125 for R = 4, S = 4; 244 for R = 4, S = 8)
  const Cand cand[] = {{8, 1, 6}, {5, 1, 8}, {4, 1, 9}, {8, 2, 3}, {4, 2, 5}, {4, 4, 3}, {8, 4, 2}, {4, 8, 2}};
  const bool may_chunk = h->fuse_combine && h->d.store_all_values;
  bool shape_seen = false, lds_rejected = false, chunk_rejected = false;
  size_t lds_least = 0;
  for (const Cand& c : cand) {
    const int r = c.r, sl = c.s, nw = r + sl - 1, ts = 64 * sl;
    if (h->win_r && r != h->win_r) continue;
    if (h->win_s && sl != h->win_s) continue;
    // 32-cell blocks: (4, 8) -- 3.47 operations per cell against (4, 4)'s 3.69, two waves per SIMD -- competes since the
    // step requests its LDS reads a step ahead (two waves then sustain 0.93 of the issue rate: 53.0 against 55.5 ms on the
    // target grid, 4 % ahead on slabs down to 125,000 states); (8, 4) measured no gain and stays opt-in (SDPGPU_WIN_R + _S)
    const bool big_block = r * sl >= 32;
    if (big_block && !(r == 4 && sl == 8) && !(h->win_r && h->win_s)) continue;
    shape_seen = true;
    const int64_t n_tiles = std::max<int64_t>(1, (nominal + ts - 1) / ts);
    const int64_t own_tiles = (hi - lo + ts - 1) / ts;
    const int d_pad = rup(D, nw);
    const int blocks_total = rup(A, r) / r;
    // cost of one R-block on one SIMD, in fp64-instruction units per lane: (5 + 4(S-1)) ops per S cells of an
    // action plus a per-step overhead (LDS read, scalar load, waits) that a bigger register block amortises
    const double block_cost = (double)D * ((5.0 + 4.0 * (sl - 1)) * r + 3.0) + 60.0 + 2.0 * (sl - 1) * r;
    // a forced chunk count is taken as "about that many": the realisable plan with the same blocks per chunk
    int forced_nch = 0;
    if (h->win_nch) {
      const int want = std::max(1, std::min(h->win_nch, blocks_total));
      const int bpc_w = (blocks_total + want - 1) / want;
      forced_nch = (blocks_total + bpc_w - 1) / bpc_w;
    }
    for (int nch = 1; nch <= blocks_total; ++nch) {
      if (forced_nch && nch != forced_nch) continue;
      const int bpc = (blocks_total + nch - 1) / nch;
      if ((blocks_total + bpc - 1) / bpc != nch) continue;  // same plan as a smaller nch
      if (nch > 1 && !may_chunk) {  // chunk rows need the deferred key/finalize scheme
        chunk_rejected = true;
        continue;
      }
      const int span = ts + bpc * r + d_pad + sl;
      const size_t smem = sdp::win_wg_lds(span, D);
      // A workgroup is four tasks, one per SIMD: `occupancy` workgroups per CU by registers, kLdsPerCU / smem by LDS
      // (gfx950: 160 KiB per CU -- the one-task-per-tile plan of the 500-action, 200-demand grid on the (4, 8) block is
      // 78.4 KiB per workgroup, and two of them are resident like any other (4, 8) plan's).
      const int occ = std::min(c.occupancy, lds_workgroups(smem));
      if (occ < 1) {
        lds_rejected = true;
        lds_least = lds_least ? std::min(lds_least, smem) : smem;
        continue;
      }
      const int64_t tasks = n_tiles * nch;
      const int64_t q = (tasks + 1023) / 1024;  // tasks of the busiest SIMD
      // A SIMD holds at most `occ` of them at a time and, with priority by progress, resident waves finish
      // together: q tasks run as groups of `occ` plus a remainder group, a group of w tasks at the fp64 issue
      // rate w resident waves sustain (per-wave stamps: 0.6 alone, 0.85 two, 0.91 three, 0.94 four, 0.97 eight).
      auto eff = [big_block](int64_t w) { return w >= 8 ? 0.97 : (w >= 4 ? 0.94 : (w >= 3 ? 0.91 : (w >= 2 ? (big_block ? 0.93 : 0.85) : 0.60))); };
      const int64_t full = q / occ, rest = q % occ;
      const double task_units = full * occ / eff(occ) + (rest ? rest / eff(rest) : 0.0);
      const double staging = 400.0 + 4.0 * span;
      const double cost = task_units * (bpc * block_cost + staging);
      if (best_cost < 0 || cost < best_cost * 0.999) {
        best_cost = cost;
        best.R = r;
        best.S = sl;
        best.d_pad = d_pad;
        best.n_chunks = nch;
        best.chunk_blocks = bpc;
        best.n_tiles = (int)own_tiles;
        best.n_tasks = (int)(own_tiles * nch);
        best.smem = smem;
      }
    }
  }
  if (!best.R && why) {
    char buf[320];
    if (!shape_seen)
      std::snprintf(buf, sizeof buf, "window kernel: no instantiation for the forced block SDPGPU_WIN_R=%d SDPGPU_WIN_S=%d "
                    "(have R x S = 8x1 5x1 4x1 8x2 4x2 4x4 8x4 4x8)", h->win_r, h->win_s);
    else if (lds_rejected)
      std::snprintf(buf, sizeof buf, "window kernel: %d actions x %d demand steps need %zu B of LDS per workgroup%s, over the %zu B of a "
                    "compute unit%s", A, D, lds_least, h->win_nch ? " with the forced chunk count" : "", kLdsPerCU,
                    chunk_rejected ? " (chunking is off: ping-pong tables or SDPGPU_FUSE_COMBINE=0)" : "");
    else
      std::snprintf(buf, sizeof buf, "window kernel: the forced plan (SDPGPU_WIN_R=%d SDPGPU_WIN_S=%d SDPGPU_WIN_NCH=%d) needs chunk rows, "
                    "which are off (ping-pong tables or SDPGPU_FUSE_COMBINE=0)", h->win_r, h->win_s, h->win_nch);
    *why = buf;
  }
  return best;
}

hipError_t ensure_partials(sdpgpu_handle* h, int b, size_t need) {
  if (need <= h->part_elems[b]) return hipSuccess;
  // (re)allocation frees a buffer earlier launches may still read: drain the stream first
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return e;
  if (h->d_part_val[b]) (void)hipFree(h->d_part_val[b]);
  if (h->d_part_idx[b]) (void)hipFree(h->d_part_idx[b]);
  h->d_part_val[b] = nullptr;
  h->d_part_idx[b] = nullptr;
  h->part_elems[b] = 0;
  e = hipMalloc((void**)&h->d_part_val[b], need * sizeof(double));
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&h->d_part_idx[b], need * sizeof(int32_t));
  if (e != hipSuccess) return e;
  h->part_elems[b] = need;
  return hipSuccess;
}

template <bool MAXDIR>
hipError_t launch_combine(const double* pv, const int32_t* pi, int n_chunks, int64_t stride, double* v_cur, int32_t* pol,
                          int64_t lo, int64_t hi, hipStream_t st) {
  unsigned blocks = (unsigned)((hi - lo + 255) / 256);
  hipLaunchKernelGGL((sdp::window_combine_kernel<MAXDIR>), dim3(blocks), dim3(256), 0, st, pv, pi, n_chunks, stride, v_cur, pol, lo, hi);
  return hipGetLastError();
}

// Turn every pending period's keys + chunk rows into its final V_t / policy rows: one launch.
hipError_t flush_pending(sdpgpu_handle* h) {
  if (h->n_pending == 0) return hipSuccess;
  // The job list lives in the handle, at a fixed address: a caller that captures a sweep in a HIP graph records this
  // upload as a memcpy node FROM that address, and every sweep writes the same list there.
  if (h->jobs_host.size() < (size_t)h->T * sizeof(sdp::FinalizeJob)) h->jobs_host.resize((size_t)h->T * sizeof(sdp::FinalizeJob));
  sdp::FinalizeJob* jobs = reinterpret_cast<sdp::FinalizeJob*>(h->jobs_host.data());
  size_t n_jobs = 0;
  int64_t total = 0;
  for (int t = 0; t < h->T; ++t) {
    if (h->pending_chunks[t] <= 0) continue;
    const PeriodInfo& p = h->per[t];
    sdp::FinalizeJob J{};
    J.keys = h->d_keys + (size_t)t * h->key_stride;
    J.part_val = h->d_chunk_val + h->chunk_off[t] - chunk_row_lo(h, p);
    J.part_idx = h->d_chunk_idx + h->chunk_off[t] - chunk_row_lo(h, p);
    J.v_out = h->d_values + p.v_off;
    J.pol_out = h->d_policy + p.pol_off - p.lo;
    J.stride = chunk_row_cap(h, p);
    J.lo = p.lo;
    J.hi = p.hi;
    // V_t is decoded over the whole row (after the all-gather every rank holds all keys), the policy
    // only for this rank's slab
    J.vlo = h->d.world_size > 1 ? 0 : p.lo;
    J.vhi = h->d.world_size > 1 ? p.S : p.hi;
    J.first = total;
    J.n_chunks = h->pending_chunks[t];
    total += J.vhi - J.vlo;
    jobs[n_jobs++] = J;
    h->pending_chunks[t] = 0;
  }
  h->n_pending = 0;
  if (n_jobs == 0 || total == 0) return hipSuccess;
  if (!h->d_jobs) {
    hipError_t e = hipMalloc((void**)&h->d_jobs, (size_t)h->T * sizeof(sdp::FinalizeJob));
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipMemcpyAsync(h->d_jobs, jobs, n_jobs * sizeof(sdp::FinalizeJob), hipMemcpyHostToDevice, h->stream);
  if (e != hipSuccess) return e;
  h->flush_uploads++;
  hipLaunchKernelGGL(sdp::finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->d_jobs,
                     (int)n_jobs, total);
  return hipGetLastError();
}

// ---- row-window kernel (F2) ---------------------------------------------------------------
template <int R, int S, bool MAXDIR>
hipError_t launch_row_r(const sdp::RowParams& W, size_t smem, bool future, const double* v_next, double* out_val,
                        int32_t* out_idx, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st) {
  if (!grid_ok((int64_t)W.n_tiles * W.n_chunks)) return hipErrorInvalidValue;
  dim3 grid((unsigned)((int64_t)W.n_tiles * W.n_chunks));
  static LdsMark mark_f, mark_l;
  hipError_t ea = future ? lds_allow(sdp::window_f2_kernel<R, S, MAXDIR, true>, smem, &mark_f)
                         : lds_allow(sdp::window_f2_kernel<R, S, MAXDIR, false>, smem, &mark_l);
  if (ea != hipSuccess) return ea;
  if (future)
    hipLaunchKernelGGL((sdp::window_f2_kernel<R, S, MAXDIR, true>), grid, dim3(256), smem, st, W, v_next, out_val, out_idx, pmf_p, lo, hi);
  else
    hipLaunchKernelGGL((sdp::window_f2_kernel<R, S, MAXDIR, false>), grid, dim3(256), smem, st, W, v_next, out_val, out_idx, pmf_p, lo, hi);
  return hipGetLastError();
}

hipError_t launch_row_window(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                             int32_t* pol, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st) {
  {
    hipError_t ef = flush_pending(h);
    if (ef != hipSuccess) return ef;
  }
  const PeriodInfo& p = h->per[period - 1];
  const int A = h->n_actions_full, D = p.nD_win;
  auto rup = [](int v, int r) { return (v + r - 1) / r * r; };
  int R = 0;
  int64_t best_cost = -1;
  const int cand[3] = {4, 5, 8};  // ties go to the smaller block (measured: R = 4 is 2-3 % ahead of R = 5 at S = 2)
  for (int r : cand) {
    if (h->win_r && r != h->win_r) continue;
    int64_t cost = (int64_t)rup(A, r) * rup(D, r);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      R = r;
    }
  }
  if (!R) R = 8;
  // states per lane (tiles of 64 S states; SDPGPU_WIN_S overrides): the S that minimises padded states x operations per cell
  // (4 + 1/S), ties to the larger S.  Measured on configs[3] (wave-private rows, 16-byte LDS reads): S = 2 / 4 -> 6.93 / 7.31e12
  // cells/s, its pipeline shape 7.98 / 8.57e12.
  int SL = h->win_s;
  if (!SL) {
    double best_c = 0;
    for (int sl : {1, 2, 4}) {
      if (sl == 4 && R == 8) continue;  // (no 8 x 4 instantiation: too many registers)
      const double c = (double)rup((int)p.g.nx, 64 * sl) * (4.0 + 1.0 / sl);
      if (!SL || c <= best_c) {
        best_c = c;
        SL = sl;
      }
    }
  }
  if (SL != 1 && SL != 2 && SL != 4) SL = 2;
  if (SL == 4 && R == 8) SL = 2;
  const int TSZ = 64 * SL;
  const bool future = period < h->T;
  h->per[period - 1].ops_cell = future ? 4.0 + 1.0 / SL : 2.0 + 1.0 / SL;  // (c0 + M shared by SL cells, see window_f2_kernel)
  h->per[period - 1].lds_cell = 8.0 * ((future ? R : 0) + 1) / (double)(R * SL);  // M and one V entry per action, per SL states
  sdp::RowParams W{};
  W.lev0 = p.g.x_lo - h->pmf_d[period - 1][0];
  W.step = h->d.step;
  W.h = h->d.holding_cost;
  W.pi = h->d.penalty_cost;
  W.K = h->d.fixed_order_cost;
  W.v = h->d.unit_order_cost;
  if (future) {
    W.idx_off = (int32_t)((W.lev0 - h->per[period].g.x_lo) / h->d.step);
    W.next_last = (int32_t)(h->per[period].g.nx - 1);
    W.next_nx = (int32_t)h->per[period].g.nx;
  }
  W.cur_nx = (int32_t)p.g.nx;
  W.nq1 = (int32_t)p.g.nq1;
  W.plane_stride = P.lead2 ? (int64_t)W.nq1 * W.next_nx : (int64_t)W.next_nx;
  W.tiles_per_row = (int32_t)((p.g.nx + TSZ - 1) / TSZ);
  W.n_actions = A;
  W.d_pad = rup(D, 4);  // the demand loop is unrolled by S (1, 2 or 4); padded steps carry p = 0
  const int span = TSZ + W.d_pad + 2;  // (window_f2_kernel: two spare slots, even)
  const int blocks_total = rup(A, R) / R;
  // one R-block per wave: chunks of 4 R-blocks; every wave stages the R row segments of its own block in its own LDS
  // region, so the budget (rows of `span` doubles) bounds the number of waves that take blocks, not the chunk
  int waves = std::min(4, blocks_total);
  // (the read-out scratch of a wave that stages rows lies inside its row region)
  auto lds = [&](int wv) { return (size_t)span * 8 * (1 + (future ? wv * R : 0)) + (size_t)(future ? 4 - wv : 4) * TSZ * 12; };
  // (two workgroups per compute unit stay resident up to half its 160 KiB each; a single wave's rows may take all of it)
  while (waves > 1 && lds(waves) > kLdsPerCU / 2) --waves;
  if (lds(waves) > kLdsPerCU) {
    char buf[200];
    std::snprintf(buf, sizeof buf, "row-window kernel: %d demand steps need %zu B of LDS per workgroup, over the %zu B of a compute unit",
                  D, lds(waves), kLdsPerCU);
    h->plan_error = buf;
    return hipErrorInvalidValue;
  }
  // the run of row tiles that covers [lo, hi)
  auto tile_of = [&](int64_t idx) { return (int32_t)((idx / p.g.nx) * W.tiles_per_row + (idx % p.g.nx) / TSZ); };
  W.tile0 = tile_of(lo);
  W.n_tiles = tile_of(hi - 1) - W.tile0 + 1;
  // Blocks per chunk: a whole number of blocks per wave, as many as still leave about 6.5 rounds of workgroups on the chip
  // (5000 at three per CU).  What a workgroup pays once -- M staged behind a barrier, the read-out behind another, a
  // chunk row of (value, action) pairs for the combine pass -- is then shared by several blocks; with ONE chunk there are no
  // chunk rows and no combine pass at all.  Measured (`profiles/r02_f2_chunk_sweep.txt`): configs[3], 800 tiles per period:
  // 13 / 7 / 5 / 1 chunks -> 27.3 / 26.7 / 27.2 / 33.8 ms; its pipeline shape, 40,000 tiles: 13 / 4 / 1 -> 93.8 / 89.6 / 88.6 ms.
  int bpc = waves;
  for (int m = 2; (m - 1) * waves < blocks_total; ++m) {
    const int cand_bpc = std::min(m * waves, rup(blocks_total, waves));
    if ((int64_t)W.n_tiles * ((blocks_total + cand_bpc - 1) / cand_bpc) < 5000) break;
    bpc = cand_bpc;
  }
  if (h->win_nch) bpc = std::max(1, (blocks_total + h->win_nch - 1) / h->win_nch);
  W.waves_active = waves;
  W.chunk_actions = bpc * R;
  W.n_chunks = (blocks_total + bpc - 1) / bpc;
  double* out_val = v_cur;
  int32_t* out_idx = pol;
  if (W.n_chunks > 1) {
    int64_t slab = hi - lo;
    const int b = period & 1;
    hipError_t e = ensure_partials(h, b, (size_t)W.n_chunks * (size_t)slab);
    if (e != hipSuccess) return e;
    W.partial_stride = slab;
    out_val = h->d_part_val[b] - lo;
    out_idx = h->d_part_idx[b] - lo;
  }
  hipError_t e = hipErrorInvalidValue;
  size_t smem = lds(waves);
#define SDP_ROW(RR, SS)                                                                                          \
  if (R == RR && SL == SS)                                                                                       \
    e = P.maxdir ? launch_row_r<RR, SS, true>(W, smem, future, v_next, out_val, out_idx, pmf_p, lo, hi, st)      \
                 : launch_row_r<RR, SS, false>(W, smem, future, v_next, out_val, out_idx, pmf_p, lo, hi, st);
  SDP_ROW(8, 1) SDP_ROW(5, 1) SDP_ROW(4, 1)
  SDP_ROW(8, 2) SDP_ROW(5, 2) SDP_ROW(4, 2)
  SDP_ROW(5, 4) SDP_ROW(4, 4)
#undef SDP_ROW
  if (e != hipSuccess) return e;
  if (W.n_chunks > 1)
    e = P.maxdir ? launch_combine<true>(out_val, out_idx, W.n_chunks, W.partial_stride, v_cur, pol, lo, hi, st)
                 : launch_combine<false>(out_val, out_idx, W.n_chunks, W.partial_stride, v_cur, pol, lo, hi, st);
  return e;
}

// The interior run of slab tiles of a period: tiles whose whole V_{t+1} footprint
// [i0 + idx_off - (D-1), i0 + 63 + idx_off + A - 1] (before the clamp to the grid) lies inside this rank's
// slab of the next period's row, or is clamped at a grid edge this rank owns.
bool window_interior_tiles(const sdpgpu_handle* h, int period, int64_t lo, int64_t hi, int* first, int* count) {
  if (h->d.family != SDPGPU_FAMILY_BACKORDER || period >= h->T || h->d.world_size == 1) return false;
  const PeriodInfo& p = h->per[period - 1];
  const PeriodInfo& pn = h->per[period];
  const int64_t ts = plan_window(h, period, lo, hi).tile_states();
  const int64_t n_tiles = (hi - lo + ts - 1) / ts;
  const double lev0 = p.g.x_lo - h->pmf_d[period - 1][0];
  const int64_t idx_off = (int64_t)((lev0 - pn.g.x_lo) / h->d.step);
  const int64_t A = h->n_actions_full, D = p.nD_win;
  int64_t f = -1, c = 0;
  for (int64_t u = 0; u < n_tiles; ++u) {
    const int64_t i0 = lo + u * ts;
    int64_t a = i0 + idx_off - (D - 1), b = i0 + ts - 1 + idx_off + A - 1;
    a = std::max<int64_t>(0, std::min<int64_t>(a, pn.g.nx - 1));  // the kernel clamps reads to the grid
    b = std::max<int64_t>(0, std::min<int64_t>(b, pn.g.nx - 1));
    const bool inside = a >= pn.lo && b < pn.hi;
    if (inside) {
      if (f < 0) f = u;
      if (u != f + c) return false;  // not one contiguous run: do not split
      ++c;
    }
  }
  if (c <= 0) return false;
  if (first) *first = (int)f;
  if (count) *count = (int)c;
  return true;
}

hipError_t launch_window(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                         int32_t* pol, const double* pmf_d, const double* pmf_p, int64_t lo, int64_t hi, hipStream_t st,
                         int part) {
  (void)pmf_d;
  if (h->d.family == SDPGPU_FAMILY_LEADTIME) {
    if (hi <= lo) return hipSuccess;
    return launch_row_window(h, P, period, v_next, v_cur, pol, pmf_p, lo, hi, st);
  }
  // (an empty slab still goes through the bookkeeping below: every rank must treat the row alike)
  PeriodInfo& p = h->per[period - 1];
  WinPlan pl = plan_window(h, period, lo, hi, &h->plan_error);
  if (!pl.R) return hipErrorInvalidValue;
  if (period == h->T && std::getenv("SDPGPU_DEBUG_PLAN"))
    std::fprintf(stderr, "[sdpgpu] window plan: R=%d S=%d chunks=%d blocks/chunk=%d tiles=%d tasks=%d lds=%zu\n", pl.R, pl.S,
                 pl.n_chunks, pl.chunk_blocks, pl.n_tiles, pl.n_tasks, pl.smem);
  const bool future = period < h->T;
  // c0 + M once per (action, m): 1/S; p * imm: 1; p * V once per window entry: (R + S - 1)/(R S); two accumulations
  p.ops_cell = future ? 3.0 + 1.0 / pl.S + (pl.R + pl.S - 1.0) / (pl.R * pl.S) : 2.0 + 1.0 / pl.S;
  p.lds_cell = 24.0 / (pl.R * pl.S);  // one {M, V} entry and one probability per demand step and lane
  const bool chunked = pl.n_chunks > 1;
  // a period is never re-run on top of its own pending rows, and a new sweep (period T) first
  // finalizes what the previous one left: the key rows are about to be reset
  // (the BOUNDARY half of a split period continues what its INTERIOR half started: no reset there)
  const bool continuing = part == SDPGPU_PART_BOUNDARY;
  if (!continuing && h->n_pending > 0 && (h->pending_chunks[period - 1] > 0 || period == h->T)) {
    hipError_t e = flush_pending(h);
    if (e != hipSuccess) return e;
  }
  // where V_{t+1} comes from: its key row while that period is still pending, else the final fp64 row
  const bool keyed_in = future && h->pending_chunks[period] > 0;
  if (chunked) {
    if (!h->d_chunk_val) {  // one-time arenas: a key row per period, the chunk rows of every chunked period
      size_t stride = 0;
      for (const PeriodInfo& q : h->per) stride = std::max<size_t>(stride, (size_t)q.S_pad);
      hipError_t e = hipSuccess;
      if (!h->d_keys) e = hipMalloc((void**)&h->d_keys, (size_t)h->T * stride * sizeof(unsigned long long));
      if (e != hipSuccess) return e;
      h->key_stride = stride;
      h->key_row_clean.assign((size_t)h->T, 0);
      h->chunk_off.assign((size_t)h->T, 0);
      size_t total = 0;
      for (int t = 0; t < h->T; ++t) {
        const PeriodInfo& q = h->per[t];
        h->chunk_off[t] = total;
        if (window_eligible(h, t + 1))
          total += (size_t)plan_window(h, t + 1, q.lo, q.hi).n_chunks * (size_t)std::max<int64_t>(chunk_row_cap(h, q), 0);
      }
      e = hipMalloc((void**)&h->d_chunk_val, std::max<size_t>(total, 1) * sizeof(double));
      if (e == hipSuccess) e = hipMalloc((void**)&h->d_chunk_idx, std::max<size_t>(total, 1) * sizeof(int32_t));
      if (e != hipSuccess) return e;
    }
    if (!continuing && !h->key_row_clean[period - 1]) {
      // reset key rows to the reduction identity: all of them when nothing is pending (the usual case:
      // period T of a new sweep), else only this period's row (periods re-run out of order)
      const bool all = h->n_pending == 0;
      const int64_t n = (all ? (int64_t)h->T : 1) * (int64_t)h->key_stride;
      unsigned long long* base = all ? h->d_keys : h->d_keys + (size_t)(period - 1) * h->key_stride;
      hipLaunchKernelGGL(sdp::key_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, base, n, (int)P.maxdir);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      if (all)
        std::fill(h->key_row_clean.begin(), h->key_row_clean.end(), 1);
      else
        h->key_row_clean[period - 1] = 1;
    }
  }
  sdp::WinParams W{};
  const double d0 = h->pmf_d[period - 1][0];
  W.lev0 = p.g.x_lo - d0;
  W.step = h->d.step;
  W.h = h->d.holding_cost;
  W.pi = h->d.penalty_cost;
  W.K = h->d.fixed_order_cost;
  W.v = h->d.unit_order_cost;
  if (future) {
    W.idx_off = (int32_t)((W.lev0 - h->per[period].g.x_lo) / h->d.step);
    W.next_last = (int32_t)(h->per[period].g.nx - 1);
  }
  if (h->level_shape) {  // user lambdas of the level shape: this period's tables (filled at allocate: sdp_custom_tabulate)
    W.m_tab = h->d_level_tabs + h->level_tab_off[(size_t)period - 1];
    W.c_tab = W.m_tab + h->level_m_n[(size_t)period - 1];
    W.m_tab_min = h->level_m_min[(size_t)period - 1];
    W.m_tab_n = h->level_m_n[(size_t)period - 1];
  }
  W.n_actions = h->n_actions_full;
  W.d_pad = pl.d_pad;
  W.d_main = p.nD_win / (pl.R + pl.S - 1) * (pl.R + pl.S - 1);
  W.maxdir = P.maxdir;
  W.n_demand = p.nD_win;
  W.n_chunks = pl.n_chunks;
  W.chunk_blocks = pl.chunk_blocks;
  W.n_tiles = pl.n_tiles;
  W.n_tasks = pl.n_tasks;
  W.tile_first = 0;
  W.prio_fair = h->win_prio_fair;
  W.tile_gap_at = pl.n_tiles;  // no gap
  W.tile_gap = 0;
  if (part != SDPGPU_PART_ALL) {
    int first = 0, count = 0;
    if (!window_interior_tiles(h, period, lo, hi, &first, &count)) return hipErrorInvalidValue;
    if (part == SDPGPU_PART_INTERIOR) {
      W.tile_first = first;
      W.n_tiles = count;
    } else {  // the tiles below and above the interior run, in one launch
      // The boundary runs are a handful of tiles on the critical path behind the exchange: cut them finer than
      // the plan does (64-state tiles, 4-action register blocks -- same chunks, so the chunk rows line up) so
      // that the few tasks spread over more SIMDs and each is short.
      const int ratio = pl.S;  // plan tiles are ratio x 64 states
      const int fine_r = (pl.R % 4 == 0) ? 4 : pl.R;
      if (ratio > 1 || fine_r != pl.R) {
        const int chunk_actions = pl.chunk_blocks * pl.R;
        pl.n_tiles = (int)((hi - lo + 63) / 64);
        first *= ratio;
        count = std::min(count * ratio, pl.n_tiles - first);  // (the last plan tile may be a partial one)
        pl.chunk_blocks = chunk_actions / fine_r;
        pl.R = fine_r;
        pl.S = 1;
        pl.d_pad = (p.nD_win + fine_r - 1) / fine_r * fine_r;
        pl.smem = sdp::win_wg_lds(64 + chunk_actions + pl.d_pad + 1, p.nD_win);
        W.d_pad = pl.d_pad;
        W.d_main = p.nD_win / fine_r * fine_r;
        W.chunk_blocks = pl.chunk_blocks;
      }
      W.n_tiles = pl.n_tiles - count;
      W.tile_gap_at = first;
      W.tile_gap = count;
    }
    W.n_tasks = W.n_tiles * pl.n_chunks;
    W.tile_gap_at = std::min(W.tile_gap_at, W.n_tiles);
    if (W.n_tiles == 0) return hipSuccess;
  }
  double* out_val = v_cur;
  int32_t* out_idx = pol;
  // one task per tile: the action index goes straight into the policy slab, which holds this rank's states only
  // (a widened range, sdpgpu_run_period_range, also computes neighbours' states: their values, not their policy)
  W.pol_lo = chunked ? INT64_MIN : p.lo;
  W.pol_hi = chunked ? INT64_MAX : p.hi;
  unsigned long long* k_cur = nullptr;
  const unsigned long long* k_next = keyed_in ? h->d_keys + (size_t)period * h->key_stride : nullptr;
  if (chunked) {
    W.partial_stride = chunk_row_cap(h, p);
    out_val = h->d_chunk_val + h->chunk_off[period - 1] - chunk_row_lo(h, p);  // the kernel indexes rows by flat state index
    out_idx = h->d_chunk_idx + h->chunk_off[period - 1] - chunk_row_lo(h, p);
    k_cur = h->d_keys + (size_t)(period - 1) * h->key_stride;
  }
  if (W.n_tasks > 0 && !grid_ok((W.n_tasks + 3) / 4)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)std::max(1, (W.n_tasks + 3) / 4));
#ifdef SDP_STAMPS
  static unsigned long long* d_stamps = nullptr;
  if (!d_stamps) (void)hipMalloc((void**)&d_stamps, (size_t)1 << 24);
  unsigned long long* stamps = (period == 2) ? d_stamps : nullptr;  // record one mid-sweep launch
#define SDP_STAMP_ARG , stamps
#else
#define SDP_STAMP_ARG
#endif
  if (W.n_tasks > 0) {
#define SDP_WIN_GO(RR, SS, FU, KI)                                                                                        \
  do {                                                                                                                    \
    static LdsMark mark;                                                                                                  \
    hipError_t ea = lds_allow(sdp::window_f1_kernel<RR, SS, FU, KI>, pl.smem, &mark);                                     \
    if (ea != hipSuccess) return ea;                                                                                      \
    hipLaunchKernelGGL((sdp::window_f1_kernel<RR, SS, FU, KI>), grid, dim3(256), pl.smem, st, W, v_next, k_next, out_val, \
                       out_idx, k_cur, pmf_p, lo, hi SDP_STAMP_ARG);                                                      \
  } while (0)
#define SDP_WIN_R(RR, SS)                      \
  if (pl.R == RR && pl.S == SS) {              \
    if (!future)                               \
      SDP_WIN_GO(RR, SS, false, false);        \
    else if (keyed_in)                         \
      SDP_WIN_GO(RR, SS, true, true);          \
    else                                       \
      SDP_WIN_GO(RR, SS, true, false);         \
    launched = true;                           \
  }
  bool launched = false;
  SDP_WIN_R(8, 1)
  SDP_WIN_R(5, 1)
  SDP_WIN_R(4, 1)
  SDP_WIN_R(8, 2)
  SDP_WIN_R(4, 2)
  SDP_WIN_R(4, 4)
  SDP_WIN_R(8, 4)
  SDP_WIN_R(4, 8)
  if (!launched) return hipErrorInvalidValue;
#undef SDP_WIN_R
#undef SDP_WIN_GO
#undef SDP_STAMP_ARG
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
#ifdef SDP_STAMPS
  if (period == 1) {
    std::vector<unsigned long long> hs((size_t)pl.n_tasks * 5);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    if (FILE* f = std::fopen("gpurun_out/stamps.txt", "w")) {
      for (size_t i = 0; i + 4 < hs.size(); i += 5)
        std::fprintf(f, "%zu %llu %llu %llu %llu %llu\n", i / 5, hs[i], hs[i + 1], hs[i + 2], hs[i + 3], hs[i + 4]);
      std::fclose(f);
    }
  }
#endif
  if (chunked && h->pending_chunks[period - 1] == 0) {
    h->pending_chunks[period - 1] = pl.n_chunks;
    h->n_pending++;
    h->key_row_clean[period - 1] = 0;  // holds data now; re-filled when the next sweep starts
  }
  return e;
}

// OPT-IN separable mode (SDPGPU_KERNEL_SEPARABLE, F1 only): one launch per period, see sdp_window.hpp
hipError_t launch_separable(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                            int32_t* pol, const double* pd, const double* pp, bool* too_big) {
  const PeriodInfo& p = h->per[period - 1];
  hipError_t es = flush_pending(h);
  if (es == hipSuccess) {
    sdp::SepParams S{};
    S.x_lo = p.g.x_lo;
    S.step = h->d.step;
    S.h = h->d.holding_cost;
    S.pi = h->d.penalty_cost;
    S.K = h->d.fixed_order_cost;
    S.v = h->d.unit_order_cost;
    S.inv_step = 1.0 / h->d.step;
    S.min_inventory = h->d.min_inventory;
    S.max_inventory = h->d.max_inventory;
    S.clamp_inventory = h->d.clamp_inventory;
    S.n_actions = h->n_actions_full;
    S.n_demand = p.nD;
    S.d_min = h->pmf_d[period - 1].front();  // demands are strictly ascending (checked at set_pmf)
    S.d_range = (int32_t)((h->pmf_d[period - 1].back() - S.d_min) / h->d.step);
    const bool future = period < h->T;
    if (future) {
      S.next_x_lo = h->per[period].g.x_lo;
      S.next_last = (int32_t)(h->per[period].g.nx - 1);
    }
    const int64_t n = p.hi - p.lo;
    if (n > 0) {
      dim3 grid((unsigned)((n + 63) / 64));
      size_t smem = (size_t)(64 + S.n_actions + S.d_range) * 16 + (size_t)(64 + S.n_actions) * 8 +
                    4 * 64 * (sizeof(double) + sizeof(int));
      if (smem > kLdsPerCU) {
        *too_big = true;
        return hipSuccess;
      }
      const bool mx = P.maxdir != 0;
      {
        static LdsMark marks[4];
        hipError_t ea = mx ? (future ? lds_allow(sdp::separable_f1_kernel<true, true>, smem, &marks[0]) : lds_allow(sdp::separable_f1_kernel<true, false>, smem, &marks[1]))
                           : (future ? lds_allow(sdp::separable_f1_kernel<false, true>, smem, &marks[2]) : lds_allow(sdp::separable_f1_kernel<false, false>, smem, &marks[3]));
        if (ea != hipSuccess) return ea;
      }
      if (mx && future) hipLaunchKernelGGL((sdp::separable_f1_kernel<true, true>), grid, dim3(256), smem, h->stream, S, v_next, v_cur, pol, pd, pp, p.lo, p.hi);
      else if (mx) hipLaunchKernelGGL((sdp::separable_f1_kernel<true, false>), grid, dim3(256), smem, h->stream, S, v_next, v_cur, pol, pd, pp, p.lo, p.hi);
      else if (future) hipLaunchKernelGGL((sdp::separable_f1_kernel<false, true>), grid, dim3(256), smem, h->stream, S, v_next, v_cur, pol, pd, pp, p.lo, p.hi);
      else hipLaunchKernelGGL((sdp::separable_f1_kernel<false, false>), grid, dim3(256), smem, h->stream, S, v_next, v_cur, pol, pd, pp, p.lo, p.hi);
      es = hipGetLastError();
    }
  }
  return es;
}

// OPT-IN separable mode, lead-time family (see separable_f2_table_kernel): the table G[q2][y], then its expansion over
// this rank's slab.  Every rank builds the whole table (it is S / nq times smaller than the period).
hipError_t launch_separable_f2(sdpgpu_handle* h, const DevParams& P, int period, const double* v_next, double* v_cur,
                               int32_t* pol, const double* pd, const double* pp) {
  const PeriodInfo& p = h->per[period - 1];
  hipError_t es = flush_pending(h);
  if (es != hipSuccess) return es;
  const bool lead2 = h->d.lead_time == 2;
  const bool future = period < h->T;
  sdp::SepF2Params S{};
  S.y_lo = p.g.x_lo;
  S.step = h->d.step;
  S.inv_step = 1.0 / h->d.step;
  S.h = h->d.holding_cost;
  S.pi = h->d.penalty_cost;
  S.K = h->d.fixed_order_cost;
  S.v = h->d.unit_order_cost;
  S.min_inventory = h->d.min_inventory;
  S.max_inventory = h->d.max_inventory;
  S.clamp_inventory = h->d.clamp_inventory;
  S.lead2 = lead2;
  S.n_actions = h->n_actions_full;
  S.n_demand = p.nD;
  S.cur_nx = (int32_t)p.g.nx;
  S.cur_nq1 = (int32_t)p.g.nq1;
  S.ny = (int32_t)(p.g.nx + p.g.nq1 - 1);
  if (future) {
    const Grid& gn = h->per[period].g;
    S.next_x_lo = gn.x_lo;
    S.next_last = (int32_t)(gn.nx - 1);
    S.next_nx = (int32_t)gn.nx;
    S.next_nq1 = (int32_t)gn.nq1;
  }
  const int64_t nq2 = lead2 ? p.g.nq / p.g.nq1 : 1;
  const size_t need = (size_t)nq2 * (size_t)S.ny;
  if (need > h->sep_elems) {
    es = hipStreamSynchronize(h->stream);  // earlier launches may still read the old table
    if (es != hipSuccess) return es;
    if (h->d_sep_val) (void)hipFree(h->d_sep_val);
    if (h->d_sep_idx) (void)hipFree(h->d_sep_idx);
    h->d_sep_val = nullptr;
    h->d_sep_idx = nullptr;
    h->sep_elems = 0;
    es = hipMalloc((void**)&h->d_sep_val, need * sizeof(double));
    if (es != hipSuccess) return es;
    es = hipMalloc((void**)&h->d_sep_idx, need * sizeof(int32_t));
    if (es != hipSuccess) return es;
    h->sep_elems = need;
  }
  if (nq2 > 65535) return hipErrorInvalidValue;
  dim3 grid((unsigned)((S.ny + 63) / 64), (unsigned)nq2);
  if (future)
    hipLaunchKernelGGL((sdp::separable_f2_table_kernel<true>), grid, dim3(256), 0, h->stream, S, v_next, h->d_sep_val, h->d_sep_idx, pd, pp);
  else
    hipLaunchKernelGGL((sdp::separable_f2_table_kernel<false>), grid, dim3(256), 0, h->stream, S, v_next, h->d_sep_val, h->d_sep_idx, pd, pp);
  es = hipGetLastError();
  if (es != hipSuccess) return es;
  const int64_t n = p.hi - p.lo;
  if (n > 0) {
    if (!grid_ok((n + 255) / 256)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sdp::separable_f2_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, S,
                       h->d_sep_val, h->d_sep_idx, v_cur, pol, p.lo, p.hi);
    es = hipGetLastError();
  }
  return es;
}

}  // namespace sdpgpu_detail

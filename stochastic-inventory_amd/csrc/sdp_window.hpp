// sdp_window.hpp -- LDS-window period kernel for the backorder family F1 (CLSP.java:251-272).
//
// Why a second kernel: in F1 everything a cell needs depends on ONE integer, m = i + k - j
// (state index + action index - demand index):
//     level      l(m) = (x_lo - d_0) + m*step                       (exact integer arithmetic)
//     imm        = (fixed + var)(k) + M(m),   M(m) = h*max(l,0) + pi*max(-l,0)
//     next state = clamp(l(m))  ->  V_{t+1}[clampidx(m)]
// so a workgroup that owns 64 consecutive states and a run of actions touches only a
// contiguous span of 62 + A_chunk + D entries of the pair table W[m] = {M(m), V_{t+1}(clamp m)}.
// The span is staged ONCE in LDS (16 B per entry); HBM/L2 sees each V_{t+1} element once per
// workgroup instead of once per cell.
//
// Thread mapping (wave64): lane = state, each lane carries R consecutive actions in registers
// and walks the demand index j = 0..D-1 serially, in the reference's order
// (Recursion.java:138-144): per cell
//     imm = c0[r] + W.x;  t = p_j*imm;  acc[r] += t;  u = p_j*W.y;  acc[r] += u;      (5 fp64 ops)
// with separate multiplies and adds (no FMA).  Cells (r, j) and (r+1, j+1) read the same W
// entry, so the R-entry register window slides by ONE new ds_read_b128 per demand step:
// LDS traffic is 16 B per R cells.  p_j is wave-uniform and comes through the scalar cache.
// The five VALU ops per cell are the floor for this operation order; the kernel is bound by
// fp64 VALU issue (4 cycles per wave64 instruction per SIMD), not by memory.
//
// Small grids (configs[1] has only 157 state tiles) do not fill 1024 SIMDs with whole-action
// workgroups, so the action range is also cut into NCH chunks across workgroups; each writes a
// partial (value, index) row and a tiny second kernel takes the lexicographic arg-opt over the
// chunks (lowest action index wins ties, exactly the strict-compare scan of Recursion.java:146-157).
#pragma once
#include "sdp_device.hpp"

namespace sdp {

struct WinParams {
  double lev0;   // level value of m = 0: x_lo(cur) - d_0
  double step;
  double h, pi, K, v;
  int32_t idx_off;    // m -> next-grid index offset: (lev0 - x_lo(next)) / step
  int32_t next_last;  // nx(next) - 1
  int32_t n_actions;  // A
  int32_t d_pad;      // demand steps, padded with p = 0 entries to a multiple of R
  int32_t n_chunks;   // action chunks across workgroups
  int32_t chunk_actions;  // actions per chunk (multiple of R)
  int32_t n_tiles;        // state tiles of 64
  int64_t partial_stride; // elements between chunk rows of the partial tables
};

// Deferred combine: when the previous launch (period t+1) left its arg-opt as per-chunk partial rows,
// this launch reads V_{t+1} as the opt over those rows while staging its window, and the workgroups
// of action chunk 0 write the final V_{t+1} / policy rows for their 64 states.  That removes the
// separate combine launch (and its kernel boundary) from every period but the last one computed.
struct FusedPrev {
  const double* part_val;   // [n_chunks][stride], indexed by flat state index
  const int32_t* part_idx;
  double* v_out;            // final V_{t+1}
  int32_t* pol_out;         // final policy of period t+1
  int32_t n_chunks;
  int64_t stride;
#ifdef SDP_STAMPS
  unsigned long long* stamps;
#endif
};

template <bool MAXDIR>
__device__ __forceinline__ double fused_value(const FusedPrev& F, int idx) {
  double v = F.part_val[idx];
  for (int c = 1; c < F.n_chunks; ++c) {
    double o = F.part_val[(int64_t)c * F.stride + idx];
    v = MAXDIR ? (o > v ? o : v) : (o < v ? o : v);
  }
  return v;
}

// W[m] for one m: the immediate-cost part that depends on the level, and the future value.
template <bool FUTURE, bool FUSED, bool MAXDIR>
__device__ __forceinline__ double2 window_entry(const WinParams& W, const double* __restrict__ v_next,
                                                const FusedPrev& F, int m) {
  double l = W.lev0 + (double)m * W.step;
  double hold = W.h * jmax(l, 0.0);
  double pen = W.pi * jmax(-l, 0.0);
  double2 e;
  e.x = hold + pen;  // one of the two is +-0: c0 + e.x == (c0 + hold) + pen bit for bit
  e.y = 0.0;
  if constexpr (FUTURE) {
    // CLSP.java:257-258: upper clamp, then lower clamp.  In the unclamped variant every real cell
    // is inside the next box by construction; the clamp then only keeps padded (p = 0) demand
    // steps, padded actions and tail lanes from reading outside the table.
    int idx = m + W.idx_off;
    idx = idx > W.next_last ? W.next_last : idx;
    idx = idx < 0 ? 0 : idx;
    if constexpr (FUSED)
      e.y = fused_value<MAXDIR>(F, idx);
    else
      e.y = v_next[idx];
  }
  return e;
}

template <int R, bool MAXDIR, bool FUTURE, bool FUSED>
__global__ __launch_bounds__(256) void window_f1_kernel(WinParams W, const double* __restrict__ v_next,
                                                        double* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                        const double* __restrict__ pmf_p, int64_t lo, int64_t hi,
                                                        FusedPrev F) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2* s_win = reinterpret_cast<double2*>(smem);
  const int span = 64 + W.chunk_actions + W.d_pad;  // entries [0, span): one spare slot in front
  double* s_val = reinterpret_cast<double*>(smem + (size_t)span * 16);
  int* s_k = reinterpret_cast<int*>(s_val + 4 * 64);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x / W.n_tiles;
  const int tile = blockIdx.x - chunk * W.n_tiles;
  const int64_t i0 = lo + (int64_t)tile * 64;
  const int kA = chunk * W.chunk_actions;
#ifdef SDP_STAMPS  // diagnostic build only (tools/stamp_window.py): per-wave timeline, never in the product
  unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_t1 = 0;
#endif

  // stage the window: slot s holds m = m_lo + s, m_lo = i0 + kA - d_pad (slot 0 is the spare)
  const int m_lo = (int)i0 + kA - W.d_pad;
  for (int s = tid; s < span; s += 256) s_win[s] = window_entry<FUTURE, FUSED, MAXDIR>(W, v_next, F, m_lo + s);
  if constexpr (FUSED) {
    // finalise period t+1 for this tile's states (same grid in both periods; chunk 0 only)
    const int64_t pidx = i0 + tid;
    if (chunk == 0 && tid < 64 && pidx < hi) {
      double bv = F.part_val[pidx];
      int bk = F.part_idx[pidx];
      for (int c = 1; c < F.n_chunks; ++c) {
        double ov = F.part_val[(int64_t)c * F.stride + pidx];
        int ok = F.part_idx[(int64_t)c * F.stride + pidx];
        if (better<MAXDIR>(ov, ok, bv, bk)) {
          bv = ov;
          bk = ok;
        }
      }
      F.v_out[pidx] = bv;
      F.pol_out[pidx] = bk;
    }
  }
  __syncthreads();

#ifdef SDP_STAMPS
  st_t1 = __builtin_amdgcn_s_memrealtime();
#endif
  double best = MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;
  int bestk = 0;
  const int blocks_in_chunk = W.chunk_actions / R;
  for (int rb = wave; rb < blocks_in_chunk; rb += 4) {
    const int k0 = kA + rb * R;
    if (k0 >= W.n_actions) break;
    double c0[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double a = (double)(k0 + r) * W.step;
      c0[r] = (a > 0 ? W.K : 0.0) + W.v * a;  // fixedCost + variableCost (wave-uniform)
    }
    // slot of (lane, r, j):  lane + (k0 - kA) + r - j + d_pad
    const int base = lane + (k0 - kA) + W.d_pad;
    double2 win[R];
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      win[r] = s_win[base + r];
      acc[r] = 0.0;
    }
    // software pipeline: the probabilities of the NEXT block of R demand steps are fetched (scalar
    // loads) while the current block computes, so no wave ever waits on the scalar cache in the loop
    double pc[R], pn[R];
#pragma unroll
    for (int t = 0; t < R; ++t) pc[t] = pmf_p[t];
#pragma unroll 1
    for (int jb = 0; jb < W.d_pad; jb += R) {
#pragma unroll
      for (int t = 0; t < R; ++t) pn[t] = pmf_p[jb + R + t];  // <= d_pad + R - 1 < D + 8: zero padding
      const double2* nxt = s_win + (base - jb - R);            // slots base-jb-1 ... base-jb-R, ascending
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const double p = pc[t];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const double2 w = win[(r - t + R) % R];
          double imm = c0[r] + w.x;
          acc[r] += p * imm;
          if constexpr (FUTURE) acc[r] += p * w.y;
        }
        // slide: the entry for (r = 0, j + 1) replaces the one (r = R-1, j) just used
        win[(R - 1 - t) % R] = nxt[R - 1 - t];
      }
#pragma unroll
      for (int t = 0; t < R; ++t) pc[t] = pn[t];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = k0 + r;
      if (k < W.n_actions && (MAXDIR ? (acc[r] > best) : (acc[r] < best))) {
        best = acc[r];
        bestk = k;
      }
    }
  }

#ifdef SDP_STAMPS
  if (F.stamps && lane == 0) {
    unsigned long long st_t2 = __builtin_amdgcn_s_memrealtime();
    unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));       // HW_REG_XCC_ID
    unsigned long long* o = F.stamps + ((size_t)blockIdx.x * 4 + wave) * 5;
    o[0] = st_t0; o[1] = st_t1; o[2] = st_t2; o[3] = hwid; o[4] = xcc;
  }
#endif
  s_val[wave * 64 + lane] = best;
  s_k[wave * 64 + lane] = bestk;
  __syncthreads();
  const int64_t idx = i0 + tid;
  if (tid < 64 && idx < hi) {
    double bv = s_val[tid];
    int bk = s_k[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      double ov = s_val[w * 64 + tid];
      int ok = s_k[w * 64 + tid];
      if (better<MAXDIR>(ov, ok, bv, bk)) {
        bv = ov;
        bk = ok;
      }
    }
    const int64_t o = (int64_t)chunk * W.partial_stride + idx;
    out_val[o] = bv;
    out_idx[o] = bk;
  }
}

// ---------------------------------------------------------------------------------------------
// F2 (Leadtime.java:50-81): state (x, preQ), level l = x + preQ - d does not depend on the action;
// the action only selects the PLANE of V_{t+1} (next preQ = action) and the ordering cost:
//     imm = (fixed + var)(k) + M(m),  next = V_{t+1}[k][clamp(m)],   m = ix + iq - j.
// A workgroup owns 64 consecutive x of one preQ row and a chunk of actions; it stages M(m) and,
// per action of the chunk, the row segment V_{t+1}[k][clamp(m)] of 64 + D entries in LDS.  A lane
// (= state) carries R actions in registers; per demand step it reads M once and one V entry per
// action (ds_read_b64, consecutive lanes -> conflict-free).  Same five fp64 ops per cell.
// ---------------------------------------------------------------------------------------------
struct RowParams {
  double lev0;  // level of m = 0: x_lo(cur) - d_0   (preQ enters through m)
  double step;
  double h, pi, K, v;
  int32_t idx_off;      // m -> next-grid inventory index offset
  int32_t next_last;    // nx(next) - 1
  int32_t next_nx;      // nx(next): plane stride of V_{t+1}
  int32_t cur_nx;       // nx(cur)
  int32_t tiles_per_row;
  int32_t n_actions;
  int32_t d_pad;
  int32_t n_chunks;
  int32_t chunk_actions;
  int32_t n_tiles;      // tiles launched (a contiguous run of row tiles)
  int32_t tile0;        // first tile of the run (tile = iq * tiles_per_row + ix / 64)
  int64_t partial_stride;
};

template <int R, bool MAXDIR, bool FUTURE>
__global__ __launch_bounds__(256) void window_f2_kernel(RowParams W, const double* __restrict__ v_next,
                                                        double* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                        const double* __restrict__ pmf_p, int64_t lo, int64_t hi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int span = 64 + W.d_pad + 1;                   // slots per row segment (slot 0 spare)
  double* s_m = reinterpret_cast<double*>(smem);       // M(m)
  double* s_v = s_m + span;                            // [chunk_actions][span]
  double* s_val = s_v + (size_t)(FUTURE ? W.chunk_actions : 0) * span;
  int* s_k = reinterpret_cast<int*>(s_val + 4 * 64);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = blockIdx.x / W.n_tiles;
  const int tile = W.tile0 + (blockIdx.x - chunk * W.n_tiles);
  const int iq = tile / W.tiles_per_row;
  const int ix0 = (tile - iq * W.tiles_per_row) * 64;
  const int kA = chunk * W.chunk_actions;
  const int m_lo = ix0 + iq - W.d_pad;  // slot s <-> m = m_lo + s

  for (int s = tid; s < span; s += 256) {
    double l = W.lev0 + (double)(m_lo + s) * W.step;
    s_m[s] = W.h * jmax(l, 0.0) + W.pi * jmax(-l, 0.0);
  }
  if constexpr (FUTURE) {
    const int total = W.chunk_actions * span;
    for (int e = tid; e < total; e += 256) {
      const int row = e / span;
      const int s = e - row * span;
      int k = kA + row;
      k = k < W.n_actions ? k : W.n_actions - 1;  // padded actions read a valid plane, never selected
      int idx = m_lo + s + W.idx_off;
      idx = idx > W.next_last ? W.next_last : idx;
      idx = idx < 0 ? 0 : idx;
      s_v[e] = v_next[(int64_t)k * W.next_nx + idx];
    }
  }
  __syncthreads();

  double best = MAXDIR ? -1.7976931348623157e308 : 1.7976931348623157e308;
  int bestk = 0;
  const int blocks_in_chunk = W.chunk_actions / R;
  const int base = lane + W.d_pad;  // slot of (lane, j): base - j
  for (int rb = wave; rb < blocks_in_chunk; rb += 4) {
    const int k0 = kA + rb * R;
    if (k0 >= W.n_actions) break;
    double c0[R], acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double a = (double)(k0 + r) * W.step;
      c0[r] = (a > 0 ? W.K : 0.0) + W.v * a;
      acc[r] = 0.0;
    }
    const double* rows = s_v + (size_t)(rb * R) * span + base;
    for (int j = 0; j < W.d_pad; ++j) {
      const double p = pmf_p[j];
      const double mj = s_m[base - j];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double imm = c0[r] + mj;
        acc[r] += p * imm;
        if constexpr (FUTURE) acc[r] += p * rows[r * span - j];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = k0 + r;
      if (k < W.n_actions && (MAXDIR ? (acc[r] > best) : (acc[r] < best))) {
        best = acc[r];
        bestk = k;
      }
    }
  }

  s_val[wave * 64 + lane] = best;
  s_k[wave * 64 + lane] = bestk;
  __syncthreads();
  const int ix = ix0 + tid;
  const int64_t idx = (int64_t)iq * W.cur_nx + ix;
  if (tid < 64 && ix < W.cur_nx && idx >= lo && idx < hi) {
    double bv = s_val[tid];
    int bk = s_k[tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      double ov = s_val[w * 64 + tid];
      int ok = s_k[w * 64 + tid];
      if (better<MAXDIR>(ov, ok, bv, bk)) {
        bv = ov;
        bk = ok;
      }
    }
    const int64_t o = (int64_t)chunk * W.partial_stride + idx;
    out_val[o] = bv;
    out_idx[o] = bk;
  }
}

// arg-opt over the action chunks: rows [c * stride + idx], c = 0..n_chunks-1
template <bool MAXDIR>
__global__ __launch_bounds__(256) void window_combine_kernel(const double* __restrict__ part_val,
                                                             const int32_t* __restrict__ part_idx, int n_chunks,
                                                             int64_t stride, double* __restrict__ v_cur,
                                                             int32_t* __restrict__ pol, int64_t lo, int64_t hi) {
  const int64_t idx = lo + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= hi) return;
  double bv = part_val[idx];
  int bk = part_idx[idx];
  for (int c = 1; c < n_chunks; ++c) {
    double ov = part_val[(int64_t)c * stride + idx];
    int ok = part_idx[(int64_t)c * stride + idx];
    if (better<MAXDIR>(ov, ok, bv, bk)) {
      bv = ov;
      bk = ok;
    }
  }
  v_cur[idx] = bv;
  pol[idx] = bk;
}

}  // namespace sdp

// kernels_rows.h -- "row-per-thread" kernels (v1 layout).  Used for the AR (multinomial) mode,
// whose per-row cost is uniform, and for the variant that writes per-row prior gradients.
//   grid  = persistent blocks of 256 threads, grid-stride over tiles of 1024 contexts
//   tile  = count rows (20 B) and prior rows (40 B) fetched as one flat coalesced stream of
//           16-byte lane loads into LDS; each thread then reads whole rows back (stride 5
//           dwords / 5 doubles is conflict-free: gcd(5, 32) = 1)
//   sums  = per-thread fp64 accumulators -> wave shuffle -> LDS -> one partial per block ->
//           fixed-order finalize kernel (no atomics: bitwise reproducible for a given grid)
#pragma once
#include "bear_common.h"

// ------------------------------------------------------------------ tile staging
// Copies `n_dwords` dwords starting at src (16-byte aligned) into LDS with 16-byte lane
// loads; the (< 4 dword) tail and anything beyond `n_dwords` is handled dword-wise.
__device__ __forceinline__ void stage_dwords(uint32_t *lds, const uint32_t *src, uint32_t n_dwords) {
  const uint32_t n_vec = n_dwords >> 2;
  const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
  uint4 *d4 = reinterpret_cast<uint4 *>(lds);
  for (uint32_t i = threadIdx.x; i < n_vec; i += BEAR_THREADS) d4[i] = s4[i];
  for (uint32_t i = (n_vec << 2) + threadIdx.x; i < n_dwords; i += BEAR_THREADS) lds[i] = src[i];
}

// ------------------------------------------------------------------ row math
// BEAR mode: LL_i and g_b = dLL_i/dalpha_b from counts c[5] and concentrations a[5].
// `tab` != NULL: the table-log form of an item (bear_dm_item_fast: ~150 instead of ~500 instructions, the planned kernels' routine)
// wherever its argument is inside that routine's domain (x > 0 and finite); the library form elsewhere.
__device__ __forceinline__ bear_dp dm_row_item(double x, double c, const double2 *tab) {
  if (tab && x > 0.0 && x < INFINITY) return bear_dm_item_fast(x, c, tab);
  return bear_dm_item(x, c);
}
__device__ __forceinline__ double dm_row(const uint32_t (&c)[5], const double (&a)[5], double (&g)[5], const double2 *tab = nullptr) {
  const double n = (((double)c[0] + (double)c[1]) + ((double)c[2] + (double)c[3])) + (double)c[4];
  double ll = 0.0;
#pragma unroll
  for (int b = 0; b < 5; ++b) g[b] = 0.0;
  if (n == 0.0) return 0.0;
  double A = ((a[0] + a[1]) + (a[2] + a[3])) + a[4];
  bear_dp tn = dm_row_item(A, n, tab);
  ll = -tn.D;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    g[b] = -tn.P;
    if (c[b] != 0) {
      bear_dp tb = dm_row_item(a[b], (double)c[b], tab);
      ll += tb.D;
      g[b] += tb.P;
    }
  }
  return ll;
}

// ------------------------------------------------------------------ mode N: counts + prior rows
template <bool AR, bool GRAD>
__global__ __launch_bounds__(BEAR_THREADS) void dm_prior_kernel(const uint32_t *__restrict__ counts,
                                                                 const double *__restrict__ prior,
                                                                 uint64_t n_rows, bear_params prm,
                                                                 double *__restrict__ grad_prior,
                                                                 const double2 *__restrict__ logtab_g,
                                                                 double *__restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cnt[BEAR_TILE_ROWS * 5];
  __shared__ __attribute__((aligned(16))) double s_pri[BEAR_TILE_ROWS * 5];
  __shared__ double2 s_log[BEAR_LOGTAB_N];
  if (threadIdx.x < BEAR_LOGTAB_N) s_log[threadIdx.x] = logtab_g[threadIdx.x];
  const uint64_t n_tiles = (n_rows + BEAR_TILE_ROWS - 1) / BEAR_TILE_ROWS;
  double acc[2] = {0.0, 0.0};
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * BEAR_TILE_ROWS;
    const uint32_t rows = (uint32_t)((n_rows - row0 < BEAR_TILE_ROWS) ? (n_rows - row0) : BEAR_TILE_ROWS);
    __syncthreads();  // previous tile fully consumed
    stage_dwords(s_cnt, counts + row0 * 5, rows * 5);
    stage_dwords(reinterpret_cast<uint32_t *>(s_pri), reinterpret_cast<const uint32_t *>(prior + row0 * 5),
                 rows * 10);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < BEAR_ROWS_PER_THREAD; ++k) {
      const uint32_t r = threadIdx.x + k * BEAR_THREADS;
      if (r >= rows) break;
      uint32_t c[5];
      double f[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        c[b] = s_cnt[r * 5 + b];
        f[b] = s_pri[r * 5 + b];
      }
      if (AR) {
        // core.py:138-139 with probs = prior + eps (bear_net.py:68)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          double p = f[b] + prm.eps;
          double cb = (double)c[b];
          if (c[b] != 0) acc[0] += cb * (p > 0.0 ? bear_log_tab(p, s_log) : bear_log(p));
          if (GRAD) grad_prior[(row0 + r) * 5 + b] = c[b] != 0 ? cb * bear_rcp(p) : 0.0;
        }
      } else {
        double a[5], g[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) a[b] = __builtin_fma(f[b], prm.inv_h, prm.eps);
        acc[0] += dm_row(c, a, g, s_log);
        double dh = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          dh = __builtin_fma(g[b], f[b], dh);
          if (GRAD) grad_prior[(row0 + r) * 5 + b] = g[b] * prm.inv_h;
        }
        acc[1] -= dh * prm.inv_h;  // d alpha_b / d h_signed = -f_b / h
      }
    }
  }
  block_store_partials<2>(acc, partials);
}

// ------------------------------------------------------------------ mode N on a ROW-WISE plan (bear_plan_create_auto)
// The dense form of the plan: where most of a table's cells are beyond the product path (counts of 1e3 ... 1e5: a k-mer table at
// k = 5, the reference's data/ysd1_lag_5 table) the sorted encoding holds nothing but overflow lists -- 16-byte records and a
// gathered prior cell per item, ~100 B per context -- and the step is bound by those gathers.  Such a plan keeps NOTHING per item:
// the step streams the caller's count rows (20 B) and prior rows (40 B), a context per thread, every cell through the table-log
// form of the Stirling difference (dm_row_item), device-resident parameters and the last block's fixed-order sum as in the
// planned kernels (one launch).  2e7 dense contexts: 1.30 -> ~1.0 ms, with gradient rows 2.76 -> 1.18 ms (round 6).
// (tiles of DPR_TILE_ROWS contexts: the per-cell routine is a dependent chain of ~150 instructions, so what this kernel needs is waves in
// flight -- 30 KB of LDS a block and four blocks = 16 waves per CU; with dm_prior_kernel's 1024-row tiles it was two blocks)
#ifndef DPR_TILE_ROWS
#define DPR_TILE_ROWS 512
#endif
#define DPR_BLOCKS_PER_CU 4
template <bool AR, bool GRAD>
__global__ __launch_bounds__(BEAR_THREADS, DPR_BLOCKS_PER_CU / 4) void dm_prior_rows_kernel(const uint32_t *__restrict__ counts, const double *__restrict__ prior,
                                                                      uint64_t n_rows, bear_params prm_arg, double *__restrict__ grad_prior,
                                                                      const double2 *__restrict__ logtab_g, double *__restrict__ partials,
                                                                      const bear_step_io io) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cnt[DPR_TILE_ROWS * 5];
  __shared__ __attribute__((aligned(16))) double s_pri[DPR_TILE_ROWS * 5];
  __shared__ double2 s_log[BEAR_LOGTAB_N];
  const bear_params prm = bear_params_of(prm_arg, io);
  if (threadIdx.x < BEAR_LOGTAB_N) s_log[threadIdx.x] = logtab_g[threadIdx.x];
  const uint64_t n_tiles = (n_rows + DPR_TILE_ROWS - 1) / DPR_TILE_ROWS;
  double acc[2] = {0.0, 0.0};
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * DPR_TILE_ROWS;
    const uint32_t rows = (uint32_t)((n_rows - row0 < DPR_TILE_ROWS) ? (n_rows - row0) : DPR_TILE_ROWS);
    __syncthreads();  // previous tile fully consumed (and the log table is in place)
    stage_dwords(s_cnt, counts + row0 * 5, rows * 5);
    stage_dwords(reinterpret_cast<uint32_t *>(s_pri), reinterpret_cast<const uint32_t *>(prior + row0 * 5), rows * 10);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < (DPR_TILE_ROWS / BEAR_THREADS); ++k) {
      const uint32_t r = threadIdx.x + k * BEAR_THREADS;
      if (r >= rows) break;
      uint32_t c[5];
      double f[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        c[b] = s_cnt[r * 5 + b];
        f[b] = s_pri[r * 5 + b];
      }
      if (AR) {      // core.py:138-139 with probs = prior + eps (bear_net.py:68)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          const double p = f[b] + prm.eps, cb = (double)c[b];
          if (c[b] != 0) acc[0] += cb * (p > 0.0 ? bear_log_tab(p, s_log) : bear_log(p));
          if (GRAD) grad_prior[(row0 + r) * 5 + b] = c[b] != 0 ? cb * bear_rcp(p) : 0.0;
        }
      } else {
        double a[5], g[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) a[b] = __builtin_fma(f[b], prm.inv_h, prm.eps);
        acc[0] += dm_row(c, a, g, s_log);
        double dh = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          dh = __builtin_fma(g[b], f[b], dh);
          if (GRAD) grad_prior[(row0 + r) * 5 + b] = g[b] * prm.inv_h;
        }
        acc[1] -= dh * prm.inv_h;  // d alpha_b / d h_signed = -f_b / h
      }
    }
  }
  __syncthreads();
  block_finish<2>(acc, partials, io);
}

// ------------------------------------------------------------------ mode R: train + reference counts
template <bool AR>
__global__ __launch_bounds__(BEAR_THREADS) void dm_ref_kernel(const uint32_t *__restrict__ train,
                                                               const uint32_t *__restrict__ ref,
                                                               uint64_t n_rows, bear_params prm,
                                                               const double2 *__restrict__ logtab_g,
                                                               double *__restrict__ partials) {
  __shared__ double2 s_log[BEAR_LOGTAB_N];
  if (threadIdx.x < BEAR_LOGTAB_N) s_log[threadIdx.x] = logtab_g[threadIdx.x];
  __shared__ __attribute__((aligned(16))) uint32_t s_trn[BEAR_TILE_ROWS * 5];
  __shared__ __attribute__((aligned(16))) uint32_t s_ref[BEAR_TILE_ROWS * 5];
  const uint64_t n_tiles = (n_rows + BEAR_TILE_ROWS - 1) / BEAR_TILE_ROWS;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * BEAR_TILE_ROWS;
    const uint32_t rows = (uint32_t)((n_rows - row0 < BEAR_TILE_ROWS) ? (n_rows - row0) : BEAR_TILE_ROWS);
    __syncthreads();
    stage_dwords(s_trn, train + row0 * 5, rows * 5);
    stage_dwords(s_ref, ref + row0 * 5, rows * 5);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < BEAR_ROWS_PER_THREAD; ++k) {
      const uint32_t r = threadIdx.x + k * BEAR_THREADS;
      if (r >= rows) break;
      uint32_t c[5];
      double rr[4];
#pragma unroll
      for (int b = 0; b < 5; ++b) c[b] = s_trn[r * 5 + b];
#pragma unroll
      for (int b = 0; b < 4; ++b) rr[b] = (double)s_ref[r * 5 + b] + prm.eps;  // bear_ref.py:335-337
      // bear_ref.py:30-33: L1-normalise, Jukes-Cantor; bear_ref.py:63-68: mix with the stop net
      const double invR = bear_rcp((rr[0] + rr[1]) + (rr[2] + rr[3]));
      double f[5], dft[5], dfn[5];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double dev = __builtin_fma(rr[b], invR, -0.25);  // norm_b - 1/4
        f[b] = __builtin_fma(prm.E, dev, 0.25) * prm.V;
        dft[b] = -prm.tauE * dev * prm.V;                 // d f_b / d tau_signed
        dfn[b] = -prm.nw * f[b] * prm.V;                  // d f_b / d nu_signed (g_net = 0)
      }
      f[4] = prm.nw * prm.V;
      dft[4] = 0.0;
      dfn[4] = prm.nw * (1.0 - f[4]) * prm.V;
      double dLdf[5];
      if (AR) {
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          double p = f[b] + prm.eps;
          double cb = (double)c[b];
          dLdf[b] = 0.0;
          if (c[b] != 0) {
            acc[0] += cb * (p > 0.0 ? bear_log_tab(p, s_log) : bear_log(p));
            dLdf[b] = cb * bear_rcp(p);
          }
        }
      } else {
        double a[5], g[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) a[b] = __builtin_fma(f[b], prm.inv_h, prm.eps);
        acc[0] += dm_row(c, a, g);
        double dh = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          dLdf[b] = g[b] * prm.inv_h;
          dh = __builtin_fma(dLdf[b], f[b], dh);
        }
        acc[1] -= dh;
      }
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        acc[2] = __builtin_fma(dLdf[b], dft[b], acc[2]);
        acc[3] = __builtin_fma(dLdf[b], dfn[b], acc[3]);
      }
    }
  }
  block_store_partials<4>(acc, partials);
}

// ------------------------------------------------------------------ mode R on the DENSE form of a reference-aware plan
// As dm_prior_rows_kernel: a table of large counts keeps nothing per item (bear_plan_create_ref decides), the step streams the
// training and reference rows (20 + 20 B), a context per thread, the cells through the table-log Stirling difference; device-resident
// parameters, the last block's fixed-order sum and -- bear_ref_train_step_f64 -- the Adam update, as the planned kernels.
template <bool AR>
__global__ __launch_bounds__(BEAR_THREADS) void dm_ref_rows_kernel(const uint32_t *__restrict__ train, const uint32_t *__restrict__ ref,
                                                                    uint64_t n_rows, bear_params prm_arg, const double2 *__restrict__ logtab_g,
                                                                    double *__restrict__ partials, const bear_step_io io,
                                                                    const bear_apply_io apply) {
  __shared__ double2 s_log[BEAR_LOGTAB_N];
  __shared__ __attribute__((aligned(16))) uint32_t s_trn[DPR_TILE_ROWS * 5];
  __shared__ __attribute__((aligned(16))) uint32_t s_ref[DPR_TILE_ROWS * 5];
  const bear_params prm = bear_params_of(prm_arg, io);
  if (threadIdx.x < BEAR_LOGTAB_N) s_log[threadIdx.x] = logtab_g[threadIdx.x];
  const uint64_t n_tiles = (n_rows + DPR_TILE_ROWS - 1) / DPR_TILE_ROWS;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * DPR_TILE_ROWS;
    const uint32_t rows = (uint32_t)((n_rows - row0 < DPR_TILE_ROWS) ? (n_rows - row0) : DPR_TILE_ROWS);
    __syncthreads();
    stage_dwords(s_trn, train + row0 * 5, rows * 5);
    stage_dwords(s_ref, ref + row0 * 5, rows * 5);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < DPR_TILE_ROWS / BEAR_THREADS; ++k) {
      const uint32_t r = threadIdx.x + k * BEAR_THREADS;
      if (r >= rows) break;
      uint32_t c[5];
      double rr[4];
#pragma unroll
      for (int b = 0; b < 5; ++b) c[b] = s_trn[r * 5 + b];
#pragma unroll
      for (int b = 0; b < 4; ++b) rr[b] = (double)s_ref[r * 5 + b] + prm.eps;  // bear_ref.py:335-337
      // bear_ref.py:30-33: L1-normalise, Jukes-Cantor; bear_ref.py:63-68: mix with the stop net
      const double invR = bear_rcp((rr[0] + rr[1]) + (rr[2] + rr[3]));
      double f[5], dft[5], dfn[5];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double dev = __builtin_fma(rr[b], invR, -0.25);  // norm_b - 1/4
        f[b] = __builtin_fma(prm.E, dev, 0.25) * prm.V;
        dft[b] = -prm.tauE * dev * prm.V;                       // d f_b / d tau_signed
        dfn[b] = -prm.nw * f[b] * prm.V;                        // d f_b / d nu_signed (g_net = 0)
      }
      f[4] = prm.nw * prm.V;
      dft[4] = 0.0;
      dfn[4] = prm.nw * (1.0 - f[4]) * prm.V;
      double dLdf[5];
      if (AR) {
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          const double p = f[b] + prm.eps, cb = (double)c[b];
          dLdf[b] = 0.0;
          if (c[b] != 0) {
            acc[0] += cb * (p > 0.0 ? bear_log_tab(p, s_log) : bear_log(p));
            dLdf[b] = cb * bear_rcp(p);
          }
        }
      } else {
        double a[5], g[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) a[b] = __builtin_fma(f[b], prm.inv_h, prm.eps);
        acc[0] += dm_row(c, a, g, s_log);
        double dh = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          dLdf[b] = g[b] * prm.inv_h;
          dh = __builtin_fma(dLdf[b], f[b], dh);
        }
        acc[1] -= dh;
      }
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        acc[2] = __builtin_fma(dLdf[b], dft[b], acc[2]);
        acc[3] = __builtin_fma(dLdf[b], dfn[b], acc[3]);
      }
    }
  }
  __syncthreads();
  block_finish<4>(acc, partials, io, apply);
}

// kernels_evalplan.h -- held-out evaluation (bear_net.py:323-371, bear_ref.py:391-446, one batch of h_scan :465-531) on a
// sorted plan of the TEST column.
//
// The test counts of a resident table are as constant as its training counts, so -- as for the training step
// (kernels_plan.h) -- everything that depends on them alone is done once, at load time (bear_eval_plan_create): per tile of
// EVP_ROWS consecutive contexts two lists, sorted by count,
//     cells   (row, letter) with a non-zero test count c      -> + D(conc_b, c) per DM model, c log(f_b + eps) for the AR model
//     totals  rows with a non-zero total n = sum_b t_b         -> - D(sum_b conc_b, n) per DM model, and the arg-max accuracy
// as uint16 offsets, each list padded to whole 64-lane units.  Rows without test transitions (52 % of the k=13 synthetic
// table) appear in neither list and cost nothing.
//
// Per launch: one 1024-thread block per CU owns a contiguous range of tiles; two of its 16 waves only stream -- by LDS-DMA --
// the next tile's test, training and prior rows and its two lists into the other half of an LDS double buffer (see
// dm_prior_plan_kernel for why dedicated DMA waves); the other 14 draw tickets from an LDS counter, ONE barrier per tile:
//     total units first (lane = one row, sorted by n: the whole row from LDS, for every DM model of the launch the
//       concentrations, -D(A, n) as a wave-uniform product loop + table log, and the arg-max; undecided (row, model) pairs --
//       ties within 17.5 sigma, common on sparse tables because equal counts tie -- go to an LDS list),
//     then cell units (lane = one cell, sorted by c: wave-uniform product loops for all models at once),
//     then, once every total unit of the tile has been retired (an LDS counter, no barrier), the tie list, densely: the
//       tie-breaking noise (core.py:69-71) is evaluated in fp32 on the hardware transcendentals and only repeated in fp64
//       when the two best noisy values are closer than 20x the fp32 error bound.
// Sums: per-thread fp64 accumulators -> wave shuffle -> LDS -> one partial per block -> fixed-order finalize kernel.
#pragma once
#include "kernels_eval.h"
#include "kernels_plan.h"

#define EVP_THREADS 1024
#define EVP_WAVES (EVP_THREADS / 64)
#define EVP_DMA_WAVES 2
#ifndef EVP_ROWS
#define EVP_ROWS 704                              // contexts per tile (multiple of 64; row0 * 20 B stays 16-byte aligned)
#endif
#define EVP_CELL_CAP (EVP_ROWS * 5)               // cells of a tile (multiple of 64)
#define EVP_ITEMS_CAP (EVP_CELL_CAP + EVP_ROWS)   // uint16 entries per tile in the plan (fixed stride)
#define EVP_SENT_CELL (EVP_ROWS * 5)              // neutral cell: test count 0 (padding of the last unit)
#define EVP_SENT_ROW EVP_ROWS                     // neutral row: no test transitions
#define EVP_MAXC 4                                // DM models per launch (register accumulators; more models = more launches)
#define EVP_TIECAP (EVP_ROWS * (EVP_MAXC + 1))    // undecided (row, model) pairs of a tile: every pair fits
#define EVP_SLOT_ARM 15u                          // model slot of the AR model in a tie entry
static_assert(EVP_ROWS % 64 == 0 && (EVP_ROWS * 20) % 16 == 0, "tile geometry");
static_assert(EVP_ROWS * 16 + 15 < 65536, "tie entries are uint16: row * 16 + model slot");

// ---------------------------------------------------------------------------------------------------- plan construction
// One block per tile: counting sort of the tile's cells and totals by min(count, 32) (LDS histogram, rank = the atomic's
// return value), lists written to the tile's fixed-stride slot; tile_info[t] = n_cells | n_totals << 16.
__global__ __launch_bounds__(256) void evp_build_kernel(const uint32_t *__restrict__ test, uint64_t n_rows, uint64_t n_tiles,
                                                        uint16_t *__restrict__ items, uint32_t *__restrict__ tile_info) {
  constexpr int RPT = (EVP_ROWS + 255) / 256;
  __shared__ uint32_t hist[2][34], offs[2][34];
  const uint32_t tid = threadIdx.x;
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t row0 = t * EVP_ROWS;
    const uint32_t rows = (uint32_t)(n_rows - row0 < EVP_ROWS ? n_rows - row0 : EVP_ROWS);
    if (tid < 68) (&hist[0][0])[tid] = 0;
    __syncthreads();
    uint32_t key[RPT][6], rank[RPT][6];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const uint32_t lr = tid + 256u * k;
      uint32_t nsat = 0;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        uint32_t c = 0;
        if (lr < rows) {
          if (b < 5) {
            c = test[(row0 + lr) * 5 + b];
            const uint32_t s = nsat + c;
            nsat = s < nsat ? 0xffffffffu : s;
          } else {
            c = nsat;
          }
        }
        key[k][b] = c > 32u ? 32u : c;
        rank[k][b] = key[k][b] ? atomicAdd(&hist[b == 5][key[k][b]], 1u) : 0u;
      }
    }
    __syncthreads();
    if (tid < 2) {
      uint32_t run = 0;
      for (int q = 1; q <= 32; ++q) {
        offs[tid][q] = run;
        run += hist[tid][q];
      }
      offs[tid][33] = run;
    }
    __syncthreads();
    const uint32_t n_cells = offs[0][33], n_tots = offs[1][33];
    const uint32_t pad_cells = (n_cells + 63u) & ~63u, pad_tots = (n_tots + 63u) & ~63u;
    uint16_t *dst = items + t * (uint64_t)EVP_ITEMS_CAP;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const uint32_t lr = tid + 256u * k;
#pragma unroll
      for (int b = 0; b < 5; ++b)
        if (key[k][b]) dst[offs[0][key[k][b]] + rank[k][b]] = (uint16_t)(lr * 5 + b);
      if (key[k][5]) dst[pad_cells + offs[1][key[k][5]] + rank[k][5]] = (uint16_t)lr;
    }
    for (uint32_t i = n_cells + tid; i < pad_cells; i += 256) dst[i] = (uint16_t)EVP_SENT_CELL;
    for (uint32_t i = n_tots + tid; i < pad_tots; i += 256) dst[pad_cells + i] = (uint16_t)EVP_SENT_ROW;
    if (tid == 0) tile_info[t] = n_cells | (n_tots << 16);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------- per-launch evaluation
struct evp_buf {
  __attribute__((aligned(16))) uint32_t tst[EVP_ROWS * 5 + 8];   // [EVP_SENT_CELL ..] = 0
  __attribute__((aligned(16))) uint32_t trn[EVP_ROWS * 5 + 8];
  __attribute__((aligned(16))) double pri[EVP_ROWS * 5 + 6];     // sentinel row = 1
  __attribute__((aligned(16))) uint16_t items[EVP_ITEMS_CAP];
};
struct evp_lds {
  evp_buf buf[2];
  double2 logtab[BEAR_LOGTAB_N];
  double red[EVP_WAVES][EVS_NOUT];
  uint16_t tie[EVP_TIECAP];
  uint32_t info[2];        // n_cells | n_totals << 16 of the tile in each slot (written by the DMA wave that staged it)
  uint32_t ticket[2], tie_ticket[2], tot_done[2], n_tie[2];
};

// D(x, c) for MC models at once on the product path (1 <= c <= SRT_CL; lanes with c == 0 yield 0), wave-uniform bounds.
template <int MC>
__device__ __forceinline__ void evp_light_D(const double (&x)[MC], uint32_t c, uint32_t cmin, uint32_t cmax, const double2 *logtab,
                                            double (&D)[MC]) {
  double p[MC], t[MC];
#pragma unroll
  for (int i = 0; i < MC; ++i) {
    p[i] = 1.0;
    t[i] = x[i];
  }
  uint32_t j = 0;
  for (; j < cmin; ++j) {
#pragma unroll
    for (int i = 0; i < MC; ++i) {
      p[i] *= t[i];
      t[i] += 1.0;
    }
  }
  for (; j < cmax; ++j) {
    if (j < c) {
#pragma unroll
      for (int i = 0; i < MC; ++i) {
        p[i] *= t[i];
        t[i] += 1.0;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MC; ++i) D[i] = c != 0 ? bear_log_tab(p[i], logtab) : 0.0;
}

// Items outside the product path -- counts above SRT_CL (Stirling form) or concentrations outside (0, 2^30] -- for the lanes
// that have one (rare on k-mer tables; every lane on dense ones): ONE call site in a rolled loop over the models, so the hot
// path keeps its registers.
template <int MC>
__device__ __forceinline__ void evp_general_D(const double (&x)[MC], double c, bool live, bool heavy, int m_cnt, const double2 *logtab,
                                              double (&D)[MC]) {
  bool odd = false;
#pragma unroll
  for (int i = 0; i < MC; ++i) odd |= i < m_cnt && live && (heavy || !(x[i] > 0.0 && x[i] <= SRT_XMAX));
  if (!__builtin_amdgcn_ballot_w64(odd)) return;
#pragma unroll 1
  for (int mi = 0; mi < m_cnt; ++mi) {
    double xx = x[0];
#pragma unroll
    for (int i = 1; i < MC; ++i) xx = i == mi ? x[i] : xx;
    if (live && (heavy || !(xx > 0.0 && xx <= SRT_XMAX))) {
      const double d = srt_general_fast(xx, c, logtab).D;
#pragma unroll
      for (int i = 0; i < MC; ++i) D[i] = i == mi ? d : D[i];
    }
  }
}

template <int MC>
__global__ __launch_bounds__(EVP_THREADS) void eval_plan_kernel(const uint32_t *__restrict__ test, const uint32_t *__restrict__ train,
                                                                 const double *__restrict__ prior, uint64_t n_rows, evl_args A, int m0,
                                                                 int m_cnt, int do_common, const uint16_t *__restrict__ plan_items,
                                                                 const uint32_t *__restrict__ tile_info, uint64_t n_tiles,
                                                                 const double2 *__restrict__ logtab_g, double *__restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  evp_lds &S = *reinterpret_cast<evp_lds *>(srt_smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = srt_uniform(tid >> 6);
  const double eps = A.eps, sig_dm = 100.0 * A.eps;
  double acc_ll[MC], acc_cor[MC], acc_arm = 0.0, acc_carm = 0.0, acc_tot = 0.0;
#pragma unroll
  for (int k = 0; k < MC; ++k) acc_ll[k] = acc_cor[k] = 0.0;

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  for (uint32_t i = tid; i < 2u * (EVP_ROWS * 5 + 8); i += EVP_THREADS) {   // absent columns read as 0 / 1; sentinel rows
    const uint32_t b = i / (EVP_ROWS * 5 + 8), k = i % (EVP_ROWS * 5 + 8);
    if (!A.has_train || k >= EVP_ROWS * 5) S.buf[b].trn[k] = 0u;
    if (k >= EVP_ROWS * 5) S.buf[b].tst[k] = 0u;
  }
  for (uint32_t i = tid; i < 2u * (EVP_ROWS * 5 + 6); i += EVP_THREADS) {
    const uint32_t b = i / (EVP_ROWS * 5 + 6), k = i % (EVP_ROWS * 5 + 6);
    if (!A.has_prior || k >= EVP_ROWS * 5) S.buf[b].pri[k] = 1.0;
  }
  if (tid < 2) S.ticket[tid] = S.tie_ticket[tid] = S.tot_done[tid] = S.n_tie[tid] = S.info[tid] = 0u;

  const bool dma_wave = wave >= EVP_WAVES - EVP_DMA_WAVES;
  const uint32_t dw = wave - (EVP_WAVES - EVP_DMA_WAVES);
  const uint64_t G = gridDim.x;
  const uint64_t first = (n_tiles * (uint64_t)blockIdx.x) / G, count = (n_tiles * ((uint64_t)blockIdx.x + 1)) / G - first;
  const __attribute__((address_space(4))) uint32_t *info_c = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)tile_info;

  // DMA waves: every piece of tile `t` (info word `inf`) into ring slot `b`.  Row slabs are rows * 20 (or 40) bytes: whole
  // 16-byte words by LDS-DMA, the up-to-3 trailing dwords of the table's last tile through the scalar path (a vector load
  // would be followed by s_waitcnt vmcnt(0), which drains the DMA queue).
  auto stage = [&](uint64_t t, uint32_t inf, uint32_t b) {
    const uint64_t row0 = t * EVP_ROWS;
    const uint32_t rows = (uint32_t)(n_rows - row0 < EVP_ROWS ? n_rows - row0 : EVP_ROWS);
    const uint32_t n_cells = inf & 0xffffu, n_tots = inf >> 16;
    const uint32_t ibytes = (((n_cells + 63u) & ~63u) + ((n_tots + 63u) & ~63u)) * 2u;
    const uint32_t cbytes = (rows * 20u) & ~15u, pbytes = rows * 40u;   // rows * 40 is a multiple of 8: & ~15 below
    evp_buf &B = S.buf[b];
    uint32_t pc = dw;   // pieces dealt round-robin over the DMA waves across the slabs
    auto slab = [&](void *lds, const void *src, uint32_t bytes) {
      const uint32_t np = (bytes + 1023u) >> 10;
      for (; pc < np; pc += EVP_DMA_WAVES) pln_dma_piece(lds, src, bytes, pc, lane);
      pc -= np;
    };
    slab(B.items, plan_items + t * (uint64_t)EVP_ITEMS_CAP, ibytes);
    slab(B.tst, test + row0 * 5, cbytes);
    if (A.has_train) slab(B.trn, train + row0 * 5, cbytes);
    if (A.has_prior) slab(B.pri, prior + row0 * 5, pbytes & ~15u);
    if (dw == 0) {
      const uint32_t tail0 = cbytes >> 2, ndw = rows * 5u;
      for (uint32_t q = tail0; q < ndw; ++q) {
        const __attribute__((address_space(4))) uint32_t *tc = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)(test + row0 * 5 + q);
        const uint32_t v = *tc;
        uint32_t w = 0;
        if (A.has_train) {
          const __attribute__((address_space(4))) uint32_t *rc = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)(train + row0 * 5 + q);
          w = *rc;
        }
        if (lane == 0) {
          B.tst[q] = v;
          if (A.has_train) B.trn[q] = w;
        }
      }
      if (A.has_prior && (pbytes & 15u)) {
        const __attribute__((address_space(4))) double *pc8 = (const __attribute__((address_space(4))) double *)(uintptr_t)(prior + row0 * 5 + rows * 5 - 1);
        const double v = *pc8;
        if (lane == 0) B.pri[rows * 5 - 1] = v;
      }
      if (lane == 0) S.info[b] = inf;
    }
  };
  auto load_info = [&](uint64_t j) -> uint32_t { return j < count ? info_c[first + j] : 0u; };

  __syncthreads();
  uint32_t inf_next = 0;
  if (dma_wave) {
    __builtin_amdgcn_s_setprio(3);
    if (count) stage(first, load_info(0), 0);
    inf_next = load_info(1);
  }
  uint32_t slot = 0;
  for (uint64_t j = 0; j < count; ++j) {
    if (dma_wave) srt_wait_dma();   // tile j has landed (and S.info[slot] with it: lgkmcnt is waited inside srt_sync)
    srt_sync();                     // ... and every compute wave is done with tile j - 1
    if (dma_wave) {
      if (j + 1 < count) stage(first + j + 1, inf_next, slot ^ 1u);
      inf_next = load_info(j + 2);
      slot ^= 1u;
      continue;
    }
    const evp_buf &B = S.buf[slot];
    const uint32_t par = slot;
    if (tid == 0) {   // the other parity's counters: last used by tile j - 1, whose readers all passed the barrier above
      S.ticket[par ^ 1u] = 0u;
      S.tie_ticket[par ^ 1u] = 0u;
      S.tot_done[par ^ 1u] = 0u;
      S.n_tie[par ^ 1u] = 0u;
    }
    const uint64_t row0 = (first + j) * EVP_ROWS;
    const uint32_t inf = srt_uniform(S.info[slot]);
    const uint32_t n_cells = inf & 0xffffu, n_tots = inf >> 16;
    const uint32_t n_cu = (n_cells + 63u) >> 6, n_tu = (n_tots + 63u) >> 6;
    const uint32_t tot_base = n_cu * 64u;

    // concentrations of DM model (m0 + mi) for one row's letters
    auto resolve_tie = [&](uint32_t row, uint32_t slot_id) {
      double a[5];
      int im;
      if (slot_id == EVP_SLOT_ARM) {
#pragma unroll
        for (int b = 0; b < 5; ++b) a[b] = B.pri[row * 5 + b] + eps;
        im = evl_argmax_noisy(a, eps, A.seed, EVL_ID_ARM, A.row_base + row0 + row, S.logtab);
      } else {
        const int m = m0 + (int)slot_id;
        const double w = A.inv_h[m];
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          const double r = (double)B.trn[row * 5 + b];
          a[b] = m < A.n_h ? __builtin_fma(B.pri[row * 5 + b], w, r) + eps : (r + w) + eps;
        }
        im = evl_argmax_noisy(a, sig_dm, A.seed, m < A.n_h ? (uint32_t)m : EVL_ID_VAN + (uint32_t)(m - A.n_h),
                              A.row_base + row0 + row, S.logtab);
      }
      const double hit = (double)B.tst[row * 5 + im];
      if (slot_id == EVP_SLOT_ARM) acc_carm += hit;
#pragma unroll
      for (int k = 0; k < MC; ++k)
        if ((uint32_t)k == slot_id) acc_cor[k] += hit;
    };
    auto push_tie = [&](uint32_t row, uint32_t slot_id) { S.tie[atomicAdd(&S.n_tie[par], 1u)] = (uint16_t)(row * 16u + slot_id); };

    for (uint32_t w = pln_ticket(&S.ticket[par], lane); w < n_tu + n_cu; w = pln_ticket(&S.ticket[par], lane)) {
      if (w < n_tu) {
        // ---- a unit of 64 rows with test transitions, largest totals first
        const uint32_t un = n_tu - 1u - w;
        const uint32_t row = B.items[tot_base + un * 64u + lane];
        uint32_t t[5], r[5];
        double f[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          t[b] = B.tst[row * 5 + b];
          r[b] = B.trn[row * 5 + b];
          f[b] = B.pri[row * 5 + b];
        }
        const double n = (((double)t[0] + (double)t[1]) + ((double)t[2] + (double)t[3])) + (double)t[4];
        const bool live = n != 0.0;   // false only for the padding of the last unit
        // wave-uniform loop bounds: the list is sorted by min(n, 32) ascending, padding (n = 0) behind the largest
        const uint32_t occ = (un + 1u) * 64u <= n_tots ? 64u : n_tots - un * 64u;
        const uint32_t nn = n > 32.0 ? 33u : (uint32_t)n;
        const uint32_t nmin = occ == 64u ? (uint32_t)__builtin_amdgcn_readlane((int)nn, 0) : 0u;
        const uint32_t nmax = (uint32_t)__builtin_amdgcn_readlane((int)nn, (int)(occ - 1u));
        const uint32_t lmin = nmin > SRT_CL ? SRT_CL : nmin, lmax = nmax > SRT_CL ? SRT_CL : nmax;
        const bool heavy = n > (double)SRT_CL;
        if (do_common) {
          acc_tot += n;
          if (A.arm && live) {
            double p[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) p[b] = f[b] + eps;
            int im;
            if (evl_argmax_clear(p, eps, im))
              acc_carm += (double)(im == 0 ? t[0] : im == 1 ? t[1] : im == 2 ? t[2] : im == 3 ? t[3] : t[4]);
            else
              push_tie(row, EVP_SLOT_ARM);
          }
        }
        double xt[MC];
#pragma unroll
        for (int mi = 0; mi < MC; ++mi) {
          xt[mi] = 1.0;
          if (mi < m_cnt) {
            const int m = m0 + mi;
            const double w8 = A.inv_h[m];
            double a[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) a[b] = m < A.n_h ? __builtin_fma(f[b], w8, (double)r[b]) + eps : ((double)r[b] + w8) + eps;
            xt[mi] = ((a[0] + a[1]) + (a[2] + a[3])) + a[4];
            if (live) {
              int im;
              if (evl_argmax_clear(a, sig_dm, im))
                acc_cor[mi] += (double)(im == 0 ? t[0] : im == 1 ? t[1] : im == 2 ? t[2] : im == 3 ? t[3] : t[4]);
              else
                push_tie(row, (uint32_t)mi);
            }
          }
        }
        {
          double D[MC];
          evp_light_D<MC>(xt, (live && !heavy) ? nn : 0u, lmin, lmax, S.logtab, D);
          evp_general_D<MC>(xt, n, live, heavy, m_cnt, S.logtab, D);
#pragma unroll
          for (int mi = 0; mi < MC; ++mi)
            if (mi < m_cnt) acc_ll[mi] -= D[mi];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's tie pushes are in LDS before the unit counts as retired
        if (lane == 0) atomicAdd(&S.tot_done[par], 1u);
        continue;
      }
      // ---- a unit of 64 cells, largest counts first
      const uint32_t un = n_cu - 1u - (w - n_tu);
      const uint32_t idx = B.items[un * 64u + lane];
      const uint32_t c = B.tst[idx];
      const double r = (double)B.trn[idx], f = B.pri[idx];
      const uint32_t occ = (un + 1u) * 64u <= n_cells ? 64u : n_cells - un * 64u;
      const uint32_t cc = c > 32u ? 33u : c;
      const uint32_t cmin = occ == 64u ? (uint32_t)__builtin_amdgcn_readlane((int)cc, 0) : 0u;
      const uint32_t cmax = (uint32_t)__builtin_amdgcn_readlane((int)cc, (int)(occ - 1u));
      const uint32_t lmin = cmin > SRT_CL ? SRT_CL : cmin, lmax = cmax > SRT_CL ? SRT_CL : cmax;
      const bool live = c != 0u, heavy = c > SRT_CL;
      if (do_common && A.arm && live) {
        const double p = f + eps;
        acc_arm = __builtin_fma((double)c, p > 0.0 ? bear_log_tab(p, S.logtab) : bear_log(p), acc_arm);
      }
      double x[MC], D[MC];
#pragma unroll
      for (int mi = 0; mi < MC; ++mi) {
        x[mi] = 1.0;
        if (mi < m_cnt) {
          const int m = m0 + mi;
          const double w8 = A.inv_h[m];
          x[mi] = m < A.n_h ? __builtin_fma(f, w8, r) + eps : (r + w8) + eps;
        }
      }
      evp_light_D<MC>(x, (live && !heavy) ? c : 0u, lmin, lmax, S.logtab, D);
      evp_general_D<MC>(x, (double)c, live, heavy, m_cnt, S.logtab, D);
#pragma unroll
      for (int mi = 0; mi < MC; ++mi)
        if (mi < m_cnt) acc_ll[mi] += D[mi];
    }
    // ---- the undecided arg-maxes, once every total unit of the tile is retired (they were drawn first: the wait is short)
    if (n_tu) {
      while (srt_uniform(__hip_atomic_load(&S.tot_done[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < n_tu)
        __builtin_amdgcn_s_sleep(2);
      const uint32_t n_tie = srt_uniform(__hip_atomic_load(&S.n_tie[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      for (uint32_t k = pln_ticket(&S.tie_ticket[par], lane); k * 64u < n_tie; k = pln_ticket(&S.tie_ticket[par], lane)) {
        const uint32_t i = k * 64u + lane;
        if (i < n_tie) {
          const uint32_t e = S.tie[i];
          resolve_tie(e >> 4, e & 15u);
        }
      }
    }
    slot ^= 1u;
  }
  srt_wait_dma();
  static_assert(MC <= EVP_MAXC && EVP_MAXC <= EVS_CHUNK && EVP_SLOT_ARM >= EVP_MAXC, "model slots");
  // ---- block reduction -> compact partial (the layout of eval_sorted_kernel: ll[8], cor[8], ll_arm, cor_arm, total_len)
  double vals[EVS_NOUT];
#pragma unroll
  for (int k = 0; k < EVS_NOUT; ++k) vals[k] = 0.0;
#pragma unroll
  for (int k = 0; k < MC; ++k) {
    vals[k] = acc_ll[k];
    vals[EVS_CHUNK + k] = acc_cor[k];
  }
  vals[2 * EVS_CHUNK] = acc_arm;
  vals[2 * EVS_CHUNK + 1] = acc_carm;
  vals[2 * EVS_CHUNK + 2] = acc_tot;
#pragma unroll
  for (int k = 0; k < EVS_NOUT; ++k) {
    const double v = bear_wave_sum(vals[k]);
    if (lane == 0) S.red[wave][k] = v;
  }
  __syncthreads();
  if (tid < EVS_NOUT) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < EVP_WAVES; ++w) s += S.red[w][tid];
    partials[(size_t)blockIdx.x * EVS_NOUT + tid] = s;
  }
}

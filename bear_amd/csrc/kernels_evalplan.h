// kernels_evalplan.h -- held-out evaluation (bear_net.py:323-371, bear_ref.py:391-446, one batch of h_scan :465-531) on a
// sorted plan of the TEST column.
//
// The test (and training) counts of a resident table are constant, so -- as for the training step (kernels_plan.h) --
// everything that depends on them alone is done once, at load time (bear_eval_plan_create): per tile of EVP_ROWS consecutive
// contexts three lists of uint16 offsets, each padded to whole 64-lane units,
//     cells   (row, letter) with a non-zero test count c, sorted by c  -> + D(conc_b, c) per DM model, c log(f_b + eps) (AR model)
//     totals  rows with a non-zero total n = sum_b t_b, sorted by n    -> - D(sum_b conc_b, n) per DM model, the arg-max accuracy
//     ties    those rows whose two largest TRAINING counts are equal   -> the vanilla models' arg-max is decided by the noise;
//             an entry = row | (which letters tie at the top) << 9, most tied letters first (round 5: the noise of a letter costs
//             ~65 issue slots per model -- a 64-bit hash is eight quarter-rate multiplies -- and only the tied letters' is needed)
// Rows without test transitions (52 % of the k=13 synthetic table) appear in no list and cost nothing.
//
// Why the kernel looks the way it does (measured on MI355X, profiles/r02*): a unit of 64 items is one wave running one serial
// fp64 dependency chain -- ~8 clocks per instruction, 2-4 us per unit -- so throughput is a matter of how many units are in
// flight, not of instruction counts.  A first version kept kernels_plan.h's structure (two-slot ring, ONE workgroup barrier per
// tile, a dynamic tie list resolved after the tile's totals): a tile of 704 rows holds only ~13 units for 10 compute waves, and
// the waves spent 45 % of their time at the barrier or waiting for the slowest total unit (23 Gctx/s).  Hence:
//   * NO barrier in the loop.  A three-slot LDS ring; per slot an arrival counter of the DMA waves (`landed`) and a departure
//     counter of the compute waves (`left`).  A compute wave that finds a tile's tickets exhausted moves on to the next tile
//     at once; a slot is refilled when every compute wave has left it.  Stragglers finish their unit while the others are
//     already one or two tiles ahead.
//   * ALL work static.  The vanilla models' ties depend on the training counts only (integers: the arg-max of r_b + v + eps is
//     tied exactly when the two largest r_b are equal, whatever v), so the plan lists them and they are ordinary tickets; the
//     rare ties of the BEAR / AR models (continuous concentrations) are resolved in place.
//   * short units.  A row's work is split into an "H" unit (AR + BEAR models: needs the prior row) and a "V" unit (vanilla
//     models: integer arg-max, lgamma tables); ties are (row, model) units.
// Per launch: one 1024-thread block per CU (128 registers per lane, no vector spills in the <1,4> instantiation); two of its 16
// waves only stream, by LDS-DMA, the tiles' test / training / prior rows and lists into the ring (see dm_prior_plan_kernel for
// why dedicated DMA waves), the other 14 draw tickets.  (Round 2 ran 768 threads: the kernel then needed 168 registers.)
//
// The two model families cost very different amounts (measured: 854 VALU instructions per total unit when all four models of
// the bench configuration went through the general product path):
//     BEAR models   conc_b = f_b / h + r_b + eps: continuous -> wave-uniform product loop + table log per model;
//     vanilla models conc_b = r_b + v + eps with INTEGER r_b: D(conc_b, c) = T[r_b + c] - T[r_b] with T[k] = lgamma(k + v + eps)
//       from a per-launch LDS table (k < 128; larger counts take the general routine), and for the row total
//       T5[N_r + n] - T5[N_r], T5[k] = lgamma(k + 5 (v + eps)): two LDS reads per model and item.
// The tie-breaking noise (core.py:69-71) is evaluated in fp32 on the hardware transcendentals and only repeated in fp64 when
// the two best noisy values are closer than 20x the fp32 error bound (kernels_eval.h).
// Sums: per-thread fp64 accumulators -> wave shuffle -> LDS -> one partial per block -> fixed-order finalize kernel.
#pragma once
#include "kernels_eval.h"
#include "kernels_plan.h"

#ifndef EVP_THREADS
#define EVP_THREADS 1024   // round 3: the <1,4> instantiation fits 128 registers, so 16 waves (14 compute) fit a CU: -10 % on a compacted batch
#endif
#define EVP_WAVES (EVP_THREADS / 64)
#ifndef EVP_DMA_WAVES
#define EVP_DMA_WAVES 2
#endif
#define EVP_CWAVES (EVP_WAVES - EVP_DMA_WAVES)    // compute waves
// (measured in round 4: a row's vanilla part riding on its H unit -- 0.455 against 0.408 ms: the <1,4> form then spills)
#ifndef EVP_TICKET_PREFETCH
#define EVP_TICKET_PREFETCH 0   // 1: the next ticket is drawn while the current unit runs (measured in round 4, same box, three runs each:
#endif                          // 0.432 against 0.404 ms -- a wave then holds a unit it is not working on while others wait at the tile's end)
#ifndef EVP_NSLOT
#define EVP_NSLOT 3
#endif                                            // LDS ring: a tile being finished, the tile being worked on, a tile landing
#ifndef EVP_ROWS
#define EVP_ROWS 448                              // contexts per tile (multiple of 64; row0 * 20 B stays 16-byte aligned)
#endif
#define EVP_CELL_CAP (EVP_ROWS * 5)               // cells of a tile (multiple of 64)
#define EVP_ITEMS_CAP (EVP_CELL_CAP + 3 * EVP_ROWS)   // uint16 entries per tile in the plan (fixed stride): cells | totals | ties | vrows
#define EVP_SENT_CELL (EVP_ROWS * 5)              // neutral cell: test count 0 (padding of the last unit)
#define EVP_SENT_ROW EVP_ROWS                     // neutral row: no test transitions
#define EVP_MAXH 4                                // BEAR models per launch (register accumulators; more models = more launches)
#define EVP_MAXV 4                                // vanilla models per launch
#define EVP_TABK 2048                             // vanilla models: cells / rows with r + c below this are histogram bins of the plan
#define EVP_SLOT_VAN 4u                           // partial slots: 0..3 BEAR, 4..7 vanilla
// Constants of a plan (uint64 [EVP_NCONST], summed over all tiles at build time).  What a vanilla model adds for a cell or a row
// inside its lgamma tables depends on two INTEGERS of the table alone -- T[r + c] - T[r], T5[N + n] - T5[N] -- so the sum over the
// table is sum_j (how many have r + c = j, minus how many have r = j) T[j]: two EVP_TABK-bin histograms per kind, counted once,
// one lgamma difference per occupied bin, model and launch instead of a unit of work per 64 rows (round 5: the V units were 16 % of the kernel,
// the cells' table reads ~7 %).  Likewise the counts under a UNIQUE largest training count (the vanilla arg-max no noise can move)
// and the total length.  Only cells / rows beyond the bins (r + c >= EVP_TABK: dense tables) are still listed and evaluated per launch.
#define EVP_C_CELL0 0                             // [j]: in-table cells with training count j
#define EVP_C_CELL1 EVP_TABK                      // [j]: ... with training + test count j
#define EVP_C_ROW0 (2 * EVP_TABK)                 // [j]: in-table rows (with test transitions) of training total j
#define EVP_C_ROW1 (3 * EVP_TABK)                 // [j]: ... of training + test total j
#define EVP_C_COR (4 * EVP_TABK)                  // sum of the test counts at the unique largest training count
#define EVP_C_TOT (4 * EVP_TABK + 1)              // sum of all test counts
#define EVP_NCONST (4 * EVP_TABK + 2)
static_assert(EVP_ROWS % 64 == 0 && (EVP_ROWS * 20) % 16 == 0 && EVP_ROWS * 5 < 4096 && EVP_ROWS < 512, "tile geometry / info word / tie entries (row: 9 bits)");
#define EVP_INFO(n_cells, n_tots, n_ties) ((n_cells) | ((n_tots) << 12) | ((n_ties) << 22))

// ---------------------------------------------------------------------------------------------------- plan construction
// One block per tile: counting sort of the tile's cells and totals by min(count, 32) (LDS histogram, rank = the atomic's
// return value), of the tie rows by the number of letters that tie (most first), the rows beyond the vanilla models' tables in
// any order; lists written to the tile's fixed-stride slot; tile_info[t] = {EVP_INFO(...), rows beyond the tables}; the plan
// constants (EVP_C_*) counted in LDS over the block's tiles and added to `consts` (zeroed by the caller) at the end.
__global__ __launch_bounds__(256) void evp_build_kernel(const uint32_t *__restrict__ test, const uint32_t *__restrict__ train,
                                                        uint64_t n_rows, uint64_t n_tiles, uint16_t *__restrict__ items,
                                                        uint2 *__restrict__ tile_info, unsigned long long *__restrict__ consts) {
  constexpr int RPT = (EVP_ROWS + 255) / 256;
  __shared__ uint32_t hist[2][34], offs[2][34], hist_t[4], offs_t[5], n_vrow_s;   // hist_t[q]: tie rows with 5 - q letters at the top
  __shared__ uint32_t pc[4 * EVP_TABK];
  __shared__ unsigned long long red[2][4];
  const uint32_t tid = threadIdx.x;
  unsigned long long cor = 0ull, tot = 0ull;
  for (uint32_t i = tid; i < 4u * EVP_TABK; i += 256) pc[i] = 0u;
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t row0 = t * EVP_ROWS;
    const uint32_t rows = (uint32_t)(n_rows - row0 < EVP_ROWS ? n_rows - row0 : EVP_ROWS);
    if (tid < 68) (&hist[0][0])[tid] = 0;
    if (tid < 4) hist_t[tid] = 0;
    if (tid == 4) n_vrow_s = 0;
    __syncthreads();
    uint32_t key[RPT][6], rank[RPT][6], tie_at[RPT], tie_ent[RPT], vrow_at[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const uint32_t lr = tid + 256u * k;
      uint32_t nsat = 0, tv[5];
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
        if (b < 5) tv[b] = c;
        key[k][b] = c > 32u ? 32u : c;
        rank[k][b] = key[k][b] ? atomicAdd(&hist[b == 5][key[k][b]], 1u) : 0u;
      }
      tie_at[k] = vrow_at[k] = 0xffffffffu;
      tie_ent[k] = 0;
      if (lr < rows && nsat != 0) {   // a row with test transitions
        uint32_t rv[5] = {0u, 0u, 0u, 0u, 0u};
        if (train) {
#pragma unroll
          for (int b = 0; b < 5; ++b) rv[b] = train[(row0 + lr) * 5 + b];
        }
        uint32_t top = 1u, rmax = rv[0];   // which letters hold the largest training count (no training column: all five)
        unsigned long long Nr = rv[0], n = tv[0];
#pragma unroll
        for (int b = 1; b < 5; ++b) {
          top = rv[b] > rmax ? 1u << b : top | (rv[b] == rmax ? 1u << b : 0u);
          rmax = rv[b] > rmax ? rv[b] : rmax;
          Nr += rv[b];
          n += tv[b];
        }
        const uint32_t ntop = (uint32_t)__builtin_popcount(top);
        tot += n;
        if (ntop >= 2u) {              // ... whose largest training counts tie: the noise decides the vanilla arg-max
          tie_at[k] = atomicAdd(&hist_t[5u - ntop], 1u);
          tie_ent[k] = lr | (top << 9) | ((5u - ntop) << 16);
        } else {
          const uint32_t im = (uint32_t)__builtin_ctz(top);
          cor += im == 0 ? tv[0] : im == 1 ? tv[1] : im == 2 ? tv[2] : im == 3 ? tv[3] : tv[4];
        }
        if (Nr + n < (unsigned long long)EVP_TABK) {
          atomicAdd(&pc[EVP_C_ROW0 + (uint32_t)Nr], 1u);
          atomicAdd(&pc[EVP_C_ROW1 + (uint32_t)(Nr + n)], 1u);
        } else {
          vrow_at[k] = atomicAdd(&n_vrow_s, 1u);
        }
#pragma unroll
        for (int b = 0; b < 5; ++b)
          if (tv[b] != 0u && rv[b] < (uint32_t)EVP_TABK && tv[b] < (uint32_t)EVP_TABK - rv[b]) {
            atomicAdd(&pc[EVP_C_CELL0 + rv[b]], 1u);
            atomicAdd(&pc[EVP_C_CELL1 + rv[b] + tv[b]], 1u);
          }
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
    if (tid == 2) {
      uint32_t run = 0;
      for (int q = 0; q < 4; ++q) {
        offs_t[q] = run;
        run += hist_t[q];
      }
      offs_t[4] = run;
    }
    __syncthreads();
    const uint32_t n_cells = offs[0][33], n_tots = offs[1][33], n_ties = offs_t[4], n_vrows = n_vrow_s;
    const uint32_t pad_cells = (n_cells + 63u) & ~63u, pad_tots = (n_tots + 63u) & ~63u, pad_ties = (n_ties + 63u) & ~63u,
                   pad_vrows = (n_vrows + 63u) & ~63u;
    uint16_t *dst = items + t * (uint64_t)EVP_ITEMS_CAP;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const uint32_t lr = tid + 256u * k;
#pragma unroll
      for (int b = 0; b < 5; ++b)
        if (key[k][b]) dst[offs[0][key[k][b]] + rank[k][b]] = (uint16_t)(lr * 5 + b);
      if (key[k][5]) dst[pad_cells + offs[1][key[k][5]] + rank[k][5]] = (uint16_t)lr;
      if (tie_at[k] != 0xffffffffu) dst[pad_cells + pad_tots + offs_t[tie_ent[k] >> 16] + tie_at[k]] = (uint16_t)tie_ent[k];
      if (vrow_at[k] != 0xffffffffu) dst[pad_cells + pad_tots + pad_ties + vrow_at[k]] = (uint16_t)lr;
    }
    for (uint32_t i = n_cells + tid; i < pad_cells; i += 256) dst[i] = (uint16_t)EVP_SENT_CELL;
    for (uint32_t i = n_tots + tid; i < pad_tots; i += 256) dst[pad_cells + i] = (uint16_t)EVP_SENT_ROW;
    for (uint32_t i = n_ties + tid; i < pad_ties; i += 256) dst[pad_cells + pad_tots + i] = (uint16_t)EVP_SENT_ROW;
    for (uint32_t i = n_vrows + tid; i < pad_vrows; i += 256) dst[pad_cells + pad_tots + pad_ties + i] = (uint16_t)EVP_SENT_ROW;
    if (tid == 0) tile_info[t] = make_uint2(EVP_INFO(n_cells, n_tots, n_ties), n_vrows);
    __syncthreads();
  }
  // the block's share of the plan constants
  for (uint32_t i = tid; i < 4u * EVP_TABK; i += 256)
    if (pc[i]) atomicAdd(&consts[i], (unsigned long long)pc[i]);
  for (int off = 32; off > 0; off >>= 1) {
    cor += __shfl_xor(cor, off);
    tot += __shfl_xor(tot, off);
  }
  if ((tid & 63u) == 0u) {
    red[0][tid >> 6] = cor;
    red[1][tid >> 6] = tot;
  }
  __syncthreads();
  if (tid < 2) {
    const unsigned long long v = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
    if (v) atomicAdd(&consts[EVP_C_COR + tid], v);
  }
}

// ---------------------------------------------------------------------------------------------------- per-launch evaluation
struct evp_buf {
  __attribute__((aligned(16))) uint32_t tst[EVP_ROWS * 5 + 8];   // [EVP_SENT_CELL ..] = 0
  __attribute__((aligned(16))) uint32_t trn[EVP_ROWS * 5 + 8];
  __attribute__((aligned(16))) double pri[EVP_ROWS * 5 + 6];     // sentinel row = 1
  __attribute__((aligned(16))) uint16_t items[EVP_ITEMS_CAP];
  __attribute__((aligned(16))) uint32_t rid[EVP_ROWS + 4];       // table row of each row (compacted batches; ties only)
};
struct evp_lds {
  evp_buf buf[EVP_NSLOT];
  double2 logtab[BEAR_LOGTAB_N];
  double red[EVP_WAVES][EVS_NOUT];
  uint32_t info[EVP_NSLOT];     // EVP_INFO of the tile in each slot (written by the DMA wave that staged it)
  uint32_t info_v[EVP_NSLOT];   // ... and its number of rows beyond the vanilla models' tables
  uint32_t ticket[EVP_NSLOT];   // work tickets of the tile in the slot
  uint32_t landed[EVP_NSLOT];   // += 1 by the DMA wave that streamed a tile once it is in LDS: the slot's g-th tile is there at g + 1
  uint32_t left[EVP_NSLOT];     // += 1 by each compute wave that has no more work in the slot's tile: free again at 10 (g + 1)
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

__device__ __forceinline__ uint64_t evp_uniform_u64(uint64_t v) {   // a wave-uniform value back into scalar registers
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ uint32_t evp_peek(const uint32_t *p) {
  return srt_uniform(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

// NH / NV: BEAR / vanilla models this instantiation carries (nh <= NH, nv <= NV of them are live: models h0 .. h0 + nh and
// v0 .. v0 + nv of A; A.inv_h[j] = 1 / h_j for j < n_h, A.inv_h[n_h + k] = van_reg[k]).
template <int NH, int NV>
__global__ __launch_bounds__(EVP_THREADS) void eval_plan_kernel(const uint32_t *__restrict__ test, const uint32_t *__restrict__ train,
                                                                 const double *__restrict__ prior, const uint32_t *__restrict__ row_ids,
                                                                 uint64_t n_rows, evl_args A, int h0,
                                                                 int nh, int v0, int nv, int do_common,
                                                                 const uint16_t *__restrict__ plan_items,
                                                                 const uint2 *__restrict__ tile_info,
                                                                 const unsigned long long *__restrict__ plan_consts, uint64_t n_tiles,
                                                                 const double2 *__restrict__ logtab_g, double *__restrict__ partials
#ifdef EVP_STAMPS
                                                                 , unsigned long long *__restrict__ dbg
#endif
                                                                 ) {
#ifdef EVP_STAMPS   // developer build: per-wave cycle totals by phase (0 wait for a tile, 1 H units, 2 cell units, 3 V units, 4 tie units, 5 ticket draws / rest, 6 DMA issue, 7 DMA wait)
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#define EVP_STAMP(k)                                              \
  {                                                               \
    const unsigned long long now = __builtin_amdgcn_s_memtime();  \
    tph[k] += now - t_prev;                                       \
    t_prev = now;                                                 \
  }
#else
#define EVP_STAMP(k)
#endif
  constexpr int NHA = NH > 0 ? NH : 1, NVA = NV > 0 ? NV : 1;   // array extents (zero-length arrays are not C++)
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  evp_lds &S = *reinterpret_cast<evp_lds *>(srt_smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = srt_uniform(tid >> 6);
  const double eps = A.eps, sig_dm = 100.0 * A.eps;
  double accH_ll[NHA], accH_cor[NHA], accV_ll[NVA], accV_cor[NVA], acc_arm = 0.0, acc_carm = 0.0, acc_tot = 0.0;
#pragma unroll
  for (int k = 0; k < NHA; ++k) accH_ll[k] = accH_cor[k] = 0.0;
#pragma unroll
  for (int k = 0; k < NVA; ++k) accV_ll[k] = accV_cor[k] = 0.0;
  // the launch's model parameters, once (wave-uniform: SGPRs)
  double wh[NHA], ve[NVA];
#pragma unroll
  for (int k = 0; k < NHA; ++k) wh[k] = (NH > 0 && k < nh) ? A.inv_h[h0 + k] : 1.0;
#pragma unroll
  for (int k = 0; k < NVA; ++k) ve[k] = (NV > 0 && k < nv) ? A.inv_h[A.n_h + v0 + k] + eps : 1.0;

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  for (uint32_t i = tid; i < (uint32_t)EVP_NSLOT * (EVP_ROWS * 5 + 8); i += EVP_THREADS) {   // absent columns read as 0 / 1; sentinel rows
    const uint32_t b = i / (EVP_ROWS * 5 + 8), k = i % (EVP_ROWS * 5 + 8);
    if (!A.has_train || k >= EVP_ROWS * 5) S.buf[b].trn[k] = 0u;
    if (k >= EVP_ROWS * 5) S.buf[b].tst[k] = 0u;
  }
  for (uint32_t i = tid; i < (uint32_t)EVP_NSLOT * (EVP_ROWS * 5 + 6); i += EVP_THREADS) {
    const uint32_t b = i / (EVP_ROWS * 5 + 6), k = i % (EVP_ROWS * 5 + 6);
    if (!A.has_prior || k >= EVP_ROWS * 5) S.buf[b].pri[k] = 1.0;
  }
  if (tid < EVP_NSLOT) S.ticket[tid] = S.landed[tid] = S.left[tid] = S.info[tid] = S.info_v[tid] = 0u;

  const bool dma_wave = wave >= EVP_CWAVES;
  const uint32_t dw = wave - EVP_CWAVES;
  const uint64_t G = gridDim.x;
  const uint64_t first = (n_tiles * (uint64_t)blockIdx.x) / G, count = (n_tiles * ((uint64_t)blockIdx.x + 1)) / G - first;
  __syncthreads();

  if (dma_wave) {
    // ================================================================================================ the two streaming waves
    // Tile j of this block's range goes to slot j % 3 once every compute wave has left tile j - 3.  The two waves take
    // alternate tiles, each wave all pieces of its tile: a tile is published the moment it has landed, whether or not the slot
    // of the next one is free yet -- so the compute waves may run up to two tiles ahead of the slowest of them -- and up to two
    // tiles (~80 KB) are in flight per CU.
    __builtin_amdgcn_s_setprio(3);
    const __attribute__((address_space(4))) unsigned long long *info_c =      // (a uint2 per tile, read as one scalar 64-bit word)
        (const __attribute__((address_space(4))) unsigned long long *)(uintptr_t)tile_info;
    unsigned long long inf2 = dw < count ? info_c[first + dw] : 0ull;
    for (uint64_t j = dw; j < count; j += EVP_DMA_WAVES) {
      const uint32_t b = (uint32_t)(j % EVP_NSLOT), gen = (uint32_t)(j / EVP_NSLOT);
      const unsigned long long inf_next = j + EVP_DMA_WAVES < count ? info_c[first + j + EVP_DMA_WAVES] : 0ull;   // used one iteration later
      const uint32_t inf = (uint32_t)inf2, n_vr = (uint32_t)(inf2 >> 32);
      EVP_STAMP(5)
      while (evp_peek(&S.left[b]) < (uint32_t)EVP_CWAVES * gen) __builtin_amdgcn_s_sleep(1);
      EVP_STAMP(0)
      const uint64_t t = first + j, row0 = t * EVP_ROWS;
      const uint32_t rows = (uint32_t)(n_rows - row0 < EVP_ROWS ? n_rows - row0 : EVP_ROWS);
      const uint32_t n_cells = inf & 0xfffu, n_tots = (inf >> 12) & 0x3ffu, n_ties = inf >> 22;
      const uint32_t ibytes = (((n_cells + 63u) & ~63u) + ((n_tots + 63u) & ~63u) + ((n_ties + 63u) & ~63u) + ((n_vr + 63u) & ~63u)) * 2u;
      const uint32_t cbytes = (rows * 20u) & ~15u, pbytes = rows * 40u;
      evp_buf &B = S.buf[b];
      auto slab = [&](void *lds, const void *src, uint32_t bytes) {
        for (uint32_t pc = 0; (pc << 10) < bytes; ++pc) pln_dma_piece(lds, src, bytes, pc, lane);
      };
      slab(B.items, plan_items + t * (uint64_t)EVP_ITEMS_CAP, ibytes);
      slab(B.tst, test + row0 * 5, cbytes);
      if (A.has_train) slab(B.trn, train + row0 * 5, cbytes);
      if (A.has_prior) slab(B.pri, prior + row0 * 5, pbytes & ~15u);
      if (A.has_rid) slab(B.rid, row_ids + row0, (rows * 4u) & ~15u);
      {
        // the up-to-3 trailing dwords of the table's last tile through the scalar path (a vector load would be followed by
        // s_waitcnt vmcnt(0) at its use)
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
        if (A.has_rid) {
          for (uint32_t q = rows & ~3u; q < rows; ++q) {
            const __attribute__((address_space(4))) uint32_t *ic = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)(row_ids + row0 + q);
            const uint32_t v = *ic;
            if (lane == 0) B.rid[q] = v;
          }
        }
        if (A.has_prior && (pbytes & 15u)) {
          const __attribute__((address_space(4))) double *pc8 = (const __attribute__((address_space(4))) double *)(uintptr_t)(prior + row0 * 5 + rows * 5 - 1);
          const double v = *pc8;
          if (lane == 0) B.pri[rows * 5 - 1] = v;
        }
        if (lane == 0) {
          S.info[b] = inf;
          S.info_v[b] = n_vr;
          S.ticket[b] = 0u;
        }
      }
      EVP_STAMP(6)
      srt_wait_dma();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(&S.landed[b], 1u);
      EVP_STAMP(7)
      inf2 = inf_next;
    }
  } else {
    // ================================================================================================ the ten compute waves
    for (uint64_t j = 0; j < count; ++j) {
      const uint32_t b = (uint32_t)(j % EVP_NSLOT), gen = (uint32_t)(j / EVP_NSLOT);
      EVP_STAMP(5)
      while (evp_peek(&S.landed[b]) < gen + 1u) __builtin_amdgcn_s_sleep(1);
      EVP_STAMP(0)
      const evp_buf &B = S.buf[b];
      const uint64_t row0 = (first + j) * EVP_ROWS;
      const uint32_t inf = srt_uniform(S.info[b]);
      const uint32_t n_cells = inf & 0xfffu, n_tots = (inf >> 12) & 0x3ffu, n_ties = inf >> 22;
      const uint32_t n_cu = (n_cells + 63u) >> 6, n_tu = (n_tots + 63u) >> 6, n_ku = (n_ties + 63u) >> 6;
      const uint32_t n_vru = (srt_uniform(S.info_v[b]) + 63u) >> 6;
      const uint32_t tot_base = n_cu * 64u, tie_base = tot_base + n_tu * 64u, vr_base = tie_base + n_ku * 64u;
      // work list, dearest first: tie units (64 tied rows x ONE vanilla model: up to five noise draws per lane, the longest
      // units -- at the front, so that a tile's slot is not held for them), H units (AR + BEAR models of the rows with
      // transitions), V units (vanilla models of the rows BEYOND their bins: the others are in the plan's histograms), cell units
      const uint32_t n_hu = (NH > 0 || (do_common && A.arm)) ? n_tu : 0u, n_ktu = NV > 0 ? n_ku * (uint32_t)nv : 0u;
      const uint32_t n_vu = NV > 0 ? n_vru : 0u;
      const uint32_t w_t = n_ktu, w_h = w_t + n_hu, w_v = w_h + n_vu;
      // (a launch with neither BEAR nor AR model needs the cells beyond the tables only: every cell unit is looked at, cheaply)
      const uint32_t n_work = w_v + n_cu;
#if EVP_TICKET_PREFETCH   // the next ticket is drawn while the current unit runs (measured slower: see the switch above)
      uint32_t w_next = pln_ticket_issue(&S.ticket[b], lane);
      for (;;) {
        const uint32_t w = srt_uniform(w_next);
        if (w >= n_work) break;
        w_next = pln_ticket_issue(&S.ticket[b], lane);
#else
      // (every unit drawn: with a wave's first unit of a tile DEALT, as in the training kernels, a wave that has fallen behind has
      // work waiting on every tile and never catches up -- there is no barrier here that would hide it: 0.364 -> 0.387 ms, h_scan 1.94 -> 2.49)
      PLN_FOR_UNITS(w, &S.ticket[b], n_work, wave, EVP_CWAVES) {      // (compute waves only)
#endif
        if (w < w_t) {
          // ---- tie unit: 64 rows whose largest training counts tie, ONE vanilla model: the noise decides among the TIED letters
          //      (the others are a whole count below the top: 17.5 sigma = 1750 eps cannot bridge that, bear_eval_plan_f64 checks).
          //      The plan lists the rows with the most tied letters first, so the rounds of a unit are (nearly) wave-uniform.
          const uint32_t km = w / n_ku, un = w - km * n_ku;            // model of the launch, unit of the list
          const uint32_t ent = B.items[tie_base + un * 64u + lane];
          const uint32_t row = ent & 511u, top_mask = ent >> 9;      // (padding: EVP_SENT_ROW with no letter)
          const uint64_t grow = A.row_base + (A.has_rid ? (uint64_t)B.rid[row] : row0 + row);
          const uint64_t noise_base = evp_uniform_u64(mix64(A.seed + (uint64_t)(EVL_ID_VAN + (uint32_t)v0 + km)));
          float v1 = -INFINITY, v2 = -INFINITY;
          uint32_t i1 = 0u;
          bool unsure = false;
          uint32_t left_mask = top_mask;
          while (__builtin_amdgcn_ballot_w64(left_mask != 0u)) {
            const bool act = left_mask != 0u;
            const uint32_t b = act ? (uint32_t)__builtin_ctz(left_mask) : 0u;     // letters in ascending order: the first of equal values wins, as argmax does
            left_mask &= left_mask - 1u;
            bool tiny;
            const float z = evl_gauss_f32(evl_key(noise_base, grow * 5 + b), &tiny);
            if (act) {
              const bool gt = z > v1;
              v2 = gt ? v1 : (z > v2 ? z : v2);
              i1 = gt ? b : i1;
              v1 = gt ? z : v1;
              unsure |= tiny;
            }
          }
          // (the fp32 noise decides unless the two best are within 20x its error bound: ~0.05 % of the rows repeat in fp64)
          unsure = top_mask != 0u && (unsure || !(v1 - v2 > 2e-4f));
          if (__builtin_amdgcn_ballot_w64(unsure)) {
            if (unsure) {
              const double vk = A.inv_h[A.n_h + v0 + (int)km];
              double a[5], top = -INFINITY;
#pragma unroll
              for (int q = 0; q < 5; ++q) {
                a[q] = ((double)B.trn[row * 5 + q] + vk) + eps;
                top = a[q] > top ? a[q] : top;
              }
              i1 = (uint32_t)evl_argmax_exact(a[0], a[1], a[2], a[3], a[4], top, sig_dm, noise_base, grow, S.logtab);
            }
          }
          if (top_mask != 0u) {
            const double hit = (double)B.tst[row * 5 + i1];
#pragma unroll
            for (int k = 0; k < NV; ++k)
              if ((uint32_t)k == km) accV_cor[k] += hit;
          }
          EVP_STAMP(4)
          continue;
        }
        if (w < w_h) {
          // ---- H unit: 64 rows with test transitions, largest totals first: the AR model's and the BEAR models' arg-max,
          //      the BEAR models' -D(A, n)
          const uint32_t un = n_tu - 1u - (w - w_t);
          const uint32_t row = B.items[tot_base + un * 64u + lane];
          uint32_t t[5], r[5];
          double f[5];
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            t[q] = B.tst[row * 5 + q];
            r[q] = B.trn[row * 5 + q];
            f[q] = B.pri[row * 5 + q];
          }
          const double n = (((double)t[0] + (double)t[1]) + ((double)t[2] + (double)t[3])) + (double)t[4];
          const bool live = n != 0.0;   // false only for the padding of the last unit
          uint32_t undecided = 0;       // bit k: BEAR model k, bit NH: the AR model
          if (do_common && A.arm && live) {
            double p[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) p[q] = f[q] + eps;
            int im;
            if (evl_argmax_clear(p, eps, im))
              acc_carm += (double)(im == 0 ? t[0] : im == 1 ? t[1] : im == 2 ? t[2] : im == 3 ? t[3] : t[4]);
            else
              undecided |= 1u << NH;
          }
          if (NH > 0) {
            // wave-uniform loop bounds: the list is sorted by min(n, 32) ascending, padding (n = 0) behind the largest
            const uint32_t occ = (un + 1u) * 64u <= n_tots ? 64u : n_tots - un * 64u;
            const uint32_t nn = n > 32.0 ? 33u : (uint32_t)n;
            const bool heavy = n > (double)SRT_CL;
            const uint32_t nmin = occ == 64u ? (uint32_t)__builtin_amdgcn_readlane((int)nn, 0) : 0u;
            const uint32_t nmax = (uint32_t)__builtin_amdgcn_readlane((int)nn, (int)(occ - 1u));
            const uint32_t lmin = nmin > SRT_CL ? SRT_CL : nmin, lmax = nmax > SRT_CL ? SRT_CL : nmax;
            const double Nr = (((double)r[0] + (double)r[1]) + ((double)r[2] + (double)r[3])) + (double)r[4];
            const double Sf = ((f[0] + f[1]) + (f[2] + f[3])) + f[4];
            double xt[NHA], D[NHA];
#pragma unroll
            for (int mi = 0; mi < NH; ++mi) {
              xt[mi] = 1.0;
              if (mi < nh) {
                xt[mi] = __builtin_fma(Sf, wh[mi], Nr) + 5.0 * eps;
                if (live) {
                  double a[5];
#pragma unroll
                  for (int q = 0; q < 5; ++q) a[q] = __builtin_fma(f[q], wh[mi], (double)r[q]) + eps;
                  int im;
                  if (evl_argmax_clear(a, sig_dm, im))
                    accH_cor[mi] += (double)(im == 0 ? t[0] : im == 1 ? t[1] : im == 2 ? t[2] : im == 3 ? t[3] : t[4]);
                  else
                    undecided |= 1u << mi;
                }
              }
            }
            evp_light_D<NHA>(xt, (live && !heavy) ? nn : 0u, lmin, lmax, S.logtab, D);
            evp_general_D<NHA>(xt, n, live, heavy, nh, S.logtab, D);
#pragma unroll
            for (int mi = 0; mi < NH; ++mi)
              if (mi < nh) accH_ll[mi] -= D[mi];
          }
          // the rare ties of continuous concentrations (two letters within 17.5 sigma), in place, one model at a time
          if (__builtin_amdgcn_ballot_w64(undecided != 0u)) {
#pragma unroll 1
            for (int sl = 0; sl <= NH; ++sl) {
              if (!__builtin_amdgcn_ballot_w64((undecided >> sl) & 1u)) continue;
              const bool is_arm = sl == NH;
              const double wsl = is_arm ? 0.0 : A.inv_h[h0 + sl];
              if ((undecided >> sl) & 1u) {
                double a[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) a[q] = is_arm ? f[q] + eps : __builtin_fma(f[q], wsl, (double)r[q]) + eps;
                const int im = evl_argmax_noisy(a, is_arm ? eps : sig_dm, A.seed, is_arm ? EVL_ID_ARM : (uint32_t)(h0 + sl),
                                                A.row_base + (A.has_rid ? (uint64_t)B.rid[row] : row0 + row), S.logtab);
                const double hit = (double)(im == 0 ? t[0] : im == 1 ? t[1] : im == 2 ? t[2] : im == 3 ? t[3] : t[4]);
                if (is_arm) acc_carm += hit;
#pragma unroll
                for (int k = 0; k < NH; ++k)
                  if (k == sl) accH_cor[k] += hit;
              }
            }
          }
          EVP_STAMP(1)
          continue;
        }
        if (w < w_v) {
          // ---- V unit: 64 rows with test transitions whose totals leave the vanilla models' tables (N_r + n >= EVP_TABK; every
          //      row of a dense table, hardly any of a k-mer table): -D(N_r + 5 (v + eps), n) by the general routine.  Their
          //      arg-max is in the plan's constants (unique largest training count) or a tie unit's.
          const uint32_t row = B.items[vr_base + (w - w_h) * 64u + lane];
          double n = 0.0, Nr = 0.0;
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            n += (double)B.tst[row * 5 + q];
            Nr += (double)B.trn[row * 5 + q];
          }
          if (NV > 0 && row != (uint32_t)EVP_SENT_ROW) {
#pragma unroll 1
            for (int k = 0; k < nv; ++k) {
              const double d = srt_general_fast(Nr + 5.0 * (A.inv_h[A.n_h + v0 + k] + eps), n, S.logtab).D;
#pragma unroll
              for (int q = 0; q < NV; ++q)
                if (q == k) accV_ll[q] -= d;
            }
          }
          EVP_STAMP(3)
          continue;
        }
        // ---- a unit of 64 cells, largest counts first
        const uint32_t un = n_cu - 1u - (w - w_v);
        const uint32_t idx = B.items[un * 64u + lane];
        const uint32_t c = B.tst[idx], ru = B.trn[idx];
        const double f = B.pri[idx];
        const bool live = c != 0u, heavy = c > SRT_CL;
        if (do_common && A.arm && live) acc_arm = __builtin_fma((double)c, evl_log_any(f + eps, S.logtab), acc_arm);
        if (NV > 0) {      // (cells inside the tables: the plan's histograms)
          const bool in_tab = ru < (uint32_t)EVP_TABK && c < (uint32_t)EVP_TABK - ru;
          if (__builtin_amdgcn_ballot_w64(live && !in_tab)) {
#pragma unroll 1
            for (int k = 0; k < nv; ++k) {
              if (live && !in_tab) {
                const double d = srt_general_fast(((double)ru + A.inv_h[A.n_h + v0 + k]) + eps, (double)c, S.logtab).D;
#pragma unroll
                for (int q = 0; q < NV; ++q)
                  if (q == k) accV_ll[q] += d;
              }
            }
          }
        }
        if (NH > 0) {
          const uint32_t occ = (un + 1u) * 64u <= n_cells ? 64u : n_cells - un * 64u;
          const uint32_t cc = c > 32u ? 33u : c;
          const uint32_t cmin = occ == 64u ? (uint32_t)__builtin_amdgcn_readlane((int)cc, 0) : 0u;
          const uint32_t cmax = (uint32_t)__builtin_amdgcn_readlane((int)cc, (int)(occ - 1u));
          const uint32_t lmin = cmin > SRT_CL ? SRT_CL : cmin, lmax = cmax > SRT_CL ? SRT_CL : cmax;
          const double r = (double)ru;
          double x[NHA], D[NHA];
#pragma unroll
          for (int mi = 0; mi < NH; ++mi) x[mi] = mi < nh ? __builtin_fma(f, wh[mi], r) + eps : 1.0;
          evp_light_D<NHA>(x, (live && !heavy) ? c : 0u, lmin, lmax, S.logtab, D);
          evp_general_D<NHA>(x, (double)c, live, heavy, nh, S.logtab, D);
#pragma unroll
          for (int mi = 0; mi < NH; ++mi)
            if (mi < nh) accH_ll[mi] += D[mi];
        }
        EVP_STAMP(2)
      }
      EVP_STAMP(5)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot are done before it counts as gone
      if (lane == 0) atomicAdd(&S.left[b], 1u);
    }
  }
#ifdef EVP_STAMPS
  if (lane == 0 && dbg)
    for (int k = 0; k < 8; ++k) dbg[((size_t)blockIdx.x * EVP_WAVES + wave) * 8 + k] = tph[k];
#endif
  (void)ve;
  // ---- what the plan knows of the table (EVP_C_*): the vanilla models' terms inside their bins -- one lgamma difference per
  //      occupied bin, kind (cells / row totals) and model, dealt over the launch's threads --, their decided arg-max counts, the
  //      total length
  if (plan_consts) {     // (an empty plan has none)
    if (NV > 0) {
      const uint32_t n_terms = 2u * (uint32_t)nv * EVP_TABK;
      for (uint32_t i = blockIdx.x * EVP_THREADS + tid; i < n_terms; i += gridDim.x * EVP_THREADS) {
        const uint32_t j = i % EVP_TABK, km = (i / EVP_TABK) % (uint32_t)nv, rows_kind = i / (EVP_TABK * (uint32_t)nv);
        const double d = (double)plan_consts[(rows_kind ? EVP_C_ROW1 : EVP_C_CELL1) + j] - (double)plan_consts[(rows_kind ? EVP_C_ROW0 : EVP_C_CELL0) + j];
        if (j != 0u && d != 0.0) {
          const double v1 = A.inv_h[A.n_h + v0 + (int)km] + eps;
          const double T = srt_general_fast(rows_kind ? 5.0 * v1 : v1, (double)j, S.logtab).D;    // lgamma(j + x) - lgamma(x)
          const double term = rows_kind ? -(d * T) : d * T;
#pragma unroll
          for (int k = 0; k < NV; ++k)
            if ((uint32_t)k == km) accV_ll[k] += term;
        }
      }
    }
    if (blockIdx.x == 0 && tid == 0) {
      const double hits = (double)plan_consts[EVP_C_COR];
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if (k < nv) accV_cor[k] += hits;
      if (do_common) acc_tot += (double)plan_consts[EVP_C_TOT];
    }
  }
  static_assert(NH <= EVP_MAXH && NV <= EVP_MAXV && EVP_SLOT_VAN >= EVP_MAXH && EVP_SLOT_VAN + EVP_MAXV <= EVS_CHUNK, "model slots");
  // ---- block reduction -> compact partial: ll[8] = {BEAR 0..3, vanilla 0..3}, cor[8] likewise, ll_arm, cor_arm, total_len
  double vals[EVS_NOUT];
#pragma unroll
  for (int k = 0; k < EVS_NOUT; ++k) vals[k] = 0.0;
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    vals[k] = accH_ll[k];
    vals[EVS_CHUNK + k] = accH_cor[k];
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    vals[EVP_SLOT_VAN + k] = accV_ll[k];
    vals[EVS_CHUNK + EVP_SLOT_VAN + k] = accV_cor[k];
  }
  vals[2 * EVS_CHUNK] = acc_arm;
  vals[2 * EVS_CHUNK + 1] = acc_carm;
  vals[2 * EVS_CHUNK + 2] = acc_tot;
  __syncthreads();
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

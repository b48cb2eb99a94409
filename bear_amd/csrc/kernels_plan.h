// kernels_plan.h -- "planned" DM-marginal + gradient kernels: the per-step hot path for a count
// table that stays resident in HBM across optimizer steps (every epoch of bear_net.train /
// bear_ref.train re-reads the same cached count table, bear_model/dataloader.py:47-48).
//
// The counting sort of kernels_sorted.h depends on the counts only, and the counts never change
// between steps -- only h, the prior rows and (tau, nu) do.  So the sort moves to load time:
//
//   plan (built once per table, bear_plan_create):
//     * per tile of PLN_TILE contexts: the work items of the product path (column b of context r
//       with 1 <= c <= SRT_CL), as uint16 flat offsets r*5+b sorted by ascending c and padded
//       to a multiple of 64 with a sentinel; ~2.4 B per context on k-mer tables
//     * global lists of the rare items that take the Stirling path (c > SRT_CL): column items
//       {flat offset, c} and contexts {row, n}
//   step  (bear_dm_prior_plan_f64 / bear_dm_ref_plan_f64), per tile, ONE barrier:
//     0  the next tile's count rows, prior rows and item list stream into the other half of a
//        double buffer by LDS-DMA while the current tile is evaluated
//     A  one thread per context: n = sum c, S = sum prior; the context term -D(A, n) comes from
//        a per-block table whenever A is shared (S = 1 to 2 ulp; always in mode R)
//     D  units of 64 x SRT_ILP sorted items: p = prod (x+j), p' by the product rule, one table
//        log and one reciprocal per item; all lanes busy, loops wave-uniform
//     after the tiles: the global Stirling-path lists, densely packed over all threads.
#pragma once
#include "kernels_sorted.h"

#ifndef PLN_TILE
#define PLN_TILE 1024                        // contexts per tile (measured: 1024 x 1 block beats 512 x 2 blocks)
#endif
#ifndef PLN_BLOCKS_PER_CU
#define PLN_BLOCKS_PER_CU 1                  // LDS-limited residency the grid is sized for
#endif
#ifndef PLN_THREADS
#define PLN_THREADS 1024                     // evaluation kernels: 16 waves drawing work tickets
#endif
#define PLN_WAVES (PLN_THREADS / 64)
#ifndef PLN_ILP
#define PLN_ILP 1                            // items per lane per unit in the planned kernels
#endif
#define PLN_WAVES_PER_SIMD (PLN_BLOCKS_PER_CU * PLN_WAVES / 4)
#define PLN_BUILD_THREADS PLN_TILE           // plan construction: one context per thread
#define PLN_SENTINEL (PLN_TILE * 5)          // flat offset of the neutral cell (c = 0, prior = 1)
#define PLN_ITEMS_MAX (PLN_TILE * 5 + 64)    // padded light list of one tile, worst case

struct pln_tile_info {
  uint32_t off16;    // start of the tile's item list in the plan's item array, in 16-byte units
  uint32_t n_light;  // real (unpadded) number of items
};

struct pln_heavy_col {
  uint64_t off;  // flat offset row*5+b into the [N,5] arrays
  uint64_t c;
};

struct pln_heavy_row {
  uint64_t row;
  double n;  // exact row total (may exceed 2^32)
};

// ---------------------------------------------------------------------------------------------
// plan construction
// ---------------------------------------------------------------------------------------------
// Pass 1: per tile, number of product-path items in columns [0, ncol); global counts of heavy
// column items, heavy contexts (n > SRT_CL) and, for ncol == 4, heavy stop counts.
__global__ __launch_bounds__(PLN_BUILD_THREADS) void plan_count_kernel(const uint32_t *__restrict__ counts, uint64_t n_rows,
                                                                  int ncol, uint32_t *__restrict__ n_light,
                                                                  unsigned long long *__restrict__ heavy_counts,
                                                                  unsigned long long *__restrict__ hist) {
  __shared__ uint32_t s_light;
  __shared__ uint32_t s_heavy[3];
  __shared__ uint32_t s_hist[2 * SRT_NKEY];  // [0..31]: contexts by total n, [32..63]: by stop count (n, c <= SRT_CL)
  if (threadIdx.x < 2 * SRT_NKEY) s_hist[threadIdx.x] = 0;
  const uint64_t n_tiles = (n_rows + PLN_TILE - 1) / PLN_TILE;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    if (threadIdx.x == 0) {
      s_light = 0;
      s_heavy[0] = s_heavy[1] = s_heavy[2] = 0;
    }
    __syncthreads();
    const uint64_t r = tile * PLN_TILE + threadIdx.x;
    if (r < n_rows) {
      uint32_t light = 0, hcol = 0, nsat = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const uint32_t c = counts[r * 5 + b];
        const uint32_t s = nsat + c;
        nsat = s < nsat ? 0xffffffffu : s;
        if (b < ncol) {
          light += (c != 0 && c <= SRT_CL);
          hcol += (c > SRT_CL);
        }
      }
      if (light) atomicAdd(&s_light, light);
      if (hcol) atomicAdd(&s_heavy[0], hcol);
      if (nsat > SRT_CL) atomicAdd(&s_heavy[1], 1u);
      else if (nsat != 0) atomicAdd(&s_hist[nsat - 1], 1u);
      const uint32_t c4 = counts[r * 5 + 4];
      if (ncol == 4 && c4 > SRT_CL) atomicAdd(&s_heavy[2], 1u);
      else if (ncol == 4 && c4 != 0) atomicAdd(&s_hist[SRT_NKEY + c4 - 1], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      n_light[tile] = s_light;
      for (int k = 0; k < 3; ++k)
        if (s_heavy[k]) atomicAdd(&heavy_counts[k], (unsigned long long)s_heavy[k]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 2 * SRT_NKEY && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

// Pass 2: per tile, counting sort of the product-path items by c (LDS histogram, replicated 8x),
// written as a padded uint16 list; heavy items appended to the global lists.
__global__ __launch_bounds__(PLN_BUILD_THREADS) void plan_fill_kernel(const uint32_t *__restrict__ counts, uint64_t n_rows,
                                                                 int ncol, const pln_tile_info *__restrict__ info,
                                                                 uint16_t *__restrict__ items,
                                                                 pln_heavy_col *__restrict__ heavy_col,
                                                                 pln_heavy_row *__restrict__ heavy_row,
                                                                 uint64_t *__restrict__ heavy_stop,
                                                                 unsigned long long *__restrict__ cursors) {
  __shared__ uint32_t hist[SRT_NHIST];
  __shared__ uint32_t offs[SRT_NHIST];
  __shared__ uint32_t scan[SRT_WAVES];
  __shared__ uint16_t sorted[PLN_ITEMS_MAX];
  const uint32_t tid = threadIdx.x, rep = tid & (SRT_REP - 1);
  const uint64_t n_tiles = (n_rows + PLN_TILE - 1) / PLN_TILE;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    if (tid < SRT_NHIST) hist[tid] = 0;
    __syncthreads();
    const uint64_t r = tile * PLN_TILE + tid;
    uint32_t c[5] = {0, 0, 0, 0, 0}, rank[5] = {0, 0, 0, 0, 0};
    if (r < n_rows) {
      double n = 0.0;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        c[b] = counts[r * 5 + b];
        n += (double)c[b];
      }
      if (n > (double)SRT_CL) {
        const unsigned long long k = atomicAdd(&cursors[1], 1ull);
        heavy_row[k].row = r;
        heavy_row[k].n = n;
      }
      if (ncol == 4 && c[4] > SRT_CL) heavy_stop[atomicAdd(&cursors[2], 1ull)] = c[4];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        if (b >= ncol) c[b] = 0;
        if (c[b] > SRT_CL) {
          const unsigned long long k = atomicAdd(&cursors[0], 1ull);
          heavy_col[k].off = r * 5 + b;
          heavy_col[k].c = c[b];
          c[b] = 0;
        }
        if (c[b] != 0) rank[b] = atomicAdd(&hist[(c[b] - 1) * SRT_REP + rep], 1u);
      }
    }
    __syncthreads();
    {
      uint32_t total;
      const uint32_t v = tid < SRT_NHIST ? hist[tid] : 0u;
      // plain block scan (same shape as srt_block_exscan, with ordinary barriers)
      const int lane = tid & 63, wave = tid >> 6;
      uint32_t incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += o;
      }
      if (lane == 63) scan[wave] = incl;
      __syncthreads();
      uint32_t base = 0;
      total = 0;
      for (int w = 0; w < SRT_WAVES; ++w) {
        if (w < wave) base += scan[w];
        total += scan[w];
      }
      if (tid < SRT_NHIST) offs[tid] = base + incl - v;
      (void)total;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 5; ++b)
      if (c[b] != 0) sorted[offs[(c[b] - 1) * SRT_REP + rep] + rank[b]] = (uint16_t)(tid * 5 + b);
    __syncthreads();
    const pln_tile_info ti = info[tile];
    const uint32_t padded = (ti.n_light + 63u) & ~63u;
    uint16_t *dst = items + (size_t)ti.off16 * 8;
    for (uint32_t i = tid; i < padded; i += PLN_BUILD_THREADS) dst[i] = i < ti.n_light ? sorted[i] : (uint16_t)PLN_SENTINEL;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// per-step evaluation
// ---------------------------------------------------------------------------------------------
struct pln_view {  // device-side view of a plan
  const pln_tile_info *info;
  const uint16_t *items;
  const pln_heavy_col *heavy_col;
  const pln_heavy_row *heavy_row;
  const uint64_t *heavy_stop;
  const unsigned long long *hist;  // [0..31] contexts with total n = j+1, [32..63] with stop count j+1 (<= SRT_CL)
  uint64_t n_heavy_col, n_heavy_row, n_heavy_stop;
};

// DMA of `bytes` (multiple of 16) to LDS: whole 1 KiB pieces round-robin over the waves, the last
// partial piece with the surplus lanes masked off.
__device__ __forceinline__ void pln_dma(void *lds, const void *src, uint32_t bytes, uint32_t wave, uint32_t lane) {
  const uint32_t d = (uint32_t)(uintptr_t)lds;
  const unsigned char *s = static_cast<const unsigned char *>(src) + lane * 16u;
  const uint32_t pieces = (bytes + 1023u) >> 10;
  for (uint32_t piece = wave; piece < pieces; piece += PLN_WAVES) {
    const unsigned char *g = s + (piece << 10);
    const uint32_t m = srt_uniform(d + (piece << 10));
    if ((piece << 10) + lane * 16u < bytes)
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m) : "memory", "m0");
  }
}

// Units of sorted product-path items of the current tile.  `decode(off, &x)` returns the item's
// count and writes its concentration; `accumulate(x, o)` folds D, P into the thread's sums.
template <int ILP, typename Decode, typename Accum>
__device__ __forceinline__ void pln_unit(const uint16_t *items, uint32_t n_light, uint32_t un, uint32_t lane,
                                         const double2 *logtab, Decode decode, Accum accumulate) {
  const uint32_t padded = (n_light + 63u) & ~63u;
  const uint32_t base = un * 64u * ILP;
  uint32_t ci[ILP];
  double x[ILP];
  bear_dp o[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    const uint32_t idx = base + 64u * i + lane;
    const uint32_t off = idx < padded ? (uint32_t)items[idx] : (uint32_t)PLN_SENTINEL;
    ci[i] = decode(off, &x[i]);
  }
  // smallest / largest count: first lane of the unit, last occupied lane of the last occupied slice
  uint32_t cmax = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    const uint32_t lo = base + 64u * i;
    if (n_light > lo) {
      const uint32_t n = n_light - lo > 64u ? 64u : n_light - lo;
      const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)ci[i], (int)(n - 1u));
      cmax = v > cmax ? v : cmax;
    }
  }
  const uint32_t cmin = n_light - base >= 64u * ILP ? (uint32_t)__builtin_amdgcn_readlane((int)ci[0], 0) : 0u;
  srt_light<ILP>(x, ci, cmin, cmax, logtab, o);
#pragma unroll
  for (int i = 0; i < ILP; ++i) accumulate(x[i], o[i]);
}

// Dynamic work distribution inside a tile: every wave draws tickets from an LDS counter.
__device__ __forceinline__ uint32_t pln_ticket(uint32_t *counter, uint32_t lane) {
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(counter, 1u);
  return srt_uniform(t);
}

// Guarded synchronous staging for the ragged last tile (any block size).
__device__ __forceinline__ void pln_stage(uint32_t *lds, const uint32_t *src, uint32_t n_dwords) {
  for (uint32_t i = threadIdx.x; i < n_dwords; i += blockDim.x) lds[i] = src[i];
}

// ---- mode N ---------------------------------------------------------------------------------
struct pln_lds_n {
  double pri[2][PLN_TILE * 5 + 2];      // [.][PLN_SENTINEL] = 1.0
  uint32_t cnt[2][PLN_TILE * 5 + 4];    // [.][PLN_SENTINEL] = 0
  uint16_t items[2][PLN_ITEMS_MAX];
  double2 logtab[BEAR_LOGTAB_N];
  double tabD[SRT_NKEY];                // D(u + 5 eps, j + 1)
  double tabP[SRT_NKEY];
  uint32_t ticket[2];                   // per buffer parity; zeroed one tile ahead
};

// NORM: the caller asserts that every prior row sums to one (true for every ar_func of the reference,
// all of which end in a softmax, ar_funcs.py:44,97,121-126).  Then A = u + 5 eps for every context
// and the context terms collapse to the plan's histogram over n: no per-context pass at all.
template <int TIMING, bool NORM>  // TIMING 1: diagnostic build recording per-wave s_memtime totals
__global__ __launch_bounds__(PLN_THREADS, PLN_WAVES_PER_SIMD) void dm_prior_plan_kernel(const uint32_t *__restrict__ counts,
                                                                        const double *__restrict__ prior, uint64_t n_rows,
                                                                        bear_params prm, pln_view pv,
                                                                        const double2 *__restrict__ logtab_g,
                                                                        double *__restrict__ partials,
                                                                        unsigned long long *__restrict__ dbg) {
  unsigned long long tph[4] = {0, 0, 0, 0}, t_prev = 0;
#define PLN_STAMP(k)                                              \
  if (TIMING) {                                                   \
    const unsigned long long now = __builtin_amdgcn_s_memtime();  \
    tph[k] += now - t_prev;                                       \
    t_prev = now;                                                 \
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_n &S = *reinterpret_cast<pln_lds_n *>(srt_smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  const uint64_t n_tiles = (n_rows + PLN_TILE - 1) / PLN_TILE;
  double acc[2] = {0.0, 0.0};

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < SRT_NKEY) {
    const bear_dp o = srt_general(u + eps5, (double)(tid + 1));
    S.tabD[tid] = o.D;
    S.tabP[tid] = o.P;
  }
  if (tid < 2) {
    S.pri[tid][PLN_SENTINEL] = 1.0;
    S.cnt[tid][PLN_SENTINEL] = 0;
    S.ticket[tid] = 0;
  }

  auto stage = [&](uint64_t tile, uint32_t buf, pln_tile_info ti) {
    const uint64_t row0 = tile * PLN_TILE;
    if (n_rows - row0 >= PLN_TILE) {
      pln_dma(S.pri[buf], prior + row0 * 5, PLN_TILE * 40, wave, lane);
      pln_dma(S.cnt[buf], counts + row0 * 5, PLN_TILE * 20, wave, lane);
    } else {
      const uint32_t rows = (uint32_t)(n_rows - row0);
      pln_stage(S.cnt[buf], counts + row0 * 5, rows * 5);
      pln_stage(reinterpret_cast<uint32_t *>(S.pri[buf]), reinterpret_cast<const uint32_t *>(prior + row0 * 5), rows * 10);
    }
    pln_dma(S.items[buf], pv.items + (size_t)ti.off16 * 8, ((ti.n_light + 63u) & ~63u) * 2u, wave, lane);
  };
  // Scalar (s_load) fetch of a tile descriptor: a vector-memory load here would make the compiler
  // wait for vmcnt(0) at its first use -- and vmcnt is in order, so that wait would also drain
  // the LDS-DMA queue of the next tile.
  auto load_info = [&](uint64_t tile) {
    pln_tile_info ti;
    ti.off16 = 0;
    ti.n_light = 0;
    if (tile < n_tiles) {
      const __attribute__((address_space(4))) pln_tile_info *ic =
          (const __attribute__((address_space(4))) pln_tile_info *)(uintptr_t)pv.info;
      ti.off16 = ic[tile].off16;
      ti.n_light = ic[tile].n_light;
    }
    return ti;
  };

  const uint64_t G = gridDim.x;
  uint64_t tile = blockIdx.x;
  uint32_t buf = 0;
  pln_tile_info ti_cur = load_info(tile), ti_nxt = load_info(tile + G);
  if (tile < n_tiles) stage(tile, 0, ti_cur);
  for (; tile < n_tiles; tile += G, buf ^= 1u) {
    const uint64_t row0 = tile * PLN_TILE;
    const uint32_t rows = (uint32_t)((n_rows - row0 < PLN_TILE) ? (n_rows - row0) : PLN_TILE);
    if (TIMING) t_prev = __builtin_amdgcn_s_memtime();
    srt_wait_dma();  // this wave's pieces of the current tile have landed
    srt_sync();      // ... and everybody else's; the previous tile is fully consumed
    PLN_STAMP(0)
    if (tile + G < n_tiles) stage(tile + G, buf ^ 1u, ti_nxt);
    const pln_tile_info ti_nn = load_info(tile + 2 * G);
    PLN_STAMP(1)
    const double *pri = S.pri[buf];
    const uint32_t *cnt = S.cnt[buf];
    if (tid == 0) S.ticket[buf ^ 1u] = 0;  // next tile's counter (its last readers passed the barrier above)
    // Work list of the tile, most expensive first: item units from the sorted tail down, then the
    // 64-context row chunks.  Waves draw tickets until the list is exhausted.
    const uint32_t n_units = (((ti_cur.n_light + 63u) & ~63u) + 64u * PLN_ILP - 1) / (64u * PLN_ILP);
    const uint32_t n_work = n_units + (NORM ? 0u : PLN_TILE / 64);
    for (uint32_t w = pln_ticket(&S.ticket[buf], lane); w < n_work; w = pln_ticket(&S.ticket[buf], lane)) {
      if (NORM || w < n_units) {
        // ---- D: column items
        pln_unit<PLN_ILP>(
            S.items[buf], ti_cur.n_light, n_units - 1 - w, lane, S.logtab,
            [&](uint32_t off, double *x) {
              *x = __builtin_fma(pri[off], u, eps);
              return cnt[off];
            },
            [&](double x, const bear_dp &o) {
              acc[0] += o.D;
              acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
            });
        continue;
      }
      // ---- A: context terms  -D(A, n), (A - 5 eps) P(A, n)   with A = S u + 5 eps
      const uint32_t row = (w - n_units) * 64u + lane;
      const uint32_t rr = row < rows ? row : rows - 1;
      uint32_t c[5];
      double f[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        c[b] = cnt[rr * 5 + b];
        f[b] = pri[rr * 5 + b];
      }
      const double S5 = ((f[0] + f[1]) + (f[2] + f[3])) + f[4];
      uint32_t nsat = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const uint32_t s = nsat + c[b];
        nsat = s < nsat ? 0xffffffffu : s;
      }
      if (row >= rows || nsat > SRT_CL) nsat = 0;  // totals beyond SRT_CL are in the plan's heavy list
      const bool shared = __builtin_fabs(S5 - 1.0) <= SRT_SUM1_TOL;
      if (nsat != 0 && shared) {
        acc[0] -= S.tabD[nsat - 1];
        acc[1] = __builtin_fma(u, S.tabP[nsat - 1], acc[1]);
      }
      const uint32_t own = (nsat != 0 && !shared) ? nsat : 0u;  // general concentrations: own A
      if (__builtin_amdgcn_ballot_w64(own != 0)) {
        uint32_t cm = own;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const uint32_t o2 = (uint32_t)__shfl_xor((int)cm, off, 64);
          cm = o2 > cm ? o2 : cm;
        }
        const double xa[1] = {own ? __builtin_fma(S5, u, eps5) : 1.0};
        const uint32_t ca[1] = {own};
        bear_dp o[1];
        srt_light<1>(xa, ca, 0u, srt_uniform(cm), S.logtab, o);
        acc[0] -= o[0].D;
        acc[1] = __builtin_fma(xa[0] - eps5, o[0].P, acc[1]);
      }
    }
    PLN_STAMP(2)
    if (TIMING) tph[3] += 1;
    ti_cur = ti_nxt;
    ti_nxt = ti_nn;
  }
#undef PLN_STAMP
  if (TIMING && dbg && lane == 0)
    for (int k = 0; k < 4; ++k) dbg[((size_t)blockIdx.x * PLN_WAVES + wave) * 4 + k] = tph[k];
  srt_wait_dma();
  // ---- Stirling-path items of the whole table, densely packed over the grid
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    const double x = __builtin_fma(prior[h.off], u, eps);
    const bear_dp o = srt_general(x, (double)h.c);
    acc[0] += o.D;
    acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
  }
  for (uint64_t i = gtid; i < pv.n_heavy_row; i += gsz) {
    const pln_heavy_row h = pv.heavy_row[i];
    const double *f = prior + h.row * 5;
    const double A = __builtin_fma(((f[0] + f[1]) + (f[2] + f[3])) + f[4], u, eps5);
    const bear_dp o = srt_general(A, h.n);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(A - eps5, o.P, acc[1]);
  }
  if (NORM && blockIdx.x == 0 && tid < SRT_CL) {  // context terms with the shared A, weighted by their multiplicity
    const double m = (double)pv.hist[tid];
    acc[0] -= m * S.tabD[tid];
    acc[1] = __builtin_fma(u * m, S.tabP[tid], acc[1]);
  }
  __syncthreads();
  block_store_partials<2>(acc, partials);
}

// ---- mode R ---------------------------------------------------------------------------------
struct pln_lds_r {
  uint32_t trn[2][PLN_TILE * 5 + 4];   // [.][PLN_SENTINEL] = 0
  uint32_t ref[2][PLN_TILE * 5 + 4];
  uint16_t items[2][PLN_ITEMS_MAX];
  double2 logtab[BEAR_LOGTAB_N];
  double tabD[2][SRT_NKEY];            // [0]: context term (x = A), [1]: stop column (x = x4)
  double tabP[2][SRT_NKEY];
  uint32_t ticket[2];
};

__global__ __launch_bounds__(PLN_THREADS, PLN_WAVES_PER_SIMD) void dm_ref_plan_kernel(const uint32_t *__restrict__ train,
                                                                      const uint32_t *__restrict__ ref, uint64_t n_rows,
                                                                      bear_params prm, pln_view pv,
                                                                      const double2 *__restrict__ logtab_g,
                                                                      double *__restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_r &S = *reinterpret_cast<pln_lds_r *>(srt_smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
  const double u = prm.inv_h, eps = prm.eps;
  const double A = u + 5.0 * eps;              // sum_b alpha_b
  const double x4 = prm.nw * prm.V * u + eps;  // alpha of the stop column
  const double VU = prm.V * u;
  const double tau = prm.tau;
  const double w2c = tau * (eps + 0.25 * VU);  // d alpha/d tau_s = -tau x + w2c
  const double nwV = prm.nw * prm.V;
  const uint64_t n_tiles = (n_rows + PLN_TILE - 1) / PLN_TILE;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < 2 * SRT_NKEY) {
    const int which = tid / SRT_NKEY, j = tid % SRT_NKEY;
    const bear_dp o = srt_general(which ? x4 : A, (double)(j + 1));
    S.tabD[which][j] = o.D;
    S.tabP[which][j] = o.P;
  }
  if (tid < 2) S.ticket[tid] = 0;
  if (tid < 8) {  // neutral cell: count 0, reference row of zeros (4 words)
    S.trn[tid >> 2][PLN_SENTINEL + (tid & 3)] = 0;
    S.ref[tid >> 2][PLN_SENTINEL + (tid & 3)] = 0;
  }

  auto stage = [&](uint64_t tile, uint32_t buf, pln_tile_info ti) {
    const uint64_t row0 = tile * PLN_TILE;
    if (n_rows - row0 >= PLN_TILE) {
      pln_dma(S.trn[buf], train + row0 * 5, PLN_TILE * 20, wave, lane);
      pln_dma(S.ref[buf], ref + row0 * 5, PLN_TILE * 20, wave, lane);
    } else {
      const uint32_t rows = (uint32_t)(n_rows - row0);
      pln_stage(S.trn[buf], train + row0 * 5, rows * 5);
      pln_stage(S.ref[buf], ref + row0 * 5, rows * 5);
    }
    pln_dma(S.items[buf], pv.items + (size_t)ti.off16 * 8, ((ti.n_light + 63u) & ~63u) * 2u, wave, lane);
  };
  // Scalar (s_load) fetch of a tile descriptor: a vector-memory load here would make the compiler
  // wait for vmcnt(0) at its first use -- and vmcnt is in order, so that wait would also drain
  // the LDS-DMA queue of the next tile.
  auto load_info = [&](uint64_t tile) {
    pln_tile_info ti;
    ti.off16 = 0;
    ti.n_light = 0;
    if (tile < n_tiles) {
      const __attribute__((address_space(4))) pln_tile_info *ic =
          (const __attribute__((address_space(4))) pln_tile_info *)(uintptr_t)pv.info;
      ti.off16 = ic[tile].off16;
      ti.n_light = ic[tile].n_light;
    }
    return ti;
  };
  // bear_ref.py:30-33 (Jukes-Cantor on the L1-normalised reference row), :63-68 (mix), bear_ref.py:106
  auto alpha_from = [&](double rb, double R) {
    const double dev = __builtin_fma(rb + eps, bear_rcp(R), -0.25);
    return __builtin_fma(__builtin_fma(prm.E, dev, 0.25), VU, eps);
  };
  auto accumulate = [&](double x, const bear_dp &o) {
    const double w1 = eps - x;
    acc[0] += o.D;
    acc[1] = __builtin_fma(w1, o.P, acc[1]);
    acc[2] = __builtin_fma(__builtin_fma(-tau, x, w2c), o.P, acc[2]);
    acc[3] = __builtin_fma(nwV * w1, o.P, acc[3]);
  };

  const uint64_t G = gridDim.x;
  uint64_t tile = blockIdx.x;
  uint32_t buf = 0;
  pln_tile_info ti_cur = load_info(tile), ti_nxt = load_info(tile + G);
  if (tile < n_tiles) stage(tile, 0, ti_cur);
  for (; tile < n_tiles; tile += G, buf ^= 1u) {
    const uint64_t row0 = tile * PLN_TILE;
    const uint32_t rows = (uint32_t)((n_rows - row0 < PLN_TILE) ? (n_rows - row0) : PLN_TILE);
    srt_wait_dma();
    srt_sync();
    if (tile + G < n_tiles) stage(tile + G, buf ^ 1u, ti_nxt);
    const pln_tile_info ti_nn = load_info(tile + 2 * G);
    const uint32_t *trn = S.trn[buf];
    const uint32_t *rfc = S.ref[buf];
    if (tid == 0) S.ticket[buf ^ 1u] = 0;
    // The context term (x = A) and the stop column (x = x4) have the same concentration in every
    // context: their sums over the table are the plan's histograms times two small tables (added
    // once, after the loop).  Per tile only the column items b < 4 remain.
    const uint32_t n_units = (((ti_cur.n_light + 63u) & ~63u) + 64u * PLN_ILP - 1) / (64u * PLN_ILP);
    for (uint32_t w = pln_ticket(&S.ticket[buf], lane); w < n_units; w = pln_ticket(&S.ticket[buf], lane)) {
      {
        pln_unit<PLN_ILP>(
            S.items[buf], ti_cur.n_light, n_units - 1 - w, lane, S.logtab,
            [&](uint32_t off, double *x) {
              const uint32_t *rr = &rfc[((off * 52429u) >> 18) * 5u];  // row start: 5 * (off / 5), off < 2^16
              const double R = (double)(((uint64_t)rr[0] + rr[1]) + ((uint64_t)rr[2] + rr[3])) + 4.0 * eps;  // bear_ref.py:335-337, 30
              *x = alpha_from((double)rfc[off], R);
              return trn[off];
            },
            accumulate);
      }
    }
    ti_cur = ti_nxt;
    ti_nxt = ti_nn;
  }
  srt_wait_dma();
  // ---- Stirling-path items of the whole table
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    const uint32_t *rr = ref + (h.off / 5) * 5;
    const double R = (double)(((uint64_t)rr[0] + rr[1]) + ((uint64_t)rr[2] + rr[3])) + 4.0 * eps;
    const double x = alpha_from((double)ref[h.off], R);
    accumulate(x, srt_general(x, (double)h.c));
  }
  for (uint64_t i = gtid; i < pv.n_heavy_row; i += gsz) {
    const bear_dp o = srt_general(A, pv.heavy_row[i].n);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(u, o.P, acc[1]);
  }
  for (uint64_t i = gtid; i < pv.n_heavy_stop; i += gsz) {
    const bear_dp o = srt_general(x4, (double)pv.heavy_stop[i]);
    acc[0] += o.D;
    acc[1] = __builtin_fma(eps - x4, o.P, acc[1]);
    acc[3] = __builtin_fma(VU * nwV, o.P, acc[3]);
  }
  if (blockIdx.x == 0 && tid < SRT_CL) {
    const double mn = (double)pv.hist[tid], m4 = (double)pv.hist[SRT_NKEY + tid];
    acc[0] -= mn * S.tabD[0][tid];                                // context terms: -D(A, n)
    acc[1] = __builtin_fma(u * mn, S.tabP[0][tid], acc[1]);
    const double P4 = m4 * S.tabP[1][tid];                        // stop column: +D(x4, c)
    acc[0] += m4 * S.tabD[1][tid];
    acc[1] = __builtin_fma(eps - x4, P4, acc[1]);
    acc[3] = __builtin_fma(VU * nwV, P4, acc[3]);                 // d alpha_4/d nu_s = u nw V^2
  }
  __syncthreads();
  block_store_partials<4>(acc, partials);
}

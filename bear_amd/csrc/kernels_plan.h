// kernels_plan.h -- "planned" DM-marginal + gradient kernels: the per-step hot path for a count
// table that stays resident in HBM across optimizer steps (every epoch of bear_net.train /
// bear_ref.train re-reads the same cached count table, bear_model/dataloader.py:47-48).
//
// Everything that depends on the counts only is done ONCE, at load time (bear_plan_create); the
// counts never change between steps -- only h, the prior rows and (tau, nu) do.
//
//   plan = a sorted sparse encoding of the count table:
//     * the table is cut into tiles of consecutive contexts such that every tile holds at most
//       PLN_UNITS x 64 product-path work items (column b of context r with 1 <= c <= SRT_CL) and at most
//       PLN_RMAX contexts: on k-mer tables every tile is exactly 16 full 64-lane units, one per wave
//     * per tile, one contiguous block: thresholds E[c] = #items with count <= c (the items are sorted
//       by count, so an item's count follows from its position), the row totals n (uint8, 0 = none or
//       large), the items as uint16 flat offsets r*5+b
//     * global lists of the rare Stirling-path items (c > SRT_CL) and rows (n > SRT_CL), and histograms of
//       the row totals / stop counts for the terms whose concentration is shared by all contexts
//   step (bear_dm_prior_plan_f64 / bear_dm_ref_plan_f64), per tile, ONE barrier:
//     0  the prior rows (mode N) / reference rows (mode R) and the plan block of a later tile stream
//        into a ring of LDS buffers by LDS-DMA while the current tile is evaluated; the count rows
//        themselves are not read at all
//     D  wave w evaluates unit w: p = prod (x+j), p' by the product rule, one table log and one
//        reciprocal per item; loops are wave-uniform (sorted), all lanes busy (full units)
//     A  (mode N without the normalisation promise) wave w walks contexts 64w .. 64w+63: S = sum prior,
//        context term from the shared-A table when S = 1 to 2 ulp, own product otherwise
//     after the tiles: the global Stirling-path lists, densely packed over all threads, and the
//     histogram x table terms.
#pragma once
#include "kernels_sorted.h"

#ifndef PLN_THREADS
#define PLN_THREADS 1024
#endif
#define PLN_WAVES (PLN_THREADS / 64)
#ifndef PLN_UNITS
#define PLN_UNITS 32                          // units per tile: two per wave, drawn dynamically, dearest first
#endif
#define PLN_NI (PLN_UNITS * 64)               // product-path items per tile (at most)
#ifndef PLN_NI_CUT
#define PLN_NI_CUT PLN_NI                      // ... that the greedy cut gives a tile
#endif
#ifndef PLN_RMAX
#define PLN_RMAX 1664                         // contexts per tile (at most; LDS)
#endif
#ifndef PLN_HCAP
#define PLN_HCAP 128                          // large-count items / contexts evaluated inside a tile (rest: global lists)
#endif
// Row totals n in (SRT_CL, PLN_NBIG] are ALSO counted in a histogram of the plan (bear_plan::hist + 2 SRT_NKEY, [n]): where every
// context shares its concentration total A (softmax rows: A = u + 5 eps; mode R always) the context term -D(A, n) is a function of n
// alone, and sum over rows = sum_n count[n] D(A, n) -- a few thousand Stirling evaluations per LAUNCH instead of one per row
// (5 % of the rows of the k = 13 table: a 64-lane unit of ~400 dependent instructions per tile, and 3.5e5 evaluations in the mode-R
// step of configs[1]).  The lists keep every such row (kernels with a concentration total per row read them as before); kernels
// that take the histogram skip the listed rows with n <= PLN_NBIG (round 6).
#define PLN_NBIG 4096
#define PLN_BLOCKS_PER_CU (1024 / PLN_THREADS)   // resident blocks of the planned step kernels per CU (LDS: 160 KB / that many)
#define PLN_QUAD 4                            // tiles start on multiples of 4 contexts (16-byte aligned rows)
#define PLN_LIVE_STRIDE (PLN_RMAX + 8)        // uint16 per tile of the live-context lists (a multiple of 8: 16-byte rows)
#define PLN_SENTINEL (PLN_RMAX * 5)           // flat offset of the neutral cell (prior = 1 / ref row = 0)
#ifndef PLN_NBUF
#define PLN_NBUF 2                            // LDS ring depth: tiles in flight = PLN_NBUF - 1
#endif
// bytes of one tile's plan block: E | nrow | items | heavy column items (off, c) | heavy contexts (row, n)
#define PLN_BLOCK_MAX (64 + PLN_RMAX + PLN_NI * 2 + PLN_HCAP * (2 + 4 + 2 + 8))

struct pln_tile {
  uint64_t row0;
  uint32_t rows_items;  // rows << 16 | n_light
  uint32_t off16;       // start of the tile's block in the plan stream, 16-byte units
  uint32_t hc_hr;       // in-tile large-count column items << 16 | in-tile large-total contexts
  uint32_t blk16;       // size of the block in 16-byte units
  uint64_t pad;
};
static_assert(sizeof(pln_tile) == 32, "tile descriptors are fetched with one s_load_dwordx8");
#ifndef PLN_CHUNK
#define PLN_CHUNK 128                       // contexts per context-term ticket (multiple of 64)
#endif
#ifndef PLN_DMA_WAVES
#define PLN_DMA_WAVES 2                     // waves of each block that only stream tiles into LDS (see dm_prior_plan_kernel)
#endif
#ifndef PLN_PREFETCH_KIB
#define PLN_PREFETCH_KIB 24                 // L2 prefetch of the tile after next by the DMA waves of the light mode-N forms (0: off)
#endif
#define PLN_DESC_CHUNK 32                   // descriptors per 1 KiB LDS-DMA piece
#define PLN_DESC_PAD (2 * PLN_DESC_CHUNK)   // zeroed descriptors behind the last tile (plan allocation)

struct pln_layout {  // byte offsets inside a tile's block (all multiples of 16)
  uint32_t nrow, items, hoff, hcnt, hrow, hn, end;
};
__host__ __device__ inline pln_layout pln_block_layout(uint32_t rows, uint32_t n_light, uint32_t hc, uint32_t hr) {
  pln_layout L;
  L.nrow = 64u;
  L.items = L.nrow + ((rows + 15u) & ~15u);
  L.hoff = L.items + (((n_light + 63u) & ~63u) * 2u);
  L.hcnt = L.hoff + (((hc + 7u) & ~7u) * 2u);
  L.hrow = L.hcnt + (((hc + 3u) & ~3u) * 4u);
  L.hn = L.hrow + (((hr + 7u) & ~7u) * 2u);
  L.end = L.hn + (((hr + 1u) & ~1u) * 8u);
  return L;
}

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
// Pass A: number of product-path items per group of 4 contexts (uint8), global counts of heavy column
// items / heavy contexts / heavy stop counts, histograms of row totals and stop counts <= SRT_CL.
__global__ __launch_bounds__(256) void plan_scan_kernel(const uint32_t *__restrict__ counts, uint64_t n_rows, int ncol,
                                                        uint8_t *__restrict__ quad_light,  // [3][n_quads]: light, heavy cols, heavy rows
                                                        unsigned long long *__restrict__ heavy_counts,
                                                        unsigned long long *__restrict__ hist) {
  __shared__ uint32_t s_heavy[3];
  __shared__ uint32_t s_hist[2 * SRT_NKEY];
  __shared__ uint32_t s_big[PLN_NBIG + 1];          // row totals in (SRT_CL, PLN_NBIG]: hist[2 SRT_NKEY + n]
  __shared__ unsigned long long s_total, s_cells;   // sum of all counts / cells that hold one (heavy_counts[6], [7]: bear_plan::count_total)
  __shared__ uint32_t s_cmax;                       // largest count (heavy_counts[8])
  unsigned long long my_total = 0ull;
  uint32_t my_cells = 0u, my_cmax = 0u;
  if (threadIdx.x == 0) {
    s_total = s_cells = 0ull;
    s_cmax = 0u;
  }
  if (threadIdx.x < 3) s_heavy[threadIdx.x] = 0;
  if (threadIdx.x < 2 * SRT_NKEY) s_hist[threadIdx.x] = 0;
  for (uint32_t k = threadIdx.x; k <= PLN_NBIG; k += 256) s_big[k] = 0;
  __syncthreads();
  const uint64_t n_quads = (n_rows + PLN_QUAD - 1) / PLN_QUAD;
  for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < n_quads; g += (uint64_t)gridDim.x * 256) {
    uint32_t light = 0, qhc = 0, qhr = 0;
    for (uint64_t r = PLN_QUAD * g; r < PLN_QUAD * g + PLN_QUAD && r < n_rows; ++r) {
      uint32_t nsat = 0, hcol = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const uint32_t c = counts[r * 5 + b];
        const uint32_t s = nsat + c;
        nsat = s < nsat ? 0xffffffffu : s;
        my_total += c;
        my_cells += c != 0u;
        my_cmax = c > my_cmax ? c : my_cmax;
        if (b < ncol) {
          light += (c != 0 && c <= SRT_CL);
          hcol += (c > SRT_CL);
        }
      }
      qhc += hcol;
      qhr += nsat > SRT_CL;
      if (hcol) atomicAdd(&s_heavy[0], hcol);
      if (nsat > SRT_CL) {
        atomicAdd(&s_heavy[1], 1u);
        if (nsat <= PLN_NBIG) atomicAdd(&s_big[nsat], 1u);
      } else if (nsat != 0) atomicAdd(&s_hist[nsat - 1], 1u);
      const uint32_t c4 = counts[r * 5 + 4];
      if (ncol == 4 && c4 > SRT_CL) atomicAdd(&s_heavy[2], 1u);
      else if (ncol == 4 && c4 != 0) atomicAdd(&s_hist[SRT_NKEY + c4 - 1], 1u);
    }
    quad_light[g] = (uint8_t)light;
    quad_light[n_quads + g] = (uint8_t)qhc;
    quad_light[2 * n_quads + g] = (uint8_t)qhr;
  }
  if (my_total) {
    atomicAdd(&s_total, my_total);
    atomicAdd(&s_cells, (unsigned long long)my_cells);
    atomicMax(&s_cmax, my_cmax);
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_total) {
    atomicAdd(&heavy_counts[6], s_total);
    atomicAdd(&heavy_counts[7], s_cells);
    atomicMax(&heavy_counts[8], (unsigned long long)s_cmax);
  }
  if (threadIdx.x < 3 && s_heavy[threadIdx.x]) atomicAdd(&heavy_counts[threadIdx.x], (unsigned long long)s_heavy[threadIdx.x]);
  if (threadIdx.x < 2 * SRT_NKEY && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
  for (uint32_t k = threadIdx.x; k <= PLN_NBIG; k += 256)
    if (s_big[k]) atomicAdd(&hist[2 * SRT_NKEY + k], (unsigned long long)s_big[k]);
}

// ---- tile cutting on the device --------------------------------------------------------------------------------------------
// The greedy cut (a tile takes groups of 4 contexts while it holds <= PLN_RMAX contexts and <= PLN_NI product-path items) is a
// sequential scan over n_rows / 4 bytes.  A tile spans at most PLN_RMAX / 4 groups, so the cuts inside a CHUNK of groups depend
// on nothing but the chunk's entry point, one of PLN_CUT_SPAN offsets: (1) every (chunk, entry offset) is walked in parallel
// (exit offset into the next chunk + number of tiles started), (2) one thread follows the chunks' entry points, (3) one thread
// per chunk walks again from its real entry point and writes the descriptors, (4) a scan turns block sizes into stream offsets.
#define PLN_CUT_CHUNK_MAX 8192                   // groups of 4 contexts per chunk: 2048 up to 2e8 contexts (more threads in the one-thread-per-chunk
                                                 // kernels), 8192 beyond (fewer steps of the one-thread chain): plan_cut_chunk()
static inline uint32_t plan_cut_chunk(uint64_t n_quads) { return n_quads <= 50000000ull ? 2048u : (uint32_t)PLN_CUT_CHUNK_MAX; }
#define PLN_CUT_SPAN (PLN_RMAX / PLN_QUAD)       // possible entry offsets of a chunk

// one greedy tile from group q0: returns the first group of the next tile; sums of the three per-group counters
__device__ __forceinline__ uint64_t plan_cut_one(const uint8_t *__restrict__ quad, uint64_t n_quads, uint64_t q0, uint32_t *items,
                                                 uint32_t *hcol, uint32_t *hrow) {
  uint32_t it = 0, hc = 0, hr = 0, rows = 0;
  uint64_t q = q0;
  while (q < n_quads && rows + PLN_QUAD <= PLN_RMAX && it + quad[q] <= PLN_NI_CUT) {
    it += quad[q];
    if (hcol) {
      hc += quad[n_quads + q];
      hr += quad[2 * n_quads + q];
    }
    rows += PLN_QUAD;
    ++q;
  }
  *items = it;
  if (hcol) {
    *hcol = hc;
    *hrow = hr;
  }
  return q;
}
// step[q] = length (in groups) of the greedy tile that starts at group q, for every q of a chunk: one thread per chunk, a
// two-pointer window (dropping the first group of a tile can only let it reach further), O(chunk) steps
__global__ __launch_bounds__(256) void plan_cut_step_kernel(const uint8_t *__restrict__ quad, uint64_t n_quads, uint64_t n_chunks,
                                                            uint32_t chunk, uint16_t *__restrict__ step) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_chunks) return;
  const uint64_t q0 = k * chunk, q_end = q0 + chunk < n_quads ? q0 + chunk : n_quads;
  uint64_t r = q0;
  uint32_t items = 0;
  for (uint64_t q = q0; q < q_end; ++q) {
    if (r < q) {      // cannot happen (a tile holds at least one group); keeps the window well formed
      r = q;
      items = 0;
    }
    while (r < n_quads && (uint32_t)(r - q + 1) * PLN_QUAD <= PLN_RMAX && items + quad[r] <= PLN_NI_CUT) items += quad[r++];
    step[q] = (uint16_t)(r - q);
    items -= quad[q];
  }
}
// every (chunk, entry offset): follow the tiles from the entry point to the chunk's end
__global__ __launch_bounds__(256) void plan_cut_walk_kernel(const uint16_t *__restrict__ step, uint64_t n_quads, uint64_t n_chunks,
                                                            uint32_t chunk, uint32_t *__restrict__ walk) {   // exit offset | tiles started << 16
  const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= n_chunks * PLN_CUT_SPAN) return;
  const uint64_t k = id / PLN_CUT_SPAN, j = id - k * PLN_CUT_SPAN, end = (k + 1) * chunk;
  uint64_t q = k * chunk + j;
  uint32_t cnt = 0;
  while (q < end && q < n_quads) {
    q += step[q];
    ++cnt;
  }
  walk[id] = (uint32_t)(q >= end ? q - end : 0) | (cnt << 16);
}
static_assert(PLN_CUT_CHUNK_MAX < 65536 && PLN_CUT_SPAN < 65536 && PLN_CUT_SPAN <= 2048, "walk table: 16-bit fields; a tile is shorter than a chunk");
// entry[k] = first group of the first tile that starts in chunk k, base[k] = index of that tile; meta = {n_tiles}
__global__ void plan_cut_chain_kernel(const uint32_t *__restrict__ walk, uint64_t n_chunks, uint32_t chunk, uint64_t *__restrict__ entry,
                                      uint64_t *__restrict__ base, unsigned long long *__restrict__ meta) {
  if (blockIdx.x || threadIdx.x) return;
  uint64_t j = 0, tiles = 0;
  for (uint64_t k = 0; k < n_chunks; ++k) {
    entry[k] = k * chunk + j;
    base[k] = tiles;
    const uint32_t w = walk[k * PLN_CUT_SPAN + j];
    tiles += w >> 16;
    j = w & 0xffffu;
  }
  meta[0] = tiles;
}
__global__ __launch_bounds__(256) void plan_cut_write_kernel(const uint8_t *__restrict__ quad, uint64_t n_quads, uint64_t n_rows, int ncol,
                                                             uint64_t n_chunks, uint32_t chunk, const uint64_t *__restrict__ entry,
                                                             const uint64_t *__restrict__ base, pln_tile *__restrict__ tiles) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_chunks) return;
  const uint64_t end = (k + 1) * chunk;
  uint64_t q = entry[k], t = base[k];
  while (q < end && q < n_quads) {
    uint32_t items, hcol, hrow;
    const uint64_t q1 = plan_cut_one(quad, n_quads, q, &items, &hcol, &hrow);
    pln_tile ti;
    ti.row0 = q * PLN_QUAD;
    uint32_t rows = (uint32_t)(q1 - q) * PLN_QUAD;
    if (ti.row0 + rows > n_rows) rows = (uint32_t)(n_rows - ti.row0);  // ragged end of the table
    // large-count items evaluated inside the tile (their rows are in LDS anyway); the surplus of very dense tiles goes to the
    // global lists.  Mode R needs no row data for large totals: they stay global.
    const uint32_t hc = hcol < PLN_HCAP ? hcol : PLN_HCAP;
    const uint32_t hr = ncol == 5 ? (hrow < PLN_HCAP ? hrow : PLN_HCAP) : 0u;
    ti.rows_items = (rows << 16) | items;
    ti.off16 = 0;
    ti.hc_hr = (hc << 16) | hr;
    ti.blk16 = pln_block_layout(rows, items, hc, hr).end / 16;
    ti.pad = 0;
    tiles[t++] = ti;
    q = q1;
  }
}
// off16 = exclusive prefix sum of blk16 (one block; tiles are ~1e-5 of the contexts); meta[1] = total, meta[2] = 1 on overflow
__global__ __launch_bounds__(1024) void plan_cut_offsets_kernel(pln_tile *__restrict__ tiles, uint64_t n_tiles,
                                                                unsigned long long *__restrict__ meta) {
  __shared__ unsigned long long part[1024];
  const uint64_t per = (n_tiles + 1023) / 1024, lo = per * threadIdx.x, hi = lo + per < n_tiles ? lo + per : n_tiles;
  unsigned long long s = 0;
  for (uint64_t t = lo; t < hi; ++t) s += tiles[t].blk16;
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int k = 0; k < 1024; ++k) {
      const unsigned long long v = part[k];
      part[k] = run;
      run += v;
    }
    meta[1] = run;
    meta[2] = run > 0xffffffffull ? 1ull : 0ull;
  }
  __syncthreads();
  unsigned long long off = part[threadIdx.x];
  for (uint64_t t = lo; t < hi; ++t) {
    tiles[t].off16 = (uint32_t)off;
    off += tiles[t].blk16;
  }
}

// Pass B: one block per tile: counting sort of the tile's product-path items by count (LDS histogram,
// replicated 8x) -> thresholds, row totals and sorted uint16 item list written as one block; heavy
// items appended to the global lists.
__global__ __launch_bounds__(1024) void plan_fill_kernel(const uint32_t *__restrict__ counts, uint64_t n_rows, int ncol,
                                                         const pln_tile *__restrict__ tiles, uint64_t n_tiles,
                                                         unsigned char *__restrict__ stream,
                                                         pln_heavy_col *__restrict__ heavy_col,
                                                         pln_heavy_row *__restrict__ heavy_row,
                                                         uint64_t *__restrict__ heavy_stop,
                                                         unsigned long long *__restrict__ cursors) {
  constexpr int RPT = (PLN_RMAX + 1023) / 1024;  // contexts per thread
  __shared__ uint32_t hist[SRT_NHIST];
  __shared__ uint32_t offs[SRT_NHIST];
  __shared__ uint32_t scan[16];
  __shared__ uint16_t sorted[PLN_NI + 64];
  __shared__ uint32_t s_hc, s_hr;
#ifdef BEAR_DET_BUILD
  // The ranks below come from atomics: inside a count the items -- and the in-tile large-count lists -- stand in whatever order
  // the threads got there.  The deterministic build sorts them (count, then offset: a bitonic pass over the tile; the lists by
  // ranking) so that a table's plan, hence the grouping of every sum taken over it, is the same bits in every run.
  __shared__ uint32_t comp[PLN_NI];
  static_assert((PLN_NI & (PLN_NI - 1)) == 0 && PLN_NI <= 2048, "bitonic sort over the tile's items: PLN_NI / 2 pairs, at most one per thread");
  __shared__ unsigned long long hkey[PLN_HCAP], htmp[PLN_HCAP];
#endif
  const uint32_t tid = threadIdx.x, rep = tid & (SRT_REP - 1);
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const pln_tile ti = tiles[t];
    const uint32_t rows = ti.rows_items >> 16, n_light = ti.rows_items & 0xffffu;
    const uint32_t hc_cap = ti.hc_hr >> 16, hr_cap = ti.hc_hr & 0xffffu;
    const pln_layout L = pln_block_layout(rows, n_light, hc_cap, hr_cap);
    unsigned char *blk = stream + (size_t)ti.off16 * 16;
    uint16_t *E = reinterpret_cast<uint16_t *>(blk);
    uint8_t *nrow = blk + L.nrow;
    uint16_t *items = reinterpret_cast<uint16_t *>(blk + L.items);
    uint16_t *hoff = reinterpret_cast<uint16_t *>(blk + L.hoff);
    uint32_t *hcnt = reinterpret_cast<uint32_t *>(blk + L.hcnt);
    uint16_t *hrow = reinterpret_cast<uint16_t *>(blk + L.hrow);
    double *hn = reinterpret_cast<double *>(blk + L.hn);
    if (tid < SRT_NHIST) hist[tid] = 0;
    if (tid == 0) {
      s_hc = 0;
      s_hr = 0;
    }
    __syncthreads();
#ifdef BEAR_DET_BUILD
    // WHICH large-count cells / large-total rows stay inside the tile (the first hc_cap / hr_cap) must not depend on who gets to
    // the counter first: positions in row order from a block-wide scan (rows lr = tid + 1024 k, k-major)
    uint32_t det_hc[RPT], det_hr[RPT];
    {
      uint32_t base_c = 0, base_r = 0;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const uint32_t lr = tid + 1024u * k;
        uint32_t nc = 0, nr = 0;
        if (lr < rows) {
          double n = 0.0;
          for (int b = 0; b < 5; ++b) {
            const uint32_t cv = counts[(ti.row0 + lr) * 5 + b];
            n += (double)cv;
            nc += (b < ncol && cv > SRT_CL) ? 1u : 0u;
          }
          nr = n > (double)SRT_CL ? 1u : 0u;
        }
        const uint32_t packed = nc | (nr << 16);            // (at most 5 * 1024 cells, 1024 rows per pass: both halves fit)
        const int lane = tid & 63, wave = tid >> 6;
        uint32_t incl = packed;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
          if (lane >= off) incl += o;
        }
        __syncthreads();                                     // (scan[] of the previous pass has been read)
        if (lane == 63) scan[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
          if (w < wave) before += scan[w];
          total += scan[w];
        }
        const uint32_t excl = before + incl - packed;
        det_hc[k] = base_c + (excl & 0xffffu);
        det_hr[k] = base_r + (excl >> 16);
        base_c += total & 0xffffu;
        base_r += total >> 16;
      }
      __syncthreads();
      if (tid == 0) {
        s_hc = base_c;
        s_hr = base_r;
      }
    }
#endif
    uint32_t c[RPT][5], rank[RPT][5];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const uint32_t lr = tid + 1024u * k;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        c[k][b] = 0;
        rank[k][b] = 0;
      }
      if (lr < rows) {
        const uint64_t r = ti.row0 + lr;
        double n = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          c[k][b] = counts[r * 5 + b];
          n += (double)c[k][b];
        }
        nrow[lr] = n < 1.0 ? (uint8_t)0 : (n <= (double)SRT_CL ? (uint8_t)n : (uint8_t)255);  // 255: large total
        if (n > (double)SRT_CL) {
#ifdef BEAR_DET_BUILD
          const uint32_t k2 = hr_cap ? det_hr[k] : 0xffffffffu;
#else
          const uint32_t k2 = hr_cap ? atomicAdd(&s_hr, 1u) : 0xffffffffu;
#endif
          if (k2 < hr_cap) {  // evaluated inside the tile (prior row already in LDS)
            hrow[k2] = (uint16_t)lr;
            hn[k2] = n;
          } else {
            const unsigned long long q = atomicAdd(&cursors[1], 1ull);
            heavy_row[q].row = r;
            heavy_row[q].n = n;
          }
        }
        if (ncol == 4 && c[k][4] > SRT_CL) heavy_stop[atomicAdd(&cursors[2], 1ull)] = c[k][4];
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          if (b >= ncol) c[k][b] = 0;
          if (c[k][b] > SRT_CL) {
#ifdef BEAR_DET_BUILD
            const uint32_t k2 = hc_cap ? det_hc[k]++ : 0xffffffffu;
#else
            const uint32_t k2 = hc_cap ? atomicAdd(&s_hc, 1u) : 0xffffffffu;
#endif
            if (k2 < hc_cap) {
              hoff[k2] = (uint16_t)(lr * 5 + b);
              hcnt[k2] = c[k][b];
            } else {
              const unsigned long long q = atomicAdd(&cursors[0], 1ull);
              heavy_col[q].off = r * 5 + b;
              heavy_col[q].c = c[k][b];
            }
            c[k][b] = 0;
          }
          if (c[k][b] != 0) rank[k][b] = atomicAdd(&hist[(c[k][b] - 1) * SRT_REP + rep], 1u);
        }
      } else if (lr < ((rows + 15u) & ~15u)) {
        nrow[lr] = 0;
      }
    }
    __syncthreads();
    {
      const uint32_t v = tid < SRT_NHIST ? hist[tid] : 0u;
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
      for (int w = 0; w < wave; ++w) base += scan[w];
      if (tid < SRT_NHIST) offs[tid] = base + incl - v;
    }
    __syncthreads();
    // thresholds: E[c] = number of items with count <= c (start of key c in the sorted order); the last
    // two slots carry the in-tile heavy list lengths
    if (tid < 30) E[tid] = (uint16_t)(tid < SRT_CL ? offs[tid * SRT_REP] : n_light);
    if (tid == 30) E[30] = (uint16_t)hc_cap;
    if (tid == 31) E[31] = (uint16_t)hr_cap;
#pragma unroll
    for (int k = 0; k < RPT; ++k)
#pragma unroll
      for (int b = 0; b < 5; ++b)
        if (c[k][b] != 0) sorted[offs[(c[k][b] - 1) * SRT_REP + rep] + rank[k][b]] = (uint16_t)((tid + 1024u * k) * 5 + b);
    __syncthreads();
#ifdef BEAR_DET_BUILD
    {
      // key of position i = the number of thresholds at or below it (the list is sorted by count already)
      for (uint32_t i = tid; i < PLN_NI; i += 1024) {
        uint32_t key = 0;
        for (uint32_t q = 1; q < SRT_CL; ++q) key += offs[q * SRT_REP] <= i ? 1u : 0u;
        comp[i] = i < n_light ? (key << 16) | sorted[i] : 0xffffffffu;
      }
      __syncthreads();
      for (uint32_t size = 2; size <= PLN_NI; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
          const uint32_t lo = 2u * tid - (tid & (stride - 1u)), hi = lo + stride;      // PLN_NI / 2 pairs, one per thread
          if (tid < PLN_NI / 2) {
            const bool up = (lo & size) == 0u;
            const uint32_t a = comp[lo], bq = comp[hi];
            if ((a > bq) == up) {
              comp[lo] = bq;
              comp[hi] = a;
            }
          }
          __syncthreads();
        }
      for (uint32_t i = tid; i < n_light; i += 1024) sorted[i] = (uint16_t)(comp[i] & 0xffffu);
      __syncthreads();
      // the in-tile large-count lists: rank by (offset) / (row)
      const uint32_t nhc = s_hc < hc_cap ? s_hc : hc_cap, nhr = s_hr < hr_cap ? s_hr : hr_cap;
      if (tid < nhc) hkey[tid] = ((unsigned long long)hoff[tid] << 32) | hcnt[tid];
      __syncthreads();
      if (tid < nhc) {
        uint32_t r = 0;
        for (uint32_t q = 0; q < nhc; ++q) r += hkey[q] < hkey[tid] ? 1u : 0u;
        htmp[r] = hkey[tid];
      }
      __syncthreads();
      if (tid < nhc) {
        hoff[tid] = (uint16_t)(htmp[tid] >> 32);
        hcnt[tid] = (uint32_t)htmp[tid];
      }
      __syncthreads();
      if (tid < nhr) hkey[tid] = ((unsigned long long)hrow[tid] << 48) | tid;      // (rows are distinct)
      if (tid < nhr) htmp[tid] = (unsigned long long)__double_as_longlong(hn[tid]);
      __syncthreads();
      uint32_t r = 0;
      unsigned long long mine = 0ull, myn = 0ull;
      if (tid < nhr) {
        for (uint32_t q = 0; q < nhr; ++q) r += hkey[q] < hkey[tid] ? 1u : 0u;
        mine = hkey[tid];
        myn = htmp[tid];
      }
      __syncthreads();
      if (tid < nhr) {
        hrow[r] = (uint16_t)(mine >> 48);
        hn[r] = __longlong_as_double((long long)myn);
      }
      __syncthreads();
    }
#endif
    const uint32_t padded = (n_light + 63u) & ~63u;
    for (uint32_t i = tid; i < padded; i += 1024) items[i] = i < n_light ? sorted[i] : (uint16_t)PLN_SENTINEL;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// per-step evaluation
// ---------------------------------------------------------------------------------------------
// Per tile, the contexts that hold any count (nrow != 0), ascending, behind their number: kernels whose per-context work is
// heavy (the linear head's softmax and its backward) run over this list instead of over all rows of the tile.
__global__ __launch_bounds__(1024) void plan_live_kernel(const pln_tile *__restrict__ tiles, uint64_t n_tiles,
                                                         const unsigned char *__restrict__ stream, uint16_t *__restrict__ live) {
  constexpr int RPT = (PLN_RMAX + 1023) / 1024;
  __shared__ uint32_t cnt[RPT * 16];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const pln_tile ti = tiles[t];
    const uint32_t rows = ti.rows_items >> 16;
    const uint8_t *nrow = stream + (size_t)ti.off16 * 16 + 64;
    uint16_t *out = live + t * PLN_LIVE_STRIDE;
    unsigned long long mask[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const uint32_t lr = tid + 1024u * k;
      mask[k] = __builtin_amdgcn_ballot_w64(lr < rows && nrow[lr] != 0);
      if (lane == 0) cnt[k * 16 + wave] = (uint32_t)__builtin_popcountll(mask[k]);
    }
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      uint32_t base = total;
      for (int w = 0; w < 16; ++w) {
        const uint32_t c = cnt[k * 16 + w];
        if (w < (int)wave) base += c;
        total += c;
      }
      if ((mask[k] >> lane) & 1ull) out[1 + base + (uint32_t)__builtin_popcountll(mask[k] & ((1ull << lane) - 1ull))] = (uint16_t)(tid + 1024u * k);
    }
    if (tid == 0) out[0] = (uint16_t)total;
    __syncthreads();
  }
}

struct pln_view {  // device-side view of a plan
  const pln_tile *tiles;
  const unsigned char *stream;
  const pln_heavy_col *heavy_col;
  const pln_heavy_row *heavy_row;
  const uint64_t *heavy_stop;
  const unsigned long long *hist;  // [0..31] contexts with total n = j+1, [32..63] with stop count j+1 (<= SRT_CL)
  const unsigned long long *hist_big;   // [n], SRT_CL < n <= PLN_NBIG: contexts with that total (also in the lists); NULL: this launch does not add them
  int big_in_hist;                 // the listed rows with a total <= PLN_NBIG are accounted for by the histogram (this launch's, or -- a
                                   // step of two launches over subsets of the tiles -- its sibling's): kernels that take it skip them
  uint64_t n_tiles, n_heavy_col, n_heavy_row, n_heavy_stop;
  const uint16_t *live;            // [n_tiles][PLN_LIVE_STRIDE] (five-column plans): [0] = contexts with counts, then their rows, ascending
  const uint16_t *live2;           // [n_tiles][LIN_LIVE2_STRIDE] or NULL: the paired form of `live` for one set of k-mers (kernels_linear.h)
  int subset;                      // `tiles` is a subset of the plan's tiles: a descriptor's spare word holds (tile number << 32 | list length)
};

// Context terms of the rows with a total in (SRT_CL, PLN_NBIG] from the plan's histogram, for kernels whose contexts share the
// concentration total A: acc_D -= m D(A, n), acc_P += scale m P(A, n), the bins dealt over the launch's threads.
__device__ __forceinline__ void pln_big_totals(const pln_view &pv, double A, double scale, uint64_t gtid, uint64_t gsz, const double2 *logtab,
                                               double &acc_D, double &acc_P) {
  if (!pv.hist_big) return;
  for (uint64_t n = SRT_CL + 1 + gtid; n <= PLN_NBIG; n += gsz) {
    const unsigned long long m = pv.hist_big[n];
    if (m) {
      const bear_dp o = srt_general_fast(A, (double)n, logtab);
      acc_D -= (double)m * o.D;
      acc_P = __builtin_fma(scale * (double)m, o.P, acc_P);
    }
  }
}
// ... and whether a listed row is one of them (then the histogram has it)
__device__ __forceinline__ bool pln_in_big_hist(const pln_view &pv, double n) { return pv.big_in_hist && n <= (double)PLN_NBIG; }

// DMA of `bytes` (multiple of 16) to LDS: 1 KiB pieces round-robin over the waves starting at wave
// `first` (so successive slabs spread over different waves), the last piece with surplus lanes masked.
// Returns the number of DMA instructions this wave issued.  Inline asm on purpose: see kernels_sorted.h.
__device__ __forceinline__ uint32_t pln_dma(void *lds, const void *src, uint32_t bytes, uint32_t wave, uint32_t lane,
                                            uint32_t first) {
  const uint32_t d = (uint32_t)(uintptr_t)lds;
  const unsigned char *s = static_cast<const unsigned char *>(src) + lane * 16u;
  const uint32_t pieces = (bytes + 1023u) >> 10;
  uint32_t issued = 0;
  for (uint32_t piece = (wave + PLN_WAVES - (first % PLN_WAVES)) % PLN_WAVES; piece < pieces; piece += PLN_WAVES) {
    const unsigned char *g = s + (piece << 10);
    const uint32_t m = srt_uniform(d + (piece << 10));
    if ((piece << 10) + lane * 16u < bytes)
      {
      // M0 is compiler-reserved: saved and restored inside the statement that uses it (no "m0" clobber: that is undefined behaviour)
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0)
                   : "v"(g), "s"(m)
                   : "memory");
    }
    ++issued;
  }
  return issued;
}

// One 1 KiB piece (`piece` = KiB index inside a slab of `bytes` bytes, multiple of 16) as a single DMA instruction.
__device__ __forceinline__ void pln_dma_piece(void *lds, const void *src, uint32_t bytes, uint32_t piece, uint32_t lane) {
  const uint32_t m = srt_uniform((uint32_t)(uintptr_t)lds + (piece << 10));
  const unsigned char *g = static_cast<const unsigned char *>(src) + (piece << 10) + lane * 16u;
  if ((piece << 10) + lane * 16u < bytes)
    {
      // M0 is compiler-reserved: saved and restored inside the statement that uses it (no "m0" clobber: that is undefined behaviour)
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0)
                   : "v"(g), "s"(m)
                   : "memory");
    }
}

// Waits until at most `younger` of this wave's vector-memory operations are outstanding (vmcnt is an
// immediate, hence the ladder; a wave issues at most 3 DMA pieces per tile).
__device__ __forceinline__ void pln_wait_all_but(uint32_t younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Scalar (s_load) fetch of a tile descriptor: a vector-memory load here would make the compiler
// wait for vmcnt(0) at its first use -- and vmcnt is in order, so that wait would drain the DMA ring.
__device__ __forceinline__ pln_tile pln_load_tile(const pln_view &pv, uint64_t t) {
  pln_tile ti;
  ti.row0 = 0;
  ti.rows_items = 0;
  ti.off16 = 0;
  ti.hc_hr = 0;
  ti.blk16 = 0;
  ti.pad = 0;
  if (t < pv.n_tiles) {
    const __attribute__((address_space(4))) pln_tile *tc = (const __attribute__((address_space(4))) pln_tile *)(uintptr_t)pv.tiles;
    ti.row0 = tc[t].row0;
    ti.rows_items = tc[t].rows_items;
    ti.off16 = tc[t].off16;
    ti.hc_hr = tc[t].hc_hr;
    ti.blk16 = tc[t].blk16;
    ti.pad = tc[t].pad;       // (subset launches of the linear step, kernels_linear.h; zero in the plan's own array)
  }
  return ti;
}

// Dynamic work distribution inside a tile: every wave draws tickets from an LDS counter.
// Lane 0's `ds_add_rtn_u32` is spelled out: of an atomicAdd under `if (lane == 0)` the compiler makes its wave-aggregated form
// (count the active lanes with v_mbcnt x 2 + s_bcnt1, elect one, add the count, hand every lane its own offset: nine vector
// instructions and the LDS one), which a draw by one known lane does not need.  All 64 lanes are active at every call.
__device__ __forceinline__ uint32_t pln_ticket(uint32_t *counter, uint32_t lane) {
  uint32_t t = 0;
  if (lane == 0)
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"((uint32_t)(uintptr_t)counter), "v"(1u) : "memory");
  return srt_uniform(t);
}

// Work units of a tile are DRAWN: whichever wave is free takes the next one, so which thread accumulates which items changes from
// run to run and the fp64 sums of a launch are reproducible to rounding only (their order across blocks is fixed).  The
// deterministic build (-DBEAR_DET_BUILD: libbear_hip_det.so, loaded when BEAR_AMD_DETERMINISTIC is set at import) deals them
// out instead: drawing wave `first` of `stride` takes units first, first + stride, ... -- every sum is then bit-identical from
// run to run, at the price of the dynamic balance.
#ifdef BEAR_DET_BUILD
#define PLN_FOR_UNITS(w, counter, n, first, stride) for (uint32_t w = (first); w < (n); w += (stride))
#else
#define PLN_FOR_UNITS(w, counter, n, first, stride) for (uint32_t w = pln_ticket(counter, lane); w < (n); w = pln_ticket(counter, lane))
#endif
// The same with a wave's FIRST unit dealt (drawing wave `first` of `stride` starts with unit `first`, the counter starts at `stride`:
// PLN_TICKET_START): one LDS atomic round trip less per wave and tile.
#if defined(BEAR_DET_BUILD) || defined(PLN_NO_FIRST_UNIT_DEALT)
#define PLN_FOR_UNITS_F(w, counter, n, first, stride) PLN_FOR_UNITS(w, counter, n, first, stride)
#define PLN_TICKET_START(stride) 0u
#else
#define PLN_FOR_UNITS_F(w, counter, n, first, stride) for (uint32_t w = (first); w < (n); w = pln_ticket(counter, lane))
#define PLN_TICKET_START(stride) ((uint32_t)(stride))
#endif

// (Round 6, measured and not kept: the NEXT ticket drawn in front of the current unit -- the draw's LDS round trip, ~1000 clocks on a
// CU whose LDS queue the units keep busy, then hides behind the unit -- made the linear step 4.7 % SLOWER, 0.902 against 0.861 ms:
// a wave holds the phase's last units while others idle at the barrier.  The evaluation kernel had found the same in round 4.)
// The same draw without waiting for its answer: lane 0's return value, to be made uniform (srt_uniform) when it is looked at.
__device__ __forceinline__ uint32_t pln_ticket_issue(uint32_t *counter, uint32_t lane) {
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(counter, 1u);
  return t;
}

// Counts of the 64 items of unit `un` from the tile's thresholds: an item's count is the number of
// thresholds E[c] that do not exceed its index.  Returns the lane's count (0 beyond n_light) and the
// wave-uniform range of the unit.
__device__ __forceinline__ uint32_t pln_unit_counts(const uint16_t *E, uint32_t n_light, uint32_t un, uint32_t lane,
                                                    uint32_t *cmin, uint32_t *cmax) {
  const uint32_t base = un * 64u, idx = base + lane;
  const uint32_t last = (base + 64u <= n_light ? base + 64u : n_light) - 1u;  // last occupied index (unit not empty)
  const uint32_t e = lane < 30u ? (uint32_t)E[lane] : 0xffffffffu;  // E[30], E[31]: heavy list lengths
  // (s_bcnt1 by name: the 32-bit results of __builtin_popcountll stay 64-bit values to the compiler, and with no scalar 64-bit
  // "less than" the loop test below became a v_cmp_ge_u64 behind two moves)
  uint32_t lo, hi;
  asm("s_bcnt1_i32_b64 %0, %1" : "=s"(lo) : "s"(__builtin_amdgcn_ballot_w64(e <= base)) : "scc");
  asm("s_bcnt1_i32_b64 %0, %1" : "=s"(hi) : "s"(__builtin_amdgcn_ballot_w64(e <= last)) : "scc");
  uint32_t c = lo;
  for (uint32_t k = lo; k < hi; ++k) c += idx >= (uint32_t)__builtin_amdgcn_readlane((int)e, (int)k) ? 1u : 0u;
  *cmin = base + 64u <= n_light ? lo : 0u;  // un-predicated factors only in full units
  *cmax = hi;
  return idx < n_light ? c : 0u;
}

// ---- mode N ---------------------------------------------------------------------------------
struct pln_buf_n {
  double pri[PLN_RMAX * 5 + 2];                      // [PLN_SENTINEL] = 1.0
  __attribute__((aligned(16))) unsigned char blk[PLN_BLOCK_MAX];  // E[32] u16 | nrow u8[rows~16] | items u16[n_light~64]
};
struct pln_lds_n {
  pln_buf_n buf[PLN_NBUF];
  double2 logtab[BEAR_LOGTAB_N];
  double tabD[SRT_NKEY];  // D(u + 5 eps, j + 1)
  double tabP[SRT_NKEY];
  uint32_t ticket[PLN_NBUF];  // per ring slot; zeroed one tile ahead
  __attribute__((aligned(16))) pln_tile desc[2][PLN_DESC_CHUNK];  // descriptors of this block's tile range, 2 x 32
  __attribute__((aligned(16))) uint32_t pf_scratch[64];           // where the L2 prefetch's dwords land (never read)
};

// Tile descriptor j of the block's range from the LDS ring (wave-uniform: every lane reads the same address).
__device__ __forceinline__ pln_tile pln_desc(const pln_tile (*ring)[PLN_DESC_CHUNK], uint64_t j, uint64_t count) {
  pln_tile ti;
  ti.row0 = 0;
  ti.rows_items = 0;
  ti.off16 = 0;
  ti.hc_hr = 0;
  ti.blk16 = 0;
  ti.pad = 0;
  if (j < count) {
    const uint32_t *d = reinterpret_cast<const uint32_t *>(&ring[(j >> 5) & 1u][j & 31u]);
    const uint32_t lo = srt_uniform(d[0]), hi = srt_uniform(d[1]);
    ti.row0 = ((uint64_t)hi << 32) | lo;
    ti.rows_items = srt_uniform(d[2]);
    ti.off16 = srt_uniform(d[3]);
    ti.hc_hr = srt_uniform(d[4]);
    ti.blk16 = srt_uniform(d[5]);
  }
  return ti;
}

// NORM: the caller asserts that every prior row sums to one (true for every ar_func of the reference,
// all of which end in a softmax, ar_funcs.py:44,97,121-126).  Then A = u + 5 eps for every context
// and the context terms collapse to the plan's histogram over n: no per-context pass at all.
// AR: multinomial mode (train_ar, core.py:138-139 with probs = prior + eps, bear_net.py:68): sum LL =
// sum over cells c log(prior + eps); no context terms, no h gradient.
template <bool NORM, bool AR>
__global__ __launch_bounds__(PLN_THREADS, 4) void dm_prior_plan_kernel(const double *__restrict__ prior,
                                                                                    uint64_t n_rows, bear_params prm_arg,
                                                                                    pln_view pv,
                                                                                    const double2 *__restrict__ logtab_g,
                                                                                    double *__restrict__ partials,
                                                                                    const bear_step_io io
#ifdef PLN_STAMPS
                                                                                    , unsigned long long *__restrict__ dbg
#endif
                                                                                    ) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_n &S = *reinterpret_cast<pln_lds_n *>(srt_smem);
  const bear_params prm = bear_params_of(prm_arg, io);   // device-resident parameters: steps enqueued without a host round trip
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  double acc[2] = {0.0, 0.0};

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < SRT_NKEY) {
    const bear_dp o = srt_general_fast(u + eps5, (double)(tid + 1), logtab_g);
    S.tabD[tid] = o.D;
    S.tabP[tid] = o.P;
  }
  if (tid < PLN_NBUF) {
    S.buf[tid].pri[PLN_SENTINEL] = 1.0;
    S.ticket[tid] = PLN_TICKET_START(PLN_WAVES - PLN_DMA_WAVES);
  }

  // One wave of the block (the last) is the DMA wave: it streams the tiles into the LDS ring and never computes.
  // Measured on MI355X (scripts/dev/dma_issue_bench.hip): with HBM saturated every vector-memory instruction
  // blocks its wave for ~700 clocks at issue -- spread over all waves that was 20-25 % of the kernel -- while a
  // single wave per CU with <= 24 pieces in flight already sustains 5.8 TB/s.
  const bool dma_wave = wave >= PLN_WAVES - PLN_DMA_WAVES;
  const uint32_t dw = wave - (PLN_WAVES - PLN_DMA_WAVES);  // index among the DMA waves
  auto stage = [&](const pln_tile &ti, uint32_t b) {  // DMA wave only: every piece of tile `ti` into ring slot `b`
    const uint32_t rows = ti.rows_items >> 16;
    if (rows == 0) return;
    const uint32_t pbytes = (rows * 40u) & ~15u, bbytes = ti.blk16 * 16u;
    const unsigned char *psrc = reinterpret_cast<const unsigned char *>(prior + ti.row0 * 5);
    const unsigned char *bsrc = pv.stream + (size_t)ti.off16 * 16;
    for (uint32_t pc = dw; (pc << 10) < bbytes; pc += PLN_DMA_WAVES) pln_dma_piece(S.buf[b].blk, bsrc, bbytes, pc, lane);
    for (uint32_t pc = dw; (pc << 10) < pbytes; pc += PLN_DMA_WAVES) pln_dma_piece(S.buf[b].pri, psrc, pbytes, pc, lane);
    if (dw == 0 && ((rows * 40u) & 15u)) {  // odd row count (last tile only): the trailing 8 bytes through the SCALAR path --
      // a vector load here would be followed by s_waitcnt vmcnt(0), which drains the DMA queue
      const __attribute__((address_space(4))) double *tail =
          (const __attribute__((address_space(4))) double *)(uintptr_t)(prior + (ti.row0 + rows) * 5 - 1);
      const double v = *tail;
      if (lane == 0) S.buf[b].pri[rows * 5 - 1] = v;
    }
  };

  // The ring holds ONE tile in flight per CU (LDS): the bytes in flight over the chip fall short of what the HBM latency asks for
  // (5.7 TB/s on the 44 B moved, against 6.0 for a plain read stream).  Where the tile's compute is light -- rows asserted
  // normalised (no context pass) and the multinomial mode -- the DMA waves therefore also touch the first 24 KiB of the tile
  // AFTER next: one lane per 128-byte line, a dword each, landing in a scratch word of LDS; that tile's DMA, one iteration
  // later, finds those lines in the L2 / infinity cache: 0.778 -> 0.715 ms (NORM), 0.768 -> 0.705 ms (multinomial) per 1e8
  // contexts, 6.2 TB/s on the bytes moved.  NOT in the general form: its tiles are bound by their compute (the context pass),
  // and the card trades shader clock for memory power -- with the prefetch its item units ran 10 % slower (s_memtime: 2045 ->
  // 2010 ticks per us) and the kernel 0.775 -> 0.81 ms.  Returns the number of instructions this wave issued (they are younger
  // than the tile's DMA pieces: the wait for the tile leaves exactly that many outstanding).
  constexpr uint32_t PF_INSTR = (NORM || AR) ? PLN_PREFETCH_KIB / 8u : 0u;
  auto prefetch_behind = [&](const pln_tile &ti) -> uint32_t {
    const uint32_t rows = ti.rows_items >> 16;
    if (PF_INSTR == 0 || rows == 0) return 0u;
    const uint64_t end_bytes = n_rows * 40ull, base = (ti.row0 + rows) * 40ull;
    const uint32_t m = srt_uniform((uint32_t)(uintptr_t)S.pf_scratch);
    uint32_t issued = 0;
    for (uint32_t k = dw; k < PF_INSTR; k += PLN_DMA_WAVES) {
      uint64_t off = base + ((uint64_t)(k * 64u + lane) << 7);
      if (off + 4u > end_bytes) off = end_bytes - 4u;
      const unsigned char *g = reinterpret_cast<const unsigned char *>(prior) + off;
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0)
                   : "v"(g), "s"(m)
                   : "memory");
      ++issued;
    }
    return issued;
  };
  uint32_t pf_young = 0;   // prefetch instructions issued behind the DMA of the tile that lands next
#ifdef PLN_STAMPS
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
  int prev_kind = 0;
#define PLN_STAMP(k)                                              \
  {                                                               \
    const unsigned long long now = __builtin_amdgcn_s_memtime();  \
    tph[k] += now - t_prev;                                       \
    t_prev = now;                                                 \
  }
#else
#define PLN_STAMP(k)
#endif
  // Each block owns a contiguous range of tiles; their descriptors come through LDS, 32 per DMA piece (no scalar
  // global loads in the loop: they would sit in lgkmcnt in front of every LDS access).
  static_assert(PLN_NBUF == 2, "the DMA-wave protocol below is written for a two-slot ring");
  const uint64_t G = gridDim.x;
  const uint64_t first = (pv.n_tiles * (uint64_t)blockIdx.x) / G, count = (pv.n_tiles * ((uint64_t)blockIdx.x + 1)) / G - first;
  auto fetch_desc = [&](uint64_t j0) {  // DMA wave: descriptors [j0, j0 + 32) of the range -> ring half (j0 / 32) & 1
    if (dw == 0 && j0 < count)
      pln_dma_piece(S.desc[(j0 >> 5) & 1u], pv.tiles + first + j0, PLN_DESC_CHUNK * (uint32_t)sizeof(pln_tile), 0u, lane);
  };
  if (dma_wave) {
    __builtin_amdgcn_s_setprio(3);  // its few instructions per piece go ahead of the compute waves of the same SIMD
    fetch_desc(0);
    fetch_desc(PLN_DESC_CHUNK);
    srt_wait_dma();
  }
  __syncthreads();
  pln_tile cur = pln_desc(S.desc, 0, count), nxt = pln_desc(S.desc, 1, count);
  if (dma_wave) stage(cur, 0);
  uint32_t slot = 0;
  for (uint64_t j = 0; j < count; ++j) {
#ifdef PLN_STAMPS
    PLN_STAMP(prev_kind)  // last work item of the previous tile + the failed ticket draw
    prev_kind = 4;        // slot 4: first ticket draw of a tile
#endif
    if (dma_wave) pln_wait_all_but(pf_young);  // tile j has landed (the prefetch issued behind it may still be in flight)
    PLN_STAMP(1)
    srt_sync();  // ... and every compute wave is done with tile j - 1: its slot is free
    PLN_STAMP(2)  // barrier
    const pln_tile nn = pln_desc(S.desc, j + 2, count);
    if (dma_wave) {
      if ((j & (PLN_DESC_CHUNK - 1)) == 0 && j != 0) fetch_desc(j + PLN_DESC_CHUNK);  // the ring half just left behind
      stage(nxt, slot ^ 1u);
      pf_young = prefetch_behind(nxt);
      cur = nxt;
      nxt = nn;
      slot ^= 1u;
      continue;
    }
    PLN_STAMP(3)

    const pln_buf_n &B = S.buf[slot];
    const uint32_t rows = cur.rows_items >> 16, n_light = cur.rows_items & 0xffffu;
    const uint32_t hc = cur.hc_hr >> 16, hr = cur.hc_hr & 0xffffu;
    const pln_layout L = pln_block_layout(rows, n_light, hc, hr);
    const uint16_t *E = reinterpret_cast<const uint16_t *>(B.blk);
    const uint8_t *nrow = B.blk + L.nrow;
    const uint16_t *items = reinterpret_cast<const uint16_t *>(B.blk + L.items);
    if (tid == 0) S.ticket[(slot + 1) % PLN_NBUF] = PLN_TICKET_START(PLN_WAVES - PLN_DMA_WAVES);  // next tile's counter (its last readers passed the barrier above)
    // Work list of the tile, dearest first: the large-count column items and contexts (Stirling path), the
    // item units from the sorted tail down (long loops), then the 64-context chunks of the context terms.
    // Waves draw tickets until the list is exhausted.
    const uint32_t n_hcu = (hc + 63u) >> 6, n_hru = AR ? 0u : (hr + 63u) >> 6, n_heavy = n_hcu + n_hru;
    const uint32_t n_units = (n_light + 63u) >> 6;
    const uint32_t n_work = n_heavy + n_units + ((NORM || AR) ? 0u : (rows + PLN_CHUNK - 1u) / PLN_CHUNK);
    PLN_FOR_UNITS_F(w, &S.ticket[slot], n_work, wave, PLN_WAVES - PLN_DMA_WAVES) {      // (the DMA waves never get here)
#ifdef PLN_STAMPS
      {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        tph[prev_kind] += now - t_prev;  // the previous work item incl. the ticket draw that followed it
        t_prev = now;
        prev_kind = w < n_heavy ? 5 : (w < n_heavy + n_units ? 6 : 7);
      }
#endif
      if (w < n_hcu) {  // large-count column items of this tile
        const uint32_t i = w * 64u + lane;
        if (i < hc) {
          const uint32_t off = reinterpret_cast<const uint16_t *>(B.blk + L.hoff)[i];
          const double cnt = (double)reinterpret_cast<const uint32_t *>(B.blk + L.hcnt)[i];
          if (AR) {
            acc[0] = __builtin_fma(cnt, bear_log_tab(B.pri[off] + eps, S.logtab), acc[0]);
          } else {
            const double x = __builtin_fma(B.pri[off], u, eps);
            const bear_dp o = srt_general_fast(x, cnt, S.logtab);
            acc[0] += o.D;
            acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
          }
        }
        continue;
      }
      if (w < n_heavy) {  // contexts of this tile with a large total
        const uint32_t i = (w - n_hcu) * 64u + lane;
        if (i < hr) {
          const double *f = &B.pri[(uint32_t)reinterpret_cast<const uint16_t *>(B.blk + L.hrow)[i] * 5u];
          const double A = NORM ? u + eps5 : __builtin_fma(((f[0] + f[1]) + (f[2] + f[3])) + f[4], u, eps5);
          const bear_dp o = srt_general_fast(A, reinterpret_cast<const double *>(B.blk + L.hn)[i], S.logtab);
          acc[0] -= o.D;
          acc[1] = __builtin_fma(A - eps5, o.P, acc[1]);
        }
        continue;
      }
      if (NORM || AR || w < n_heavy + n_units) {
        // ---- D: one unit of column items
        const uint32_t un = n_heavy + n_units - 1u - w;
        uint32_t cmin, cmax;
        const uint32_t ci[1] = {pln_unit_counts(E, n_light, un, lane, &cmin, &cmax)};
        const uint32_t off = items[un * 64u + lane];
        if (AR) {
          acc[0] = __builtin_fma((double)ci[0], bear_log_tab(B.pri[off] + eps, S.logtab), acc[0]);
          continue;
        }
        const double x[1] = {__builtin_fma(B.pri[off], u, eps)};
        bear_dp o[1];
        srt_light<1>(x, ci, cmin, cmax, S.logtab, o);
        acc[0] += o[0].D;
        acc[1] = __builtin_fma(eps - x[0], o[0].P, acc[1]);
        continue;
      }
      // ---- A: context terms  -D(A, n), (A - 5 eps) P(A, n)   with A = S u + 5 eps; PLN_CHUNK contexts per ticket, a lane
      // takes two ADJACENT contexts: their ten doubles are 80 contiguous, 16-byte aligned bytes = five 16-byte LDS reads
      // (conflict-free at this stride) instead of ten 8-byte ones
      static_assert(PLN_CHUNK == 128, "two adjacent contexts per lane");
      const uint32_t row0c = (w - n_heavy - n_units) * PLN_CHUNK + 2u * lane;
      double S5[2];
      uint32_t nn_[2];
      {
        const uint32_t rr = row0c < rows ? row0c : 0u;          // rows is a multiple of 4 except in the table's last tile: a
        const double2 *src = reinterpret_cast<const double2 *>(&B.pri[rr * 5]);   // pair may end one row past it (still inside pri)
        const double2 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3], v4 = src[4];
        S5[0] = ((v0.x + v0.y) + (v1.x + v1.y)) + v2.x;
        S5[1] = ((v2.y + v3.x) + (v3.y + v4.x)) + v4.y;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const uint32_t n = row0c + q < rows ? (uint32_t)nrow[rr + q] : 0u;  // 0: empty context; 255: total beyond SRT_CL (heavy lists)
          nn_[q] = n == 255u ? 0u : n;
        }
      }
#pragma unroll
      for (int q = 0; q < PLN_CHUNK / 64; ++q) {
        const uint32_t n = nn_[q];
        const bool shared = __builtin_fabs(S5[q] - 1.0) <= SRT_SUM1_TOL;
        if (n != 0 && shared) {
          acc[0] -= S.tabD[n - 1];
          acc[1] = __builtin_fma(u, S.tabP[n - 1], acc[1]);
        }
        const uint32_t own = (n != 0 && !shared) ? n : 0u;  // general concentrations: own A
        if (__builtin_amdgcn_ballot_w64(own != 0)) {
          uint32_t cm = own;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o2 = (uint32_t)__shfl_xor((int)cm, off, 64);
            cm = o2 > cm ? o2 : cm;
          }
          const double xa[1] = {own ? __builtin_fma(S5[q], u, eps5) : 1.0};
          const uint32_t ca[1] = {own};
          bear_dp o[1];
          srt_light<1>(xa, ca, 0u, srt_uniform(cm), S.logtab, o);
          acc[0] -= o[0].D;
          acc[1] = __builtin_fma(xa[0] - eps5, o[0].P, acc[1]);
        }
      }
    }
    cur = nxt;
    nxt = nn;
    slot ^= 1u;
  }
  srt_wait_dma();
#ifdef PLN_STAMPS
  if (lane == 0 && dbg) {
    for (int k = 0; k < 8; ++k) dbg[((size_t)blockIdx.x * PLN_WAVES + wave) * 8 + k] = tph[k];
  }
#endif
  // ---- Stirling-path items of the whole table, densely packed over the grid
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    if (AR) {
      acc[0] = __builtin_fma((double)h.c, bear_log_tab(prior[h.off] + eps, S.logtab), acc[0]);
      continue;
    }
    const double x = __builtin_fma(prior[h.off], u, eps);
    const bear_dp o = srt_general_fast(x, (double)h.c, S.logtab);
    acc[0] += o.D;
    acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
  }
  for (uint64_t i = gtid; !AR && i < pv.n_heavy_row; i += gsz) {
    const pln_heavy_row h = pv.heavy_row[i];
    const double *f = prior + h.row * 5;
    const double A = __builtin_fma(((f[0] + f[1]) + (f[2] + f[3])) + f[4], u, eps5);
    const bear_dp o = srt_general_fast(A, h.n, S.logtab);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(A - eps5, o.P, acc[1]);
  }
  if (NORM && !AR && blockIdx.x == 0 && tid < SRT_CL) {  // context terms with the shared A, weighted by their multiplicity
    const double m = (double)pv.hist[tid];
    acc[0] -= m * S.tabD[tid];
    acc[1] = __builtin_fma(u * m, S.tabP[tid], acc[1]);
  }
  __syncthreads();
  block_finish<2>(acc, partials, io);
}

// ---- mode R ---------------------------------------------------------------------------------
struct pln_buf_r {
  uint32_t ref[PLN_RMAX * 5 + 4];  // [PLN_SENTINEL .. +3] = 0
  __attribute__((aligned(16))) unsigned char blk[PLN_BLOCK_MAX];
};
#define PLN_RSLOT 3   // mode R ring: a tile being finished, the tile being worked on, a tile landing (41 KB each)
struct pln_lds_r {
  pln_buf_r buf[PLN_RSLOT];
  double2 logtab[BEAR_LOGTAB_N];
  double tabD[2][SRT_NKEY];  // [0]: context term (x = A), [1]: stop column (x = x4)
  double tabP[2][SRT_NKEY];
  uint32_t meta[PLN_RSLOT][2];   // rows_items, hc_hr of the tile in each slot (written by the DMA wave that staged it)
  uint32_t ticket[PLN_RSLOT];    // work tickets of the tile in the slot
  uint32_t landed[PLN_RSLOT];    // += 1 by the DMA wave that streamed a tile once it is in LDS: the slot's g-th tile is there at g + 1
  uint32_t left[PLN_RSLOT];      // += 1 by each compute wave that has no more work in the slot's tile: free again at 14 (g + 1)
};
__device__ __forceinline__ uint32_t pln_peek(const uint32_t *p) {
  return srt_uniform(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

// AR: multinomial mode of bear_ref (train_ar): sum LL = sum c log(f + eps); gradients w.r.t. tau_s, nu_s only.
template <bool AR>
__global__ __launch_bounds__(PLN_THREADS, 4) void dm_ref_plan_kernel(const uint32_t *__restrict__ ref,
                                                                                  uint64_t n_rows, bear_params prm_arg,
                                                                                  pln_view pv,
                                                                                  const double2 *__restrict__ logtab_g,
                                                                                  double *__restrict__ partials,
                                                                                  const bear_step_io io, const bear_apply_io apply) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_r &S = *reinterpret_cast<pln_lds_r *>(srt_smem);
  // parameters by value, or -- for a step that is replayed from a HIP graph while the optimizer moves them -- from device memory
  const bear_params prm = bear_params_of(prm_arg, io);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
  const double u = prm.inv_h, eps = prm.eps;
  const double A = u + 5.0 * eps;              // sum_b alpha_b
  const double x4 = prm.nw * prm.V * u + eps;  // alpha of the stop column
  const double VU = prm.V * u;
  const double tau = prm.tau;
  const double w2c = tau * (eps + 0.25 * VU);  // d alpha/d tau_s = -tau x + w2c
  const double nwV = prm.nw * prm.V;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < 2 * SRT_NKEY) {
    const int which = tid / SRT_NKEY, j = tid % SRT_NKEY;
    const bear_dp o = srt_general_fast(which ? x4 : A, (double)(j + 1), logtab_g);
    S.tabD[which][j] = o.D;
    S.tabP[which][j] = o.P;
  }
  if (tid < PLN_RSLOT) {
    S.landed[tid] = S.left[tid] = 0u;
    S.ticket[tid] = PLN_TICKET_START(PLN_WAVES - PLN_DMA_WAVES);
  }
  if (tid < 4 * PLN_RSLOT) S.buf[tid >> 2].ref[PLN_SENTINEL + (tid & 3)] = 0;  // neutral cell: reference row of zeros

  // Streaming protocol WITHOUT a workgroup barrier (measured: with dm_prior_plan_kernel's two-slot ring and one barrier per
  // tile this kernel ran at the tile rate of the streaming skeleton -- 2.7 us per tile and CU, 3.5 TB/s -- because a tile is
  // only 37 KB: one tile in flight per CU cannot cover the HBM latency).  Three slots; the last PLN_DMA_WAVES waves only move
  // data, taking alternate tiles (each wave all pieces of its tile, published through `landed` the moment they are in LDS);
  // a compute wave that finds a tile's tickets exhausted adds itself to `left` and moves on; a slot is refilled when all
  // compute waves have left it.  Up to two tiles are in flight per CU and nobody waits for the slowest wave of a tile.
  const bool dma_wave = wave >= PLN_WAVES - PLN_DMA_WAVES;
  const uint32_t dw = wave - (PLN_WAVES - PLN_DMA_WAVES);
  constexpr uint32_t CWAVES = PLN_WAVES - PLN_DMA_WAVES;
  auto stage = [&](const pln_tile &ti, uint32_t b) {   // a DMA wave: every piece of tile `ti` into ring slot `b`
    const uint32_t rows = ti.rows_items >> 16;
    const uint32_t rbytes = rows * 20u, rb16 = rbytes & ~15u, bbytes = ti.blk16 * 16u;
    const unsigned char *rsrc = reinterpret_cast<const unsigned char *>(ref + ti.row0 * 5);
    const unsigned char *bsrc = pv.stream + (size_t)ti.off16 * 16;
    for (uint32_t pc = 0; (pc << 10) < bbytes; ++pc) pln_dma_piece(S.buf[b].blk, bsrc, bbytes, pc, lane);
    for (uint32_t pc = 0; (pc << 10) < rb16; ++pc) pln_dma_piece(S.buf[b].ref, rsrc, rb16, pc, lane);
    if (rbytes & 15u) {  // row count not a multiple of 4 (last tile only): trailing dwords, SCALAR path
      const uint32_t w0 = rb16 >> 2, nw = (rbytes & 15u) >> 2;
      const __attribute__((address_space(4))) uint32_t *tail =
          (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)(ref + ti.row0 * 5 + w0);
      const uint32_t v0 = tail[0], v1 = nw > 1 ? tail[1] : 0u, v2 = nw > 2 ? tail[2] : 0u;
      if (lane == 0) {
        S.buf[b].ref[w0] = v0;
        if (nw > 1) S.buf[b].ref[w0 + 1] = v1;
        if (nw > 2) S.buf[b].ref[w0 + 2] = v2;
      }
    }
    if (lane == 0) {
      S.meta[b][0] = ti.rows_items;
      S.meta[b][1] = ti.hc_hr;
      S.ticket[b] = PLN_TICKET_START(PLN_WAVES - PLN_DMA_WAVES);
    }
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
  // AR mode: item (ref count rb, row total R, count c): f = (1/4 + E dev) V, dLL/df = c / (f + eps)
  auto accumulate_ar = [&](double rb, double R, double c) {
    const double dev = __builtin_fma(rb + eps, bear_rcp(R), -0.25);
    const double f = __builtin_fma(prm.E, dev, 0.25) * prm.V;
    const double p = f + eps;
    const double dLdf = c * bear_rcp(p);
    acc[0] = __builtin_fma(c, bear_log_tab(p, S.logtab), acc[0]);
    acc[2] = __builtin_fma(dLdf, -prm.tauE * dev * prm.V, acc[2]);  // d f / d tau_s
    acc[3] = __builtin_fma(dLdf, -nwV * f, acc[3]);                 // d f / d nu_s (net function is 0 here)
  };

  const uint64_t G = gridDim.x;
  const uint64_t first = (pv.n_tiles * (uint64_t)blockIdx.x) / G, count = (pv.n_tiles * ((uint64_t)blockIdx.x + 1)) / G - first;
  __syncthreads();
  if (dma_wave) {
    __builtin_amdgcn_s_setprio(3);
    pln_tile ti = pln_load_tile(pv, dw < count ? first + dw : pv.n_tiles);   // descriptors by scalar loads, one tile ahead
    for (uint64_t j = dw; j < count; j += PLN_DMA_WAVES) {
      const uint32_t b = (uint32_t)(j % PLN_RSLOT), gen = (uint32_t)(j / PLN_RSLOT);
      const pln_tile ti_next = pln_load_tile(pv, j + PLN_DMA_WAVES < count ? first + j + PLN_DMA_WAVES : pv.n_tiles);
      while (pln_peek(&S.left[b]) < CWAVES * gen) __builtin_amdgcn_s_sleep(1);
      stage(ti, b);
      srt_wait_dma();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(&S.landed[b], 1u);
      ti = ti_next;
    }
  }
  for (uint64_t j = 0; !dma_wave && j < count; ++j) {
    const uint32_t slot = (uint32_t)(j % PLN_RSLOT), gen = (uint32_t)(j / PLN_RSLOT);
    while (pln_peek(&S.landed[slot]) < gen + 1u) __builtin_amdgcn_s_sleep(1);
    const pln_buf_r &B = S.buf[slot];
    const uint32_t rows_items = srt_uniform(S.meta[slot][0]), hc_hr = srt_uniform(S.meta[slot][1]);
    const uint32_t rows = rows_items >> 16, n_light = rows_items & 0xffffu;
    const uint32_t hc = hc_hr >> 16;
    const pln_layout L = pln_block_layout(rows, n_light, hc, hc_hr & 0xffffu);
    const uint16_t *E = reinterpret_cast<const uint16_t *>(B.blk);
    const uint16_t *items = reinterpret_cast<const uint16_t *>(B.blk + L.items);
    // The context term (x = A) and the stop column (x = x4) have the same concentration in every
    // context: their sums over the table are the plan's histograms times two small tables (added
    // once, after the loop).  Per tile only the column items b < 4 remain; waves draw units, dearest first.
    const uint32_t n_hcu = (hc + 63u) >> 6;
    const uint32_t n_units = (n_light + 63u) >> 6;
    PLN_FOR_UNITS_F(w, &S.ticket[slot], n_hcu + n_units, wave, CWAVES) {      // (compute waves only)
      if (w < n_hcu) {  // large-count column items of this tile (Stirling path), first
        const uint32_t i = w * 64u + lane;
        if (i < hc) {
          const uint32_t off = reinterpret_cast<const uint16_t *>(B.blk + L.hoff)[i];
          const uint32_t *rr = &B.ref[((off * 52429u) >> 18) * 5u];
          const double R = (double)(((uint64_t)rr[0] + rr[1]) + ((uint64_t)rr[2] + rr[3])) + 4.0 * eps;
          const double cnt = (double)reinterpret_cast<const uint32_t *>(B.blk + L.hcnt)[i];
          if (AR) {
            accumulate_ar((double)B.ref[off], R, cnt);
          } else {
            const double x = alpha_from((double)B.ref[off], R);
            accumulate(x, srt_general_fast(x, cnt, S.logtab));
          }
        }
        continue;
      }
      const uint32_t un = n_hcu + n_units - 1u - w;
      uint32_t cmin, cmax;
      const uint32_t ci[1] = {pln_unit_counts(E, n_light, un, lane, &cmin, &cmax)};
      const uint32_t off = items[un * 64u + lane];
      const uint32_t *rr = &B.ref[((off * 52429u) >> 18) * 5u];  // row start: 5 * (off / 5), off < 2^16
      const double R = (double)(((uint64_t)rr[0] + rr[1]) + ((uint64_t)rr[2] + rr[3])) + 4.0 * eps;  // bear_ref.py:335-337, 30
      if (AR) {
        accumulate_ar((double)B.ref[off], R, (double)ci[0]);
        continue;
      }
      const double x[1] = {alpha_from((double)B.ref[off], R)};
      bear_dp o[1];
      srt_light<1>(x, ci, cmin, cmax, S.logtab, o);
      accumulate(x[0], o[0]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the slot are done before it counts as gone
    if (lane == 0) atomicAdd(&S.left[slot], 1u);
  }
  srt_wait_dma();
  // ---- Stirling-path items of the whole table
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    const uint32_t *rr = ref + (h.off / 5) * 5;
    const double R = (double)(((uint64_t)rr[0] + rr[1]) + ((uint64_t)rr[2] + rr[3])) + 4.0 * eps;
    if (AR) {
      accumulate_ar((double)ref[h.off], R, (double)h.c);
      continue;
    }
    const double x = alpha_from((double)ref[h.off], R);
    accumulate(x, srt_general_fast(x, (double)h.c, S.logtab));
  }
  if (AR) {
    // stop column: f_4 = nw V for every context, so its term is (sum of all stop counts) log(f_4 + eps)
    const double f4 = nwV, p4 = f4 + eps;
    double c4sum = 0.0;
    if (blockIdx.x == 0 && tid < SRT_CL) c4sum = (double)(tid + 1) * (double)pv.hist[SRT_NKEY + tid];
    for (uint64_t i = gtid; i < pv.n_heavy_stop; i += gsz) c4sum += (double)pv.heavy_stop[i];
    acc[0] = __builtin_fma(c4sum, bear_log_tab(p4, S.logtab), acc[0]);
    acc[3] = __builtin_fma(c4sum * bear_rcp(p4), nwV * (1.0 - f4), acc[3]);  // d f_4 / d nu_s = nw V (1 - f_4)
    __syncthreads();
    block_finish<4>(acc, partials, io, apply);
    return;
  }
  for (uint64_t i = gtid; i < pv.n_heavy_row; i += gsz) {
    const double n = pv.heavy_row[i].n;
    if (pln_in_big_hist(pv, n)) continue;      // (totals up to PLN_NBIG: the plan's histogram, next line)
    const bear_dp o = srt_general_fast(A, n, S.logtab);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(u, o.P, acc[1]);
  }
  pln_big_totals(pv, A, u, gtid, gsz, S.logtab, acc[0], acc[1]);
  for (uint64_t i = gtid; i < pv.n_heavy_stop; i += gsz) {
    const bear_dp o = srt_general_fast(x4, (double)pv.heavy_stop[i], S.logtab);
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
  block_finish<4>(acc, partials, io, apply);
}

// ---- mode N with gradient rows --------------------------------------------------------------
// d sum LL / d prior_ib = u (P_ib - P_n,i)  (bear_net.py:193: what the tape hands back to ar_func).  The rows
// are assembled in LDS next to the prior tile -- context chunks write the base -u P_n into all five cells, the
// item units add u P_b into their own cell (every cell belongs to at most one item) -- and leave as one
// coalesced stream of 16-byte lane stores.  No prefetch here: with 40 B written per context on top of the 44 B
// read the kernel is bound by HBM traffic either way.  Items that overflowed to the plan's global lists are
// applied by dm_prior_grad_fixup_kernel (stream-ordered after this kernel).
struct pln_lds_g {
  double pri[PLN_RMAX * 5 + 2];
  double grad[PLN_RMAX * 5 + 2];
  __attribute__((aligned(16))) unsigned char blk[PLN_BLOCK_MAX];
  double2 logtab[BEAR_LOGTAB_N];
  double tabD[SRT_NKEY];
  double tabP[SRT_NKEY];
  uint32_t ticket;
  __attribute__((aligned(16))) uint32_t pf_scratch[64];   // where the L2 prefetch's dwords land (never read)
};
static_assert(sizeof(pln_lds_g) <= 158 * 1024, "gradient-row kernel: LDS budget");

// One lane per 128-byte line of [src, src + bytes), a dword each into a scratch word of LDS: the lines are in the L2 when the DMA
// that wants them is issued (the kernels here hold ONE tile in flight per CU; see dm_prior_plan_kernel).  Instruction k of the
// range covers lines [64 k, 64 k + 64); nothing is waited for.
__device__ __forceinline__ void pln_touch_lines(uint32_t *scratch, const void *src, uint32_t bytes, uint32_t k, uint32_t lane) {
  const uint32_t off = (k * 64u + lane) << 7;
  if (off < bytes) {
    const unsigned char *g = static_cast<const unsigned char *>(src) + off;
    const uint32_t m = srt_uniform((uint32_t)(uintptr_t)scratch);
    uint32_t keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0)
                 : "v"(g), "s"(m)
                 : "memory");
  }
}

template <bool NORM, bool AR>
__global__ __launch_bounds__(PLN_THREADS, 4) void dm_prior_plan_grad_kernel(const double *__restrict__ prior,
                                                                                         bear_params prm_arg, pln_view pv,
                                                                                         const double2 *__restrict__ logtab_g,
                                                                                         double *__restrict__ grad_out,
                                                                                         double *__restrict__ partials,
                                                                                         const bear_step_io io) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_g &S = *reinterpret_cast<pln_lds_g *>(srt_smem);
  const bear_params prm = bear_params_of(prm_arg, io);   // device-resident parameters for HIP-graph replay
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  double acc[2] = {0.0, 0.0};
  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < SRT_NKEY) {
    const bear_dp o = srt_general_fast(u + eps5, (double)(tid + 1), logtab_g);
    S.tabD[tid] = o.D;
    S.tabP[tid] = o.P;
  }
  if (tid == 0) {
    S.pri[PLN_SENTINEL] = 1.0;
    S.ticket = PLN_TICKET_START(PLN_WAVES);
  }
  // (the descriptor of a tile is fetched one tile ahead, right behind the issue of the current tile's DMA: a scalar load shares
  // lgkmcnt with the LDS, so fetched at the top of its own tile it cost every wave a memory latency per tile -- round 4)
  pln_tile nxt = pln_load_tile(pv, blockIdx.x);
  for (uint64_t t = blockIdx.x; t < pv.n_tiles; t += gridDim.x) {
    const pln_tile cur = nxt;
    const uint32_t rows = cur.rows_items >> 16, n_light = cur.rows_items & 0xffffu;
    const uint32_t hc = cur.hc_hr >> 16, hr = cur.hc_hr & 0xffffu;
    const pln_layout L = pln_block_layout(rows, n_light, hc, hr);
    __syncthreads();  // previous tile written out (plain stores: vmcnt(0) inside __syncthreads is what we want here)
    {
      const uint32_t pbytes = rows * 40u;
      pln_dma(S.pri, prior + cur.row0 * 5, pbytes & ~15u, wave, lane, 0);
      if (pbytes & 15u) {
        const __attribute__((address_space(4))) double *tail =
            (const __attribute__((address_space(4))) double *)(uintptr_t)(prior + (cur.row0 + rows) * 5 - 1);
        const double v = *tail;
        if (tid == 0) S.pri[rows * 5 - 1] = v;
      }
      pln_dma(S.blk, pv.stream + (size_t)cur.off16 * 16, cur.blk16 * 16u, wave, lane, (pbytes + 1023u) >> 10);
    }
    if (tid == 0) S.ticket = PLN_TICKET_START(PLN_WAVES);
    nxt = pln_load_tile(pv, t + gridDim.x);      // returns while this tile's DMA is waited for
    srt_wait_dma();
    srt_sync();
    const uint16_t *E = reinterpret_cast<const uint16_t *>(S.blk);
    const uint8_t *nrow = S.blk + L.nrow;
    const uint16_t *items = reinterpret_cast<const uint16_t *>(S.blk + L.items);
    // ---- 1: context chunks: context terms and the base -u P_n of all five cells (AR mode: zeros)
    for (uint32_t row = tid; AR && row < rows; row += PLN_THREADS) {
#pragma unroll
      for (int b = 0; b < 5; ++b) S.grad[row * 5 + b] = 0.0;
    }
    for (uint32_t row = tid; !AR && row < rows; row += PLN_THREADS) {
      double f[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) f[b] = S.pri[row * 5 + b];
      const double S5 = ((f[0] + f[1]) + (f[2] + f[3])) + f[4];
      uint32_t n = nrow[row];
      if (n == 255u) n = 0u;  // large totals: step 2 / fix-up kernel
      const bool shared = NORM || __builtin_fabs(S5 - 1.0) <= SRT_SUM1_TOL;
      double Pn = 0.0;
      if (n != 0 && shared) {
        Pn = S.tabP[n - 1];
        acc[0] -= S.tabD[n - 1];
        acc[1] = __builtin_fma(u, Pn, acc[1]);
      }
      if (n != 0 && !shared) {  // own A: rare (general concentrations), evaluated lane by lane
        const double A = __builtin_fma(S5, u, eps5);
        const bear_dp o = srt_general_fast(A, (double)n, S.logtab);
        Pn = o.P;
        acc[0] -= o.D;
        acc[1] = __builtin_fma(A - eps5, o.P, acc[1]);
      }
      const double base = -u * Pn;
#pragma unroll
      for (int b = 0; b < 5; ++b) S.grad[row * 5 + b] = base;
    }
    srt_sync();
    // ---- 2: contexts of this tile with a large total overwrite their base
    for (uint32_t i = tid; !AR && i < hr; i += PLN_THREADS) {
      const uint32_t row = reinterpret_cast<const uint16_t *>(S.blk + L.hrow)[i];
      const double *f = &S.pri[row * 5u];
      const double A = NORM ? u + eps5 : __builtin_fma(((f[0] + f[1]) + (f[2] + f[3])) + f[4], u, eps5);
      const bear_dp o = srt_general_fast(A, reinterpret_cast<const double *>(S.blk + L.hn)[i], S.logtab);
      acc[0] -= o.D;
      acc[1] = __builtin_fma(A - eps5, o.P, acc[1]);
#pragma unroll
      for (int b = 0; b < 5; ++b) S.grad[row * 5 + b] = -u * o.P;
    }
    srt_sync();
    // The tile this block takes next: its prior rows and its plan block into the L2 while this one's items are computed and its
    // rows stored (the kernel is single-buffered: load, compute and store take turns, and the next load then starts from the L2):
    // 1.764 -> 1.62 ms per 1e8 contexts.  Issued here, behind the context pass: from the top of the tile the lines had to outlive
    // the whole tile in an L2 that the stores of 32 CUs stream through, and 60 % of them were fetched twice (FETCH_SIZE 70 B per
    // context; here 51 against 44 without any prefetch); before the store phase it is too late (1.90 ms).  Not in the multinomial
    // mode, whose tiles have next to no compute to hide anything under (1.51 -> 1.62).
    if (!AR && wave < 10u) {
      const pln_tile &nx = nxt;
      const uint32_t nrows = nx.rows_items >> 16;
      if (wave < 9u) pln_touch_lines(S.pf_scratch, prior + nx.row0 * 5, nrows * 40u, wave, lane);
      else pln_touch_lines(S.pf_scratch, pv.stream + (size_t)nx.off16 * 16, nx.blk16 * 16u, 0u, lane);
    }
    // ---- 3: item units (tickets, dearest first): ELBO, d/dh, and u P_b into the item's own cell
    const uint32_t n_hcu = (hc + 63u) >> 6, n_units = (n_light + 63u) >> 6;
    PLN_FOR_UNITS_F(w, &S.ticket, n_hcu + n_units, wave, PLN_WAVES) {
      if (w < n_hcu) {
        const uint32_t i = w * 64u + lane;
        if (i < hc) {
          const uint32_t off = reinterpret_cast<const uint16_t *>(S.blk + L.hoff)[i];
          const double cnt = (double)reinterpret_cast<const uint32_t *>(S.blk + L.hcnt)[i];
          if (AR) {
            const double pp = S.pri[off] + eps;
            acc[0] = __builtin_fma(cnt, bear_log_tab(pp, S.logtab), acc[0]);
            S.grad[off] = cnt * bear_rcp(pp);
          } else {
            const double x = __builtin_fma(S.pri[off], u, eps);
            const bear_dp o = srt_general_fast(x, cnt, S.logtab);
            acc[0] += o.D;
            acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
            S.grad[off] = __builtin_fma(u, o.P, S.grad[off]);
          }
        }
        continue;
      }
      const uint32_t un = n_hcu + n_units - 1u - w;
      uint32_t cmin, cmax;
      const uint32_t ci[1] = {pln_unit_counts(E, n_light, un, lane, &cmin, &cmax)};
      const uint32_t off = items[un * 64u + lane];
      if (AR) {
        const double pp = S.pri[off] + eps;
        acc[0] = __builtin_fma((double)ci[0], bear_log_tab(pp, S.logtab), acc[0]);
        if (ci[0] != 0) S.grad[off] = (double)ci[0] * bear_rcp(pp);
        continue;
      }
      const double x[1] = {__builtin_fma(S.pri[off], u, eps)};
      bear_dp o[1];
      srt_light<1>(x, ci, cmin, cmax, S.logtab, o);
      acc[0] += o[0].D;
      acc[1] = __builtin_fma(eps - x[0], o[0].P, acc[1]);
      if (ci[0] != 0) S.grad[off] = __builtin_fma(u, o[0].P, S.grad[off]);
    }
    __syncthreads();
    // ---- 4: the tile's gradient rows leave as one coalesced stream
    {
      const uint32_t n_dw = rows * 10u;  // dwords
      const uint4 *src = reinterpret_cast<const uint4 *>(S.grad);
      uint4 *dst = reinterpret_cast<uint4 *>(grad_out + cur.row0 * 5);
      for (uint32_t i = tid; i < (n_dw >> 2); i += PLN_THREADS) dst[i] = src[i];
      if ((n_dw & 3u) && tid == 0) grad_out[(cur.row0 + rows) * 5 - 1] = S.grad[rows * 5 - 1];  // odd row count
    }
  }
  // ---- ELBO / d/dh of the items that overflowed to the global lists (their gradient cells: fix-up kernel)
  __syncthreads();
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    if (AR) {
      acc[0] = __builtin_fma((double)h.c, bear_log_tab(prior[h.off] + eps, S.logtab), acc[0]);
      continue;
    }
    const double x = __builtin_fma(prior[h.off], u, eps);
    const bear_dp o = srt_general_fast(x, (double)h.c, S.logtab);
    acc[0] += o.D;
    acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
  }
  for (uint64_t i = gtid; !AR && i < pv.n_heavy_row; i += gsz) {
    const pln_heavy_row h = pv.heavy_row[i];
    const double *f = prior + h.row * 5;
    const double A = NORM ? u + eps5 : __builtin_fma(((f[0] + f[1]) + (f[2] + f[3])) + f[4], u, eps5);
    const bear_dp o = srt_general_fast(A, h.n, S.logtab);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(A - eps5, o.P, acc[1]);
  }
  __syncthreads();
  block_finish<2>(acc, partials, io);
}

// ---- the same for rows the caller asserts NORMALISED (every ar_func of the reference ends in a softmax): the gradient block
// is formed IN PLACE over the prior rows, which frees the LDS for a second row buffer -- tile t + 1 lands while tile t is
// computed and stored (the kernel above holds one single-buffered block per CU: its load, compute and store take turns;
// 1.70 ms per 1e8 contexts against the 1.46 ms of this one = 84 B per context at 5.7 TB/s).  The multinomial mode stays
// with the kernel above: its items cost one logarithm each and the single-buffered form already streams (1.49 ms; in place 1.61).
// With normalised rows A = u + 5 eps for every context, so nothing needs a row's f once its items are done:
//   1. items (tickets): cell <- -(u P(x, c)): strictly negative, the other cells keep f >= 0;
//   2. one thread per row: base = -u P(A, n) from the table, cell <- (cell < 0 ? -cell : 0) + base; context terms into the sums;
//   3. in-tile large totals add their own base;  4. the rows leave as a coalesced stream.
// Precondition (as for log(f + eps)): prior rows are non-negative.
struct pln_lds_gi {
  double pri[2][PLN_RMAX * 5 + 2];
  __attribute__((aligned(16))) unsigned char blk[2][PLN_BLOCK_MAX];
  double2 logtab[BEAR_LOGTAB_N];
  double tabD[SRT_NKEY];
  double tabP[SRT_NKEY];
  uint32_t ticket[2];
};
static_assert(sizeof(pln_lds_gi) <= 158 * 1024, "in-place gradient kernel: LDS budget");

__global__ __launch_bounds__(PLN_THREADS, 4) void dm_prior_plan_grad_inplace_kernel(const double *__restrict__ prior,
                                                                                                 bear_params prm_arg, pln_view pv,
                                                                                                 const double2 *__restrict__ logtab_g,
                                                                                                 double *__restrict__ grad_out,
                                                                                                 double *__restrict__ partials,
                                                                                                 const bear_step_io io) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_gi &S = *reinterpret_cast<pln_lds_gi *>(srt_smem);
  const bear_params prm = bear_params_of(prm_arg, io);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  double acc[2] = {0.0, 0.0};
  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < SRT_NKEY) {
    const bear_dp o = srt_general_fast(u + eps5, (double)(tid + 1), logtab_g);
    S.tabD[tid] = o.D;
    S.tabP[tid] = o.P;
  }
  if (tid == 0) {
    S.pri[0][PLN_SENTINEL] = 1.0;
    S.pri[1][PLN_SENTINEL] = 1.0;
    S.ticket[0] = PLN_TICKET_START(PLN_WAVES);
    S.ticket[1] = PLN_TICKET_START(PLN_WAVES);
  }
  auto stage = [&](const pln_tile &ti, uint32_t b) {
    const uint32_t rows = ti.rows_items >> 16;
    if (rows == 0) return;
    const uint32_t pbytes = rows * 40u;
    pln_dma(S.pri[b], prior + ti.row0 * 5, pbytes & ~15u, wave, lane, 0);
    if (pbytes & 15u) {   // odd row count: the trailing 8 bytes through the scalar path (a vector load would drain the DMA queue)
      const __attribute__((address_space(4))) double *tail =
          (const __attribute__((address_space(4))) double *)(uintptr_t)(prior + (ti.row0 + rows) * 5 - 1);
      const double v = *tail;
      if (tid == 0) S.pri[b][rows * 5 - 1] = v;
    }
    pln_dma(S.blk[b], pv.stream + (size_t)ti.off16 * 16, ti.blk16 * 16u, wave, lane, (pbytes + 1023u) >> 10);
  };
  pln_tile cur = pln_load_tile(pv, blockIdx.x);
  stage(cur, 0);
  uint32_t b = 0;
  for (uint64_t t = blockIdx.x; t < pv.n_tiles; t += gridDim.x, b ^= 1u) {
    const pln_tile nxt = pln_load_tile(pv, t + gridDim.x);
    const uint32_t rows = cur.rows_items >> 16, n_light = cur.rows_items & 0xffffu;
    const uint32_t hc = cur.hc_hr >> 16, hr = cur.hc_hr & 0xffffu;
    const pln_layout L = pln_block_layout(rows, n_light, hc, hr);
    srt_wait_dma();   // this tile has landed (requested a whole iteration ago); this wave's stores of the previous tile are out
    srt_sync();       // ... everybody's; nobody reads the other buffer any more (its rows left LDS before the stores were issued)
    stage(nxt, b ^ 1u);
    if (tid == 0) S.ticket[b ^ 1u] = PLN_TICKET_START(PLN_WAVES);
    double *P = S.pri[b];
    const unsigned char *blk = S.blk[b];
    const uint16_t *E = reinterpret_cast<const uint16_t *>(blk);
    const uint8_t *nrow = blk + L.nrow;
    const uint16_t *items = reinterpret_cast<const uint16_t *>(blk + L.items);
    // ---- 1: item units (tickets, dearest first): ELBO, d/dh, and the marked gradient into the item's own cell
    const uint32_t n_hcu = (hc + 63u) >> 6, n_units = (n_light + 63u) >> 6;
    PLN_FOR_UNITS_F(w, &S.ticket[b], n_hcu + n_units, wave, PLN_WAVES) {
      if (w < n_hcu) {
        const uint32_t i = w * 64u + lane;
        if (i < hc) {
          const uint32_t off = reinterpret_cast<const uint16_t *>(blk + L.hoff)[i];
          const double cnt = (double)reinterpret_cast<const uint32_t *>(blk + L.hcnt)[i];
          const double x = __builtin_fma(P[off], u, eps);
          const bear_dp o = srt_general_fast(x, cnt, S.logtab);
          acc[0] += o.D;
          acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
          P[off] = -(u * o.P);
        }
        continue;
      }
      const uint32_t un = n_hcu + n_units - 1u - w;
      uint32_t cmin, cmax;
      const uint32_t ci[1] = {pln_unit_counts(E, n_light, un, lane, &cmin, &cmax)};
      const uint32_t off = items[un * 64u + lane];
      const double x[1] = {__builtin_fma(P[off], u, eps)};
      bear_dp o[1];
      srt_light<1>(x, ci, cmin, cmax, S.logtab, o);
      acc[0] += o[0].D;
      acc[1] = __builtin_fma(eps - x[0], o[0].P, acc[1]);
      if (ci[0] != 0) P[off] = -(u * o[0].P);
    }
    srt_sync();
    // ---- 2: one thread per row: the context term (shared A) and the finished gradient row
    for (uint32_t row = tid; row < rows; row += PLN_THREADS) {
      double base = 0.0;
      uint32_t n = nrow[row];
      if (n == 255u) n = 0u;   // large totals: step 3 / fix-up kernel
      if (n != 0) {
        const double Pn = S.tabP[n - 1];
        acc[0] -= S.tabD[n - 1];
        acc[1] = __builtin_fma(u, Pn, acc[1]);
        base = -u * Pn;
      }
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const double v = P[row * 5 + c];
        P[row * 5 + c] = (v < 0.0 ? -v : 0.0) + base;
      }
    }
    srt_sync();
    // ---- 3: contexts of this tile with a large total add their base
    for (uint32_t i = tid; i < hr; i += PLN_THREADS) {
      const uint32_t row = reinterpret_cast<const uint16_t *>(blk + L.hrow)[i];
      const bear_dp o = srt_general_fast(u + eps5, reinterpret_cast<const double *>(blk + L.hn)[i], S.logtab);
      acc[0] -= o.D;
      acc[1] = __builtin_fma(u, o.P, acc[1]);
#pragma unroll
      for (int c = 0; c < 5; ++c) P[row * 5 + c] -= u * o.P;
    }
    srt_sync();
    // ---- 4: the tile's gradient rows leave as one coalesced stream (from registers: the buffer is free once they are read)
    {
      const uint32_t n_dw = rows * 10u;  // dwords
      const uint4 *src = reinterpret_cast<const uint4 *>(P);
      uint4 *dst = reinterpret_cast<uint4 *>(grad_out + cur.row0 * 5);
      for (uint32_t i = tid; i < (n_dw >> 2); i += PLN_THREADS) dst[i] = src[i];
      if ((n_dw & 3u) && tid == 0) grad_out[(cur.row0 + rows) * 5 - 1] = P[rows * 5 - 1];  // odd row count
    }
    cur = nxt;
  }
  // ---- ELBO / d/dh of the items that overflowed to the global lists (their gradient cells: fix-up kernel)
  srt_wait_dma();
  __syncthreads();
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    const double x = __builtin_fma(prior[h.off], u, eps);
    const bear_dp o = srt_general_fast(x, (double)h.c, S.logtab);
    acc[0] += o.D;
    acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
  }
  for (uint64_t i = gtid; i < pv.n_heavy_row; i += gsz) {
    const bear_dp o = srt_general_fast(u + eps5, pv.heavy_row[i].n, S.logtab);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(u, o.P, acc[1]);
  }
  __syncthreads();
  block_finish<2>(acc, partials, io);
}

// Gradient cells of the items in the plan's global overflow lists (tiles of a dense table: 98 % of their items).  A cell can be hit
// by a column item and by its context, so the two lists go in TWO launches (ROWS = the contexts' base first, then the column
// items), each a plain read-add-write: inside a list a cell occurs once, and the lists are in row order (bear_plan_create sorts
// them), so neighbouring threads touch neighbouring cells.  (Round 4: one launch with fp64 atomics on lists in the order of the
// builder's cursor bumps -- 5.5 ms per 2e7 dense contexts, four times the main kernel.)
template <bool NORM, bool AR, bool ROWS>
__global__ __launch_bounds__(256) void dm_prior_grad_fixup_kernel(const double *__restrict__ prior, bear_params prm_arg, pln_view pv,
                                                                   const double2 *__restrict__ logtab_g,
                                                                   double *__restrict__ grad_out,
                                                                   const bear_step_io io) {
  const bear_params prm = bear_params_of(prm_arg, io);
  __shared__ double2 logtab[BEAR_LOGTAB_N];
  if (threadIdx.x < BEAR_LOGTAB_N) logtab[threadIdx.x] = logtab_g[threadIdx.x];
  __syncthreads();
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  const uint64_t gtid = (uint64_t)blockIdx.x * 256 + threadIdx.x, gsz = (uint64_t)gridDim.x * 256;
  for (uint64_t i = gtid; !ROWS && i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    if (AR) {
      grad_out[h.off] += (double)h.c * bear_rcp(prior[h.off] + eps);
      continue;
    }
    const double x = __builtin_fma(prior[h.off], u, eps);
    const bear_dp o = srt_general_fast(x, (double)h.c, logtab);
    grad_out[h.off] += u * o.P;
  }
  for (uint64_t i = gtid; ROWS && !AR && i < pv.n_heavy_row; i += gsz) {
    const pln_heavy_row h = pv.heavy_row[i];
    const double *f = prior + h.row * 5;
    const double A = NORM ? u + eps5 : __builtin_fma(((f[0] + f[1]) + (f[2] + f[3])) + f[4], u, eps5);
    const bear_dp o = srt_general_fast(A, h.n, logtab);
    for (int b = 0; b < 5; ++b) grad_out[h.row * 5 + b] -= u * o.P;
  }
}

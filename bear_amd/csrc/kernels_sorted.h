// kernels_sorted.h -- the fused DM-marginal + gradient kernels (BEAR mode), "sorted work items".
//
// Why: with integer counts the per-context log marginal (bear_model/core.py:73-74) is a sum of
// log rising factorials whose cost is proportional to the counts.  k-mer count tables are heavy
// tailed, so a row-per-lane kernel runs every wave at the speed of its largest count (measured:
// 10 % of the HBM roofline).  Here each tile of contexts is flattened into independent work items
//     (x, c):  D = lgamma(x+c) - lgamma(x),  P = psi(x+c) - psi(x)
// one per non-zero column (x = alpha_b, c = c_b) and one per non-empty context (x = A, c = n).
// The ELBO and every parameter gradient are linear in the D's and P's of the items with
// coefficients that depend only on x and the item kind, so items never have to be routed back to
// their rows.  A counting sort by c inside LDS makes the product loops wave-uniform.
//
// One block = 512 threads, 2 blocks per CU, one tile = SRT_TILE (512) contexts per iteration:
//   0  stage    the NEXT tile's rows stream into the other half of a double buffer by LDS-DMA
//               (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPRs), issued before the
//               current tile is evaluated and waited for (vmcnt) only at the top of the next
//               iteration -- HBM stays busy while the SIMDs compute.  Barriers inside the loop
//               are raw s_barrier + lgkmcnt waits so they never drain the DMA queue.
//   A  count    one thread per context: n = sum c, S = sum prior; items whose x is the same for
//               every context are served from per-block tables (no sort, no log); the others get
//               key = min(c, 32) - 1 and rank = LDS atomic on an 8-way replicated histogram
//   B  scan     256-entry exclusive scan -> offsets (histogram re-zeroed in place)
//   C  scatter  item record (index into the extended [tile,5 | tile] arrays) -> LDS, sorted by c
//   D  evaluate units of 64 x SRT_ILP items, ascending c: p = prod (x+j), p' by the product rule
//               (un-predicated up to the unit's smallest c), D = log p (table log), P = p'/p
//               (v_rcp_f64 + Newton); c > SRT_CL: shifted Stirling series (bear_math.h)
//   sums        per-thread fp64 accumulators -> block partial -> fixed-order finalize kernel.
#pragma once
#include "bear_common.h"

#define SRT_THREADS 512
#define SRT_WAVES (SRT_THREADS / 64)
#define SRT_TILE 512
#define SRT_NKEY 32  // key = min(c, 32) - 1
#define SRT_REP 8
#define SRT_NHIST (SRT_NKEY * SRT_REP)
#ifndef SRT_CL
#define SRT_CL 24
#endif
// product path for c <= SRT_CL (keys 0 .. SRT_CL-1), Stirling path above
#define SRT_XMAX 0x1p30  // products of <= 31 factors stay finite below this
#define SRT_ILP 2        // light items evaluated per lane per step (independent dependency chains)
#define SRT_UNIT (64 * SRT_ILP)
#define SRT_SUM1_TOL 4.5e-16  // |sum(prior row) - 1| below which A = u + 5 eps is shared (2 ulp)

static_assert(SRT_TILE == SRT_THREADS, "phase A maps one context to one thread");
static_assert((SRT_TILE * 20) % 1024 == 0 && (SRT_TILE * 40) % 1024 == 0, "tiles are whole 1 KiB DMA pieces");

// ---- shared pieces ----------------------------------------------------------------------
__device__ __forceinline__ uint32_t srt_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t srt_key(uint32_t c) { return (c > SRT_NKEY ? SRT_NKEY : c) - 1; }  // c >= 1

// Workgroup barrier that makes prior LDS traffic of every wave visible but leaves vector-memory
// (LDS-DMA) operations in flight -- __syncthreads() would wait for vmcnt(0) as well.
__device__ __forceinline__ void srt_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void srt_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Synchronous, guarded staging (ragged last tile): 16-byte lane loads, dword tail.
__device__ __forceinline__ void srt_stage(uint32_t *lds, const uint32_t *src, uint32_t n_dwords) {
  const uint32_t n_vec = n_dwords >> 2;
  const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
  uint4 *d4 = reinterpret_cast<uint4 *>(lds);
  for (uint32_t i = threadIdx.x; i < n_vec; i += SRT_THREADS) d4[i] = s4[i];
  for (uint32_t i = (n_vec << 2) + threadIdx.x; i < n_dwords; i += SRT_THREADS) lds[i] = src[i];
}

// Asynchronous staging of a whole tile slab (`bytes` a multiple of 1 KiB): wave w moves pieces
// w, w + 8, ...; each piece is one global_load_lds_dwordx4 (lane l: 16 B at +16 l, LDS address
// M0 + 16 l, contiguous).  Issued from inline asm on purpose: through the builtin the compiler
// assumes every later LDS access may alias the in-flight DMA and puts s_waitcnt vmcnt(0) in front
// of it, which serialises the prefetch with the evaluation of the current tile.  Ordering is
// explicit instead: srt_wait_dma() + srt_sync() at the top of the next iteration.
__device__ __forceinline__ void srt_dma(void *lds, const void *src, uint32_t bytes, uint32_t wave, uint32_t lane) {
  const uint32_t d = (uint32_t)(uintptr_t)lds;  // LDS byte address (low 32 bits of the generic pointer)
  const unsigned char *s = static_cast<const unsigned char *>(src) + lane * 16u;
  for (uint32_t piece = wave; piece < (bytes >> 10); piece += SRT_WAVES) {
    const unsigned char *g = s + (piece << 10);
    const uint32_t m = srt_uniform(d + (piece << 10));
    {
      // M0 is compiler-reserved: saved and restored inside the statement that uses it (no "m0" clobber: that is undefined behaviour)
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0)
                   : "v"(g), "s"(m)
                   : "memory");
    }
  }
}

// Exclusive scan of one uint32 per thread over the block.  `scratch` holds SRT_WAVES words.
__device__ __forceinline__ uint32_t srt_block_exscan(uint32_t v, uint32_t *scratch, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) scratch[wave] = incl;
  srt_sync();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SRT_WAVES; ++w) {
    uint32_t t = scratch[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + incl - v;
}

// The general routine, kept out of line so the hot loops stay small (registers, I-cache).
__device__ __noinline__ bear_dp srt_general(double x, double c) {
  if (!(x > 0.0)) return bear_dp{__builtin_nan(""), __builtin_nan("")};
  return bear_dm_item(x, c);
}
// The same on the table log (LDS or global table): ~3x fewer instructions.
__device__ __noinline__ bear_dp srt_general_fast(double x, double c, const double2 *tab);
// the table-log form wherever its argument is in that routine's domain (x > 0 and finite), the library form elsewhere: the heavy
// items of the UNPLANNED sorted kernels (round 6: they had kept the ~500-instruction form; a dense table ran 2.24 ms per 2e7 contexts)
__device__ __forceinline__ bear_dp srt_general_auto(double x, double c, const double2 *tab) {
  return x > 0.0 && x < INFINITY ? srt_general_fast(x, c, tab) : srt_general(x, c);
}
__device__ __noinline__ bear_dp srt_general_fast(double x, double c, const double2 *tab) {
  if (!(x > 0.0) || !(x < 0x1p1000)) return bear_dp{__builtin_nan(""), __builtin_nan("")};
  return bear_dm_item_fast(x, c, tab);
}

// Product-path evaluation of ILP light items per lane: D = log prod_{j<c}(x+j), P = sum 1/(x+j).
// Every lane runs `cmin` un-predicated factors (wave-uniform lower bound of the occupied lanes'
// counts; unoccupied lanes carry c == 0 and a harmless x), then the ragged remainder up to `cmax`
// under predication.  c == 0 yields D = P = 0.
// `in_domain` (wave-uniform): the CALLER guarantees 0 < x <= SRT_XMAX for every occupied lane (the linear step: x = f u + eps with
// f a softmax output it formed itself), so the per-unit domain test -- eight vector instructions of a unit's ~95 -- is skipped.
template <int ILP, bool COEF_V = false>      // (COEF_V: bear_log1p_small)
__device__ __forceinline__ void srt_light(const double (&x)[ILP], const uint32_t (&c)[ILP], uint32_t cmin, uint32_t cmax,
                                          const double2 *logtab, bear_dp (&o)[ILP], bool in_domain = false) {
  double p[ILP], dp[ILP], t[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    p[i] = 1.0;
    dp[i] = 0.0;
    t[i] = x[i];
  }
  // One factor: dp = dp t + p, p = p t, t = t + 1 -- three VOP3 instructions, spelled out: left to itself the compiler forms the
  // first as v_fmac (accumulator = a COPY of p) and shuffles the pairs around it, five instructions per factor instead of three;
  // under predication it turned the three results into six v_cndmask instead of masking the lanes (-ffp-contract=off: the same
  // three roundings either way).
#define SRT_FACTOR(dp, p, t) \
  asm("v_fma_f64 %0, %0, %2, %1\n\tv_mul_f64 %1, %1, %2\n\tv_add_f64 %2, %2, 1.0" : "+v"(dp), "+v"(p), "+v"(t))
  uint32_t j = 0;
  for (; j < cmin; ++j) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) SRT_FACTOR(dp[i], p[i], t[i]);
  }
  for (; j < cmax; ++j) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (j < c[i]) SRT_FACTOR(dp[i], p[i], t[i]);
    }
  }
#undef SRT_FACTOR
  bool odd = false;
#pragma unroll
  for (int i = 0; i < ILP; ++i) {
    const bool live = c[i] != 0;
    o[i].D = live ? bear_log_tab<COEF_V>(p[i], logtab) : 0.0;
    o[i].P = live ? dp[i] * bear_rcp(p[i]) : 0.0;
    if (!in_domain) odd |= live && !(x[i] > 0.0 && x[i] <= SRT_XMAX);
  }
  // Out-of-domain / out-of-range arguments take the general routine (rare, wave-uniform test).
  if (!in_domain && __builtin_amdgcn_ballot_w64(odd)) {
#pragma unroll
    for (int i = 0; i < ILP; ++i)
      if (c[i] != 0 && !(x[i] > 0.0 && x[i] <= SRT_XMAX)) o[i] = srt_general(x[i], (double)c[i]);
  }
}

// Smallest / largest count of a unit of ascending-sorted items [base, min(base + SRT_UNIT, end)):
// the first lane of the first slice and the last occupied lane of the last occupied slice.
__device__ __forceinline__ void srt_unit_range(const uint32_t (&c)[SRT_ILP], uint32_t base, uint32_t end, uint32_t *cmin,
                                               uint32_t *cmax) {
  uint32_t cm = 0;
#pragma unroll
  for (int i = 0; i < SRT_ILP; ++i) {
    const uint32_t lo = base + 64u * i;
    if (end > lo) {
      const uint32_t n = end - lo > 64u ? 64u : end - lo;
      const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)c[i], (int)(n - 1u));
      cm = v > cm ? v : cm;
    }
  }
  *cmax = cm;
  // un-predicated factors are only safe when every lane of every slice is occupied
  *cmin = end - base >= SRT_UNIT ? (uint32_t)__builtin_amdgcn_readlane((int)c[0], 0) : 0u;
}

// =========================================================================================
// mode N: counts + prior rows -> [sum LL, d/dh_signed]          (bear_net.py:146-197, BEAR mode)
//   column item  (row, b): x = prior_b u + eps,    c = c_b :  +D,  (eps - x) P
//   context item (row):    x = S u + 5 eps = A,    c = n   :  -D,  (x - 5 eps) P     (S = sum_b prior_b)
//   (d alpha_b / d h_signed = -(alpha_b - eps); u = 1/h.)  Both kinds share one sorted list: an item
//   is an index into the extended arrays [pri | S] and [cnt | n].  When S = 1 to 2 ulp (any softmax
//   output) A = u + 5 eps is the same for every such context and its item is a table look-up.
// =========================================================================================
struct srt_lds_n {
  double pri[2][SRT_TILE * 5];    // 2 x 20480 B  double-buffered prior rows
  uint32_t cnt[2][SRT_TILE * 5];  // 2 x 10240 B  double-buffered count rows
  double rowS[SRT_TILE];          //  4096 B   S = sum_b prior_b of the current tile
  uint32_t rowN[SRT_TILE];        //  2048 B   n = sum_b c_b (saturating)
  double2 logtab[BEAR_LOGTAB_N];  //  2048 B
  double tabD[SRT_NKEY];          //   256 B   D(u + 5 eps, j + 1)
  double tabP[SRT_NKEY];          //   256 B
  uint32_t hist[SRT_NHIST];       //  1024 B   [key][replica]
  uint16_t offs[SRT_NHIST];       //   512 B
  uint16_t items[SRT_TILE * 6];   //  6144 B
  uint32_t scan[SRT_WAVES];
  uint32_t l_end, h_end;          // light items [0, l_end), heavy [l_end, h_end)
};

template <int STOP>  // developer cut-off: 0 = full kernel, 1 = staging only (stream-rate probe)
__global__ __launch_bounds__(SRT_THREADS, 4) void dm_prior_sorted_kernel(const uint32_t *__restrict__ counts,
                                                                          const double *__restrict__ prior,
                                                                          uint64_t n_rows, bear_params prm,
                                                                          const double2 *__restrict__ logtab_g,
                                                                          double *__restrict__ partials,
                                                                          unsigned long long *__restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  srt_lds_n &S = *reinterpret_cast<srt_lds_n *>(srt_smem);
  // STOP == 9: per-wave cycle totals per phase (s_memtime), diagnostic build only
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
#define SRT_STAMP(k)                                              \
  if (STOP == 9) {                                                \
    const unsigned long long now = __builtin_amdgcn_s_memtime();  \
    tph[k] += now - t_prev;                                       \
    t_prev = now;                                                 \
  }
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6), rep = lane & (SRT_REP - 1);
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  const uint64_t n_tiles = (n_rows + SRT_TILE - 1) / SRT_TILE;
  double acc[2] = {0.0, 0.0};

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < SRT_NKEY) {
    const bear_dp o = srt_general(u + eps5, (double)(tid + 1));
    S.tabD[tid] = o.D;
    S.tabP[tid] = o.P;
  }
  if (tid < SRT_NHIST) S.hist[tid] = 0;

  auto stage = [&](uint64_t tile, uint32_t buf) {
    const uint64_t row0 = tile * SRT_TILE;
    if (n_rows - row0 >= SRT_TILE) {
      srt_dma(S.pri[buf], prior + row0 * 5, SRT_TILE * 40, wave, lane);
      srt_dma(S.cnt[buf], counts + row0 * 5, SRT_TILE * 20, wave, lane);
    } else {
      const uint32_t rows = (uint32_t)(n_rows - row0);
      srt_stage(S.cnt[buf], counts + row0 * 5, rows * 5);
      srt_stage(reinterpret_cast<uint32_t *>(S.pri[buf]), reinterpret_cast<const uint32_t *>(prior + row0 * 5), rows * 10);
    }
  };

  uint64_t tile = blockIdx.x;
  uint32_t buf = 0;
  if (tile < n_tiles) stage(tile, 0);
  for (; tile < n_tiles; tile += gridDim.x, buf ^= 1u) {
    const uint64_t row0 = tile * SRT_TILE;
    const uint32_t rows = (uint32_t)((n_rows - row0 < SRT_TILE) ? (n_rows - row0) : SRT_TILE);
    if (STOP == 9) t_prev = __builtin_amdgcn_s_memtime();
    srt_wait_dma();  // this wave's pieces of the current tile have landed
    srt_sync();      // ... and everybody else's; the previous tile is fully consumed
    SRT_STAMP(0)
    if (tile + gridDim.x < n_tiles) stage(tile + gridDim.x, buf ^ 1u);
    const double *pri = S.pri[buf];
    const uint32_t *cnt = S.cnt[buf];
    if (STOP == 1) {
      acc[0] += (double)cnt[tid] + pri[tid] + (double)cnt[tid + 2048] + pri[tid + 2048];
      continue;
    }
    // ---- A: count (LDS reads first, then the atomics back to back; ranks are not needed until C)
    uint32_t c[5], rank[6];
    double S5;
    {
      const uint32_t rr = tid < rows ? tid : rows - 1;
      double f[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        c[b] = cnt[rr * 5 + b];
        f[b] = pri[rr * 5 + b];
      }
      S5 = ((f[0] + f[1]) + (f[2] + f[3])) + f[4];
    }
    uint32_t nsat = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      if (tid >= rows) c[b] = 0;
      const uint32_t s = nsat + c[b];
      nsat = s < nsat ? 0xffffffffu : s;  // saturating
      rank[b] = 0;
      if (c[b] != 0) rank[b] = atomicAdd(&S.hist[srt_key(c[b]) * SRT_REP + rep], 1u);
    }
    S.rowS[tid] = S5;
    S.rowN[tid] = nsat;
    bool n_item = nsat != 0;
    rank[5] = 0;
    if (n_item && nsat <= SRT_CL && __builtin_fabs(S5 - 1.0) <= SRT_SUM1_TOL) {  // shared A: table look-up
      acc[0] -= S.tabD[nsat - 1];
      acc[1] = __builtin_fma(u, S.tabP[nsat - 1], acc[1]);
      n_item = false;
    }
    if (n_item) rank[5] = atomicAdd(&S.hist[srt_key(nsat) * SRT_REP + rep], 1u);
    srt_sync();
    SRT_STAMP(1)
    // ---- B: scan (entry order = [key][replica]); the histogram is re-zeroed for the next tile
    {
      uint32_t total;
      const uint32_t v = tid < SRT_NHIST ? S.hist[tid] : 0u;
      const uint32_t ex = srt_block_exscan(v, S.scan, &total);
      if (tid < SRT_NHIST) {
        S.offs[tid] = (uint16_t)ex;
        S.hist[tid] = 0;
      }
      if (tid == SRT_CL * SRT_REP) S.l_end = ex;
      if (tid == 0) S.h_end = total;
    }
    srt_sync();
    SRT_STAMP(2)
    // ---- C: scatter
#pragma unroll
    for (int b = 0; b < 5; ++b)
      if (c[b] != 0) S.items[(uint32_t)S.offs[srt_key(c[b]) * SRT_REP + rep] + rank[b]] = (uint16_t)(tid * 5 + b);
    if (n_item) S.items[(uint32_t)S.offs[srt_key(nsat) * SRT_REP + rep] + rank[5]] = (uint16_t)(SRT_TILE * 5 + tid);
    srt_sync();
    SRT_STAMP(3)
    // ---- D: evaluate
    const uint32_t l_end = srt_uniform(S.l_end), h_end = srt_uniform(S.h_end);
    const uint32_t n_units = (l_end + SRT_UNIT - 1) / SRT_UNIT;
    const uint32_t n_work = n_units + ((h_end - l_end + 63) >> 6);
    for (uint32_t un = wave; un < n_work; un += SRT_WAVES) {
      if (STOP == 9 && un >= n_units) { SRT_STAMP(4) }
      if (un < n_units) {  // product path
        const uint32_t base = un * SRT_UNIT;
        uint32_t ci[SRT_ILP];
        double x[SRT_ILP], ek[SRT_ILP];
        bool isn[SRT_ILP];
        bear_dp o[SRT_ILP];
#pragma unroll
        for (int i = 0; i < SRT_ILP; ++i) {
          const uint32_t idx = base + 64u * i + lane;
          const bool on = idx < l_end;
          const uint32_t rec = on ? (uint32_t)S.items[idx] : 0u;
          isn[i] = rec >= SRT_TILE * 5;
          const double xr = isn[i] ? S.rowS[rec - SRT_TILE * 5] : pri[rec];
          const uint32_t cr = isn[i] ? S.rowN[rec - SRT_TILE * 5] : cnt[rec];
          ci[i] = on ? cr : 0u;
          ek[i] = isn[i] ? eps5 : eps;
          x[i] = on ? __builtin_fma(xr, u, ek[i]) : 1.0;
        }
        uint32_t cmin, cmax;
        srt_unit_range(ci, base, l_end, &cmin, &cmax);
        srt_light<SRT_ILP>(x, ci, cmin, cmax, S.logtab, o);
#pragma unroll
        for (int i = 0; i < SRT_ILP; ++i) {
          const double wP = (ek[i] - x[i]) * o[i].P;
          acc[0] += isn[i] ? -o[i].D : o[i].D;
          acc[1] += isn[i] ? -wP : wP;
        }
      } else {  // Stirling path
        const uint32_t idx = l_end + ((un - n_units) << 6) + lane;
        if (idx < h_end) {
          const uint32_t rec = S.items[idx];
          if (rec >= SRT_TILE * 5) {
            const uint32_t row = rec - SRT_TILE * 5;
            const uint32_t *cr = &cnt[row * 5];
            const double A = __builtin_fma(S.rowS[row], u, eps5);
            const double n = (((double)cr[0] + (double)cr[1]) + ((double)cr[2] + (double)cr[3])) + (double)cr[4];
            const bear_dp o = srt_general_auto(A, n, S.logtab);
            acc[0] -= o.D;
            acc[1] = __builtin_fma(A - eps5, o.P, acc[1]);
          } else {
            const double x = __builtin_fma(pri[rec], u, eps);
            const bear_dp o = srt_general_auto(x, (double)cnt[rec], S.logtab);
            acc[0] += o.D;
            acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
          }
        }
        if (STOP == 9) { SRT_STAMP(5) }
      }
    }
    SRT_STAMP(4)
  }
#undef SRT_STAMP
  if (STOP == 9 && dbg && lane == 0)
    for (int k = 0; k < 6; ++k) dbg[((size_t)blockIdx.x * SRT_WAVES + wave) * 6 + k] = tph[k];
  srt_wait_dma();
  __syncthreads();
  block_store_partials<2>(acc, partials);
}

// =========================================================================================
// mode R: train + reference counts -> [sum LL, d/dh_s, d/dtau_s, d/dnu_s]   (bear_ref.py:207-259,
// stop net function).  f_b = (1/4 + E (r_b/R - 1/4)) V for b < 4, f_4 = nw V, alpha = f u + eps.
//   * sum_b f_b = 1, so A = u + 5 eps is the same for every context: the context item and the
//     stop-column item are table look-ups over c <= SRT_CL (tables built once per block).
//   * column items b < 4 are sorted and evaluated as in mode N; their gradient weights are
//     affine in x:  d alpha/d h_s = eps - x,  d alpha/d tau_s = -tau (x - eps - u V / 4),
//     d alpha/d nu_s = nw V (eps - x).  The context item only feeds sum LL and d/dh_s (weights of
//     tau_s and nu_s sum to zero over a row because sum_b f_b is constant).
// =========================================================================================
struct srt_lds_r {
  uint32_t trn[2][SRT_TILE * 5];   // 2 x 10240 B
  uint32_t ref[2][SRT_TILE * 5];   // 2 x 10240 B
  double invR[SRT_TILE];           //  4096 B   1 / sum_b (ref_b + eps)
  double2 logtab[BEAR_LOGTAB_N];   //  2048 B
  double tabD[2][SRT_NKEY];        //   512 B   [0]: context item (x = A), [1]: stop column (x = x4)
  double tabP[2][SRT_NKEY];        //   512 B
  uint32_t hist[SRT_NHIST];        //  1024 B
  uint16_t offs[SRT_NHIST];        //   512 B
  uint16_t items[SRT_TILE * 4];    //  4096 B   flat offset row*5+b
  uint16_t heavy_n[SRT_TILE];      //  1024 B   contexts with n > SRT_CL
  uint16_t heavy_4[SRT_TILE];      //  1024 B   contexts with c_stop > SRT_CL
  uint32_t n_heavy_n[2], n_heavy_4[2];  // per buffer parity: zeroed one tile ahead
  uint32_t scan[SRT_WAVES];
  uint32_t l_end, h_end;
};

__global__ __launch_bounds__(SRT_THREADS, 4) void dm_ref_sorted_kernel(const uint32_t *__restrict__ train,
                                                                        const uint32_t *__restrict__ ref,
                                                                        uint64_t n_rows, bear_params prm,
                                                                        const double2 *__restrict__ logtab_g,
                                                                        double *__restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  srt_lds_r &S = *reinterpret_cast<srt_lds_r *>(srt_smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6), rep = lane & (SRT_REP - 1);
  const double u = prm.inv_h, eps = prm.eps;
  const double A = u + 5.0 * eps;                // sum_b alpha_b
  const double x4 = prm.nw * prm.V * u + eps;    // alpha of the stop column
  const double VU = prm.V * u;
  const double tau = prm.tau;
  const double w2c = tau * (eps + 0.25 * VU);    // d alpha/d tau_s = -tau x + w2c
  const double nwV = prm.nw * prm.V;
  const uint64_t n_tiles = (n_rows + SRT_TILE - 1) / SRT_TILE;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < 2 * SRT_NKEY) {
    const int which = tid / SRT_NKEY, j = tid % SRT_NKEY;
    const bear_dp o = srt_general(which ? x4 : A, (double)(j + 1));
    S.tabD[which][j] = o.D;
    S.tabP[which][j] = o.P;
  }
  if (tid < SRT_NHIST) S.hist[tid] = 0;
  if (tid < 2) {
    S.n_heavy_n[tid] = 0;
    S.n_heavy_4[tid] = 0;
  }

  auto stage = [&](uint64_t tile, uint32_t buf) {
    const uint64_t row0 = tile * SRT_TILE;
    if (n_rows - row0 >= SRT_TILE) {
      srt_dma(S.trn[buf], train + row0 * 5, SRT_TILE * 20, wave, lane);
      srt_dma(S.ref[buf], ref + row0 * 5, SRT_TILE * 20, wave, lane);
    } else {
      const uint32_t rows = (uint32_t)(n_rows - row0);
      srt_stage(S.trn[buf], train + row0 * 5, rows * 5);
      srt_stage(S.ref[buf], ref + row0 * 5, rows * 5);
    }
  };

  uint64_t tile = blockIdx.x;
  uint32_t buf = 0;
  if (tile < n_tiles) stage(tile, 0);
  for (; tile < n_tiles; tile += gridDim.x, buf ^= 1u) {
    const uint64_t row0 = tile * SRT_TILE;
    const uint32_t rows = (uint32_t)((n_rows - row0 < SRT_TILE) ? (n_rows - row0) : SRT_TILE);
    srt_wait_dma();
    srt_sync();
    if (tile + gridDim.x < n_tiles) stage(tile + gridDim.x, buf ^ 1u);
    if (tid == 0) {  // counters of the next tile (their last readers passed the barrier above)
      S.n_heavy_n[buf ^ 1u] = 0;
      S.n_heavy_4[buf ^ 1u] = 0;
    }
    const uint32_t *trn = S.trn[buf];
    const uint32_t *rfc = S.ref[buf];
    // ---- A: per context: normaliser, table items, histogram of column items
    uint32_t c[5], rf[4], rank[4];
    {
      const uint32_t rr = tid < rows ? tid : rows - 1;
#pragma unroll
      for (int b = 0; b < 5; ++b) c[b] = trn[rr * 5 + b];
#pragma unroll
      for (int b = 0; b < 4; ++b) rf[b] = rfc[rr * 5 + b];
    }
    uint32_t nsat = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      if (tid >= rows) c[b] = 0;
      const uint32_t s = nsat + c[b];
      nsat = s < nsat ? 0xffffffffu : s;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      rank[b] = 0;
      if (c[b] != 0) rank[b] = atomicAdd(&S.hist[srt_key(c[b]) * SRT_REP + rep], 1u);
    }
    {
      const double R = (double)(((uint64_t)rf[0] + rf[1]) + ((uint64_t)rf[2] + rf[3])) + 4.0 * eps;  // bear_ref.py:335-337, 30
      S.invR[tid] = bear_rcp(R);
    }
    if (c[4] != 0) {  // stop column: x4 is the same for every context
      if (c[4] <= SRT_CL) {
        const double P = S.tabP[1][c[4] - 1];
        acc[0] += S.tabD[1][c[4] - 1];
        acc[1] = __builtin_fma(eps - x4, P, acc[1]);
        acc[3] = __builtin_fma(VU * nwV, P, acc[3]);  // d alpha_4/d nu_s = u nw V^2
      } else {
        S.heavy_4[atomicAdd(&S.n_heavy_4[buf], 1u)] = (uint16_t)tid;
      }
    }
    if (nsat != 0) {  // context item: x = A for every context
      if (nsat <= SRT_CL) {
        acc[0] -= S.tabD[0][nsat - 1];
        acc[1] = __builtin_fma(u, S.tabP[0][nsat - 1], acc[1]);
      } else {
        S.heavy_n[atomicAdd(&S.n_heavy_n[buf], 1u)] = (uint16_t)tid;
      }
    }
    srt_sync();
    // ---- B: scan
    {
      uint32_t total;
      const uint32_t v = tid < SRT_NHIST ? S.hist[tid] : 0u;
      const uint32_t ex = srt_block_exscan(v, S.scan, &total);
      if (tid < SRT_NHIST) {
        S.offs[tid] = (uint16_t)ex;
        S.hist[tid] = 0;
      }
      if (tid == SRT_CL * SRT_REP) S.l_end = ex;
      if (tid == 0) S.h_end = total;
    }
    srt_sync();
    // ---- C: scatter
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (c[b] != 0) S.items[(uint32_t)S.offs[srt_key(c[b]) * SRT_REP + rep] + rank[b]] = (uint16_t)(tid * 5 + b);
    srt_sync();
    // ---- D: evaluate column items b < 4
    const uint32_t l_end = srt_uniform(S.l_end), h_end = srt_uniform(S.h_end);
    const uint32_t nh_n = srt_uniform(S.n_heavy_n[buf]), nh_4 = srt_uniform(S.n_heavy_4[buf]);
    const uint32_t n_units = (l_end + SRT_UNIT - 1) / SRT_UNIT;
    const uint32_t n_work = n_units + ((h_end - l_end + 63) >> 6);
    // bear_ref.py:30-33 (Jukes-Cantor on the L1-normalised reference row), :63-68 (mix), bear_ref.py:106
    auto alpha_of = [&](uint32_t off) {
      const uint32_t row = (off * 52429u) >> 18;  // off / 5 for off < 2^16
      const double dev = __builtin_fma((double)rfc[off] + eps, S.invR[row], -0.25);
      return __builtin_fma(__builtin_fma(prm.E, dev, 0.25), VU, eps);
    };
    for (uint32_t un = wave; un < n_work; un += SRT_WAVES) {
      if (un < n_units) {
        const uint32_t base = un * SRT_UNIT;
        uint32_t ci[SRT_ILP];
        double x[SRT_ILP];
        bear_dp o[SRT_ILP];
#pragma unroll
        for (int i = 0; i < SRT_ILP; ++i) {
          const uint32_t idx = base + 64u * i + lane;
          const bool on = idx < l_end;
          const uint32_t off = on ? (uint32_t)S.items[idx] : 0u;
          ci[i] = on ? trn[off] : 0u;
          x[i] = on ? alpha_of(off) : 1.0;
        }
        uint32_t cmin, cmax;
        srt_unit_range(ci, base, l_end, &cmin, &cmax);
        srt_light<SRT_ILP>(x, ci, cmin, cmax, S.logtab, o);
#pragma unroll
        for (int i = 0; i < SRT_ILP; ++i) {
          const double w1 = eps - x[i];
          acc[0] += o[i].D;
          acc[1] = __builtin_fma(w1, o[i].P, acc[1]);
          acc[2] = __builtin_fma(__builtin_fma(-tau, x[i], w2c), o[i].P, acc[2]);
          acc[3] = __builtin_fma(nwV * w1, o[i].P, acc[3]);
        }
      } else {
        const uint32_t idx = l_end + ((un - n_units) << 6) + lane;
        if (idx < h_end) {
          const uint32_t off = S.items[idx];
          const double x = alpha_of(off);
          const bear_dp o = srt_general_auto(x, (double)trn[off], S.logtab);
          const double w1 = eps - x;
          acc[0] += o.D;
          acc[1] = __builtin_fma(w1, o.P, acc[1]);
          acc[2] = __builtin_fma(__builtin_fma(-tau, x, w2c), o.P, acc[2]);
          acc[3] = __builtin_fma(nwV * w1, o.P, acc[3]);
        }
      }
    }
    // ---- rare: contexts whose total / stop count needs the Stirling path
    for (uint32_t i = tid; i < nh_n; i += SRT_THREADS) {
      const uint32_t *cr = &trn[(uint32_t)S.heavy_n[i] * 5];
      const double n = (((double)cr[0] + (double)cr[1]) + ((double)cr[2] + (double)cr[3])) + (double)cr[4];
      const bear_dp o = srt_general_auto(A, n, S.logtab);
      acc[0] -= o.D;
      acc[1] = __builtin_fma(u, o.P, acc[1]);
    }
    for (uint32_t i = tid; i < nh_4; i += SRT_THREADS) {
      const bear_dp o = srt_general_auto(x4, (double)trn[(uint32_t)S.heavy_4[i] * 5 + 4], S.logtab);
      acc[0] += o.D;
      acc[1] = __builtin_fma(eps - x4, o.P, acc[1]);
      acc[3] = __builtin_fma(VU * nwV, o.P, acc[3]);
    }
  }
  srt_wait_dma();
  __syncthreads();
  block_store_partials<4>(acc, partials);
}

// =========================================================================================
// Item-level entry (tests / diagnostics): D and P of independent (x, c) items through exactly the
// code paths the fused kernels use.  path 0: as the kernels choose (product path for
// c <= SRT_CL), 1: general routine for every item.
// =========================================================================================
__global__ __launch_bounds__(256) void dm_items_kernel(const double *__restrict__ x, const uint32_t *__restrict__ c,
                                                       uint64_t n, int path, const double2 *__restrict__ logtab_g,
                                                       double *__restrict__ D, double *__restrict__ P) {
  __shared__ double2 logtab[BEAR_LOGTAB_N];
  if (threadIdx.x < BEAR_LOGTAB_N) logtab[threadIdx.x] = logtab_g[threadIdx.x];
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool on = i < n;
  const uint32_t ci = on ? c[i] : 0u;
  const bool light = path == 0 && ci <= SRT_CL;  // path 1 / 2: library-log / table-log general routine for every item
  const double xi[1] = {on && light ? x[i] : 1.0};
  const uint32_t cl[1] = {light ? ci : 0u};
  // wave-uniform bounds of an unsorted wave: cmin = 0 (everything predicated), cmax = wave maximum
  uint32_t cm = cl[0];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o2 = (uint32_t)__shfl_xor((int)cm, off, 64);
    cm = o2 > cm ? o2 : cm;
  }
  bear_dp o[1];
  srt_light<1>(xi, cl, 0u, srt_uniform(cm), logtab, o);
  if (!light) {
    o[0].D = 0.0;
    o[0].P = 0.0;
    if (ci != 0) o[0] = path == 2 ? srt_general_fast(x[i], (double)ci, logtab) : srt_general(x[i], (double)ci);
  }
  if (on) {
    D[i] = o[0].D;
    P[i] = o[0].P;
  }
}

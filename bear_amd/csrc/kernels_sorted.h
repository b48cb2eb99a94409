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
// One tile (SRT_TILE contexts) per block iteration, 512 threads, 2 blocks per CU:
//   0  stage    flat, fully coalesced 16-byte lane loads of the [tile,5] rows into LDS
//   A  count    one thread per context: n = sum c, histogram key = min(c, 32) - 1 per item,
//               rank = LDS atomic on an 8-way replicated histogram (replica = lane & 7)
//   B  scan     512-entry exclusive scan (one entry per thread)
//   C  scatter  item record (row << 3 | slot) -> LDS at offset[key][replica] + rank
//   D  evaluate 64-item chunks, kinds never mixed in a chunk, ascending c inside a kind:
//               c <= 31: p = prod (x+j), p' by the product rule, D = log p (table log),
//               P = p'/p (v_rcp_f64 + Newton); c >= 32: shifted Stirling series (bear_math.h)
//   sums        per-thread fp64 accumulators -> block partial -> fixed-order finalize kernel.
#pragma once
#include "bear_common.h"

#define SRT_THREADS 512
#define SRT_WAVES (SRT_THREADS / 64)
#define SRT_TILE 1024
#define SRT_RPT (SRT_TILE / SRT_THREADS)
#define SRT_NKEY 32  // keys 0..30: c = key + 1 (product path); key 31: c >= 32 (Stirling path)
#define SRT_REP 8
#define SRT_XMAX 0x1p30  // products of <= 31 factors stay finite below this

// ---- shared pieces ----------------------------------------------------------------------
__device__ __forceinline__ void srt_stage(uint32_t *lds, const uint32_t *src, uint32_t n_dwords) {
  const uint32_t n_vec = n_dwords >> 2;
  const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
  uint4 *d4 = reinterpret_cast<uint4 *>(lds);
  for (uint32_t i = threadIdx.x; i < n_vec; i += SRT_THREADS) d4[i] = s4[i];
  for (uint32_t i = (n_vec << 2) + threadIdx.x; i < n_dwords; i += SRT_THREADS) lds[i] = src[i];
}

__device__ __forceinline__ uint32_t srt_wave_max(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// Exclusive scan of one uint32 per thread over the block (SRT_THREADS threads).  `scratch` holds
// SRT_WAVES + 1 words.  Returns the exclusive prefix; *total receives the block total.
__device__ __forceinline__ uint32_t srt_block_exscan(uint32_t v, uint32_t *scratch, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
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

// Product-path evaluation of one light item per lane: D = log prod_{j<c}(x+j), P = sum 1/(x+j).
// `cmax` is wave-uniform (>= every lane's c); lanes with c == 0 return D = P = 0.
__device__ __forceinline__ bear_dp srt_light(double x, uint32_t c, uint32_t cmax, const double2 *logtab) {
  double p = 1.0, dp = 0.0, t = x;
  for (uint32_t j = 0; j < cmax; ++j) {
    if (j < c) {
      dp = __builtin_fma(dp, t, p);
      p *= t;
      t += 1.0;
    }
  }
  bear_dp o;
  o.D = bear_log_tab(p, logtab);
  o.P = dp * bear_rcp(p);
  if (c == 0) o.D = 0.0;
  // Out-of-domain / out-of-range arguments take the general routine (rare, wave-uniform test).
  const bool odd = c != 0 && !(x > 0.0 && x <= SRT_XMAX);
  if (__builtin_amdgcn_ballot_w64(odd)) {
    if (odd) o = (x > 0.0) ? bear_dm_item(x, (double)c) : bear_dp{__builtin_nan(""), __builtin_nan("")};
  }
  return o;
}

struct srt_segs {
  uint32_t l0_end;  // kind-0 light items  [0, l0_end)
  uint32_t h0_end;  // kind-0 heavy items  [l0_end, h0_end)
  uint32_t l1_end;  // kind-1 light items  [h0_end, l1_end)
  uint32_t h1_end;  // kind-1 heavy items  [l1_end, h1_end)
};

// =========================================================================================
// mode N: counts + prior rows -> [sum LL, d/dh_signed]          (bear_net.py:146-197, BEAR mode)
//   kind 0 = column items (x = prior_b / h + eps, c = c_b), kind 1 = context items (x = A, c = n)
//   sum LL      = sum_kind0 D - sum_kind1 D
//   d/dh_signed = sum_kind0 (eps - x) P + sum_kind1 (x - 5 eps) P        (d alpha_b/d h_s = -(x - eps))
// =========================================================================================
struct srt_lds_n {
  double pri[SRT_TILE * 5];               // 40960 B
  uint32_t cnt[SRT_TILE * 5];             // 20480 B
  double2 logtab[BEAR_LOGTAB_N];          //  2048 B
  uint32_t hist[2 * SRT_NKEY * SRT_REP];  //  2048 B   [kind][key][replica]
  uint16_t items[SRT_TILE * 6];           // 12288 B
  uint8_t nkey[SRT_TILE];                 //  1024 B   min(n, 32) - 1 per context (255: empty)
  uint32_t scan[SRT_WAVES + 1];
  srt_segs segs;
};

__global__ __launch_bounds__(SRT_THREADS, 4) void dm_prior_sorted_kernel(const uint32_t *__restrict__ counts,
                                                                          const double *__restrict__ prior,
                                                                          uint64_t n_rows, bear_params prm,
                                                                          const double2 *__restrict__ logtab_g,
                                                                          double *__restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  srt_lds_n &S = *reinterpret_cast<srt_lds_n *>(srt_smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rep = lane & (SRT_REP - 1);
  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  const uint64_t n_tiles = (n_rows + SRT_TILE - 1) / SRT_TILE;
  double acc[2] = {0.0, 0.0};

  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * SRT_TILE;
    const uint32_t rows = (uint32_t)((n_rows - row0 < SRT_TILE) ? (n_rows - row0) : SRT_TILE);
    __syncthreads();  // previous tile fully consumed
    // ---- 0: stage
    srt_stage(S.cnt, counts + row0 * 5, rows * 5);
    srt_stage(reinterpret_cast<uint32_t *>(S.pri), reinterpret_cast<const uint32_t *>(prior + row0 * 5), rows * 10);
    S.hist[tid] = 0;  // 2 * 32 * 8 == SRT_THREADS
    __syncthreads();
    // ---- A: count
    uint32_t c[SRT_RPT][5], rank[SRT_RPT][6], nk[SRT_RPT];
#pragma unroll
    for (int k = 0; k < SRT_RPT; ++k) {
      const uint32_t r = tid + k * SRT_THREADS;
      const bool valid = r < rows;
      uint32_t nsat = 0;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        c[k][b] = valid ? S.cnt[r * 5 + b] : 0u;
        const uint32_t s = nsat + c[k][b];
        nsat = s < nsat ? 0xffffffffu : s;  // saturating: only "is it > 31" matters here
        rank[k][b] = 0;
        if (c[k][b] != 0) {
          const uint32_t key = (c[k][b] > SRT_NKEY ? SRT_NKEY : c[k][b]) - 1;
          rank[k][b] = atomicAdd(&S.hist[(key * SRT_REP) + rep], 1u);
        }
      }
      nk[k] = 255u;
      rank[k][5] = 0;
      if (nsat != 0) {
        nk[k] = (nsat > SRT_NKEY ? SRT_NKEY : nsat) - 1;
        rank[k][5] = atomicAdd(&S.hist[((SRT_NKEY + nk[k]) * SRT_REP) + rep], 1u);
      }
      if (valid) S.nkey[r] = (uint8_t)nk[k];
    }
    __syncthreads();
    // ---- B: scan (entry order = [kind][key][replica])
    {
      uint32_t total;
      const uint32_t v = S.hist[tid];
      const uint32_t ex = srt_block_exscan(v, S.scan, &total);
      S.hist[tid] = ex;
      if (tid == (SRT_NKEY - 1) * SRT_REP) S.segs.l0_end = ex;
      if (tid == SRT_NKEY * SRT_REP) S.segs.h0_end = ex;
      if (tid == (2 * SRT_NKEY - 1) * SRT_REP) S.segs.l1_end = ex;
      if (tid == 0) S.segs.h1_end = total;
    }
    __syncthreads();
    // ---- C: scatter
#pragma unroll
    for (int k = 0; k < SRT_RPT; ++k) {
      const uint32_t r = tid + k * SRT_THREADS;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        if (c[k][b] != 0) {
          const uint32_t key = (c[k][b] > SRT_NKEY ? SRT_NKEY : c[k][b]) - 1;
          S.items[S.hist[key * SRT_REP + rep] + rank[k][b]] = (uint16_t)((r << 3) | b);
        }
      }
      if (nk[k] != 255u) S.items[S.hist[(SRT_NKEY + nk[k]) * SRT_REP + rep] + rank[k][5]] = (uint16_t)((r << 3) | 5u);
    }
    __syncthreads();
    // ---- D: evaluate
    const srt_segs sg = S.segs;
    const uint32_t ch_l0 = (sg.l0_end + 63) >> 6;
    const uint32_t ch_h0 = ch_l0 + ((sg.h0_end - sg.l0_end + 63) >> 6);
    const uint32_t ch_l1 = ch_h0 + ((sg.l1_end - sg.h0_end + 63) >> 6);
    const uint32_t ch_h1 = ch_l1 + ((sg.h1_end - sg.l1_end + 63) >> 6);
    for (uint32_t ch = wave; ch < ch_h1; ch += SRT_WAVES) {
      if (ch < ch_l0) {  // kind 0, product path
        const uint32_t idx = (ch << 6) + lane;
        const bool on = idx < sg.l0_end;
        const uint32_t rec = on ? S.items[idx] : 0u;
        const uint32_t off = (rec >> 3) * 5 + (rec & 7u);
        const uint32_t ci = on ? S.cnt[off] : 0u;
        const double x = __builtin_fma(S.pri[off], u, eps);
        const bear_dp o = srt_light(x, ci, srt_wave_max(ci), S.logtab);
        acc[0] += o.D;
        acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
      } else if (ch < ch_h0) {  // kind 0, Stirling path
        const uint32_t idx = sg.l0_end + ((ch - ch_l0) << 6) + lane;
        if (idx < sg.h0_end) {
          const uint32_t rec = S.items[idx];
          const uint32_t off = (rec >> 3) * 5 + (rec & 7u);
          const double x = __builtin_fma(S.pri[off], u, eps);
          const bear_dp o = (x > 0.0) ? bear_dm_item(x, (double)S.cnt[off]) : bear_dp{__builtin_nan(""), __builtin_nan("")};
          acc[0] += o.D;
          acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
        }
      } else {  // kind 1: context items, x = A = sum_b alpha_b
        const bool light = ch < ch_l1;
        const uint32_t idx = light ? sg.h0_end + ((ch - ch_h0) << 6) + lane : sg.l1_end + ((ch - ch_l1) << 6) + lane;
        const bool on = idx < (light ? sg.l1_end : sg.h1_end);
        const uint32_t row = on ? (uint32_t)(S.items[idx] >> 3) : 0u;
        const double *f = &S.pri[row * 5];
        const double A = __builtin_fma(((f[0] + f[1]) + (f[2] + f[3])) + f[4], u, eps5);
        bear_dp o;
        if (light) {
          const uint32_t ci = on ? (uint32_t)S.nkey[row] + 1u : 0u;
          o = srt_light(A, ci, srt_wave_max(ci), S.logtab);
        } else {
          o.D = 0.0;
          o.P = 0.0;
          if (on) {
            const uint32_t *cr = &S.cnt[row * 5];
            const double n = (((double)cr[0] + (double)cr[1]) + ((double)cr[2] + (double)cr[3])) + (double)cr[4];
            o = (A > 0.0) ? bear_dm_item(A, n) : bear_dp{__builtin_nan(""), __builtin_nan("")};
          }
        }
        acc[0] -= o.D;
        acc[1] = __builtin_fma(A - eps5, o.P, acc[1]);
      }
    }
  }
  block_store_partials<2>(acc, partials);
}

// =========================================================================================
// mode R: train + reference counts -> [sum LL, d/dh_s, d/dtau_s, d/dnu_s]   (bear_ref.py:207-259,
// stop net function).  f_b = (1/4 + E (r_b/R - 1/4)) V for b < 4, f_4 = nw V, alpha = f u + eps.
//   * sum_b f_b = 1, so A = u + 5 eps is the same for every context: the context item and the
//     stop-column item are table look-ups over c <= 31 (tables built once per block).
//   * column items b < 4 are sorted and evaluated as in mode N; their gradient weights are
//     affine in x:  d alpha/d h_s = eps - x,  d alpha/d tau_s = -tau (x - eps - u V / 4),
//     d alpha/d nu_s = nw V (eps - x).  The context item only feeds sum LL and d/dh_s (weights of
//     tau_s and nu_s sum to zero over a row because sum_b f_b is constant).
// =========================================================================================
struct srt_lds_r {
  uint32_t trn[SRT_TILE * 5];          // 20480 B
  uint32_t ref[SRT_TILE * 5];          // 20480 B
  double invR[SRT_TILE];               //  8192 B   1 / sum_b (ref_b + eps)
  double2 logtab[BEAR_LOGTAB_N];       //  2048 B
  double tabD[2][SRT_NKEY];            //   512 B   [0]: context item (x = A), [1]: stop column (x = x4)
  double tabP[2][SRT_NKEY];            //   512 B
  uint32_t hist[SRT_NKEY * SRT_REP];   //  1024 B
  uint16_t items[SRT_TILE * 4];        //  8192 B
  uint16_t heavy_n[SRT_TILE];          //  2048 B   contexts with n >= 32
  uint16_t heavy_4[SRT_TILE];          //  2048 B   contexts with c_stop >= 32
  uint32_t n_heavy_n, n_heavy_4;
  uint32_t scan[SRT_WAVES + 1];
  srt_segs segs;
};

__global__ __launch_bounds__(SRT_THREADS, 4) void dm_ref_sorted_kernel(const uint32_t *__restrict__ train,
                                                                        const uint32_t *__restrict__ ref,
                                                                        uint64_t n_rows, bear_params prm,
                                                                        const double2 *__restrict__ logtab_g,
                                                                        double *__restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  srt_lds_r &S = *reinterpret_cast<srt_lds_r *>(srt_smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rep = lane & (SRT_REP - 1);
  const double u = prm.inv_h, eps = prm.eps;
  const double A = u + 5.0 * eps;                // sum_b alpha_b
  const double x4 = prm.nw * prm.V * u + eps;    // alpha of the stop column
  const double VU = prm.V * u;
  const double tau = prm.tau;
  const double w2c = tau * (eps + 0.25 * VU);    // d alpha/d tau_s = -tau x + w2c
  const double nwV = prm.nw * prm.V;
  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < 2 * SRT_NKEY) {
    const int which = tid / SRT_NKEY, j = tid % SRT_NKEY;
    const bear_dp o = bear_dm_item(which ? x4 : A, (double)(j + 1));
    S.tabD[which][j] = o.D;
    S.tabP[which][j] = o.P;
  }
  const uint64_t n_tiles = (n_rows + SRT_TILE - 1) / SRT_TILE;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};

  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * SRT_TILE;
    const uint32_t rows = (uint32_t)((n_rows - row0 < SRT_TILE) ? (n_rows - row0) : SRT_TILE);
    __syncthreads();
    srt_stage(S.trn, train + row0 * 5, rows * 5);
    srt_stage(S.ref, ref + row0 * 5, rows * 5);
    if (tid < SRT_NKEY * SRT_REP) S.hist[tid] = 0;
    if (tid == 0) {
      S.n_heavy_n = 0;
      S.n_heavy_4 = 0;
    }
    __syncthreads();
    // ---- A: per context: normaliser, table items, histogram of column items
    uint32_t c[SRT_RPT][4], rank[SRT_RPT][4];
#pragma unroll
    for (int k = 0; k < SRT_RPT; ++k) {
      const uint32_t r = tid + k * SRT_THREADS;
      const bool valid = r < rows;
      uint32_t nsat = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        c[k][b] = valid ? S.trn[r * 5 + b] : 0u;
        const uint32_t s = nsat + c[k][b];
        nsat = s < nsat ? 0xffffffffu : s;
        rank[k][b] = 0;
        if (c[k][b] != 0) {
          const uint32_t key = (c[k][b] > SRT_NKEY ? SRT_NKEY : c[k][b]) - 1;
          rank[k][b] = atomicAdd(&S.hist[key * SRT_REP + rep], 1u);
        }
      }
      const uint32_t c4 = valid ? S.trn[r * 5 + 4] : 0u;
      {
        const uint32_t s = nsat + c4;
        nsat = s < nsat ? 0xffffffffu : s;
      }
      if (valid) {
        const uint32_t *rr = &S.ref[r * 5];
        const double R = (double)((uint64_t)rr[0] + rr[1] + rr[2] + rr[3]) + 4.0 * eps;  // bear_ref.py:335-337, 30
        S.invR[r] = bear_rcp(R);
      }
      // stop column: x4 is the same for every context
      if (c4 != 0) {
        if (c4 < SRT_NKEY) {
          const double P = S.tabP[1][c4 - 1];
          acc[0] += S.tabD[1][c4 - 1];
          acc[1] = __builtin_fma(eps - x4, P, acc[1]);
          acc[3] = __builtin_fma(VU * nwV, P, acc[3]);  // d alpha_4/d nu_s = u nw V^2
        } else {
          S.heavy_4[atomicAdd(&S.n_heavy_4, 1u)] = (uint16_t)r;
        }
      }
      // context item: x = A for every context
      if (nsat != 0) {
        if (nsat < SRT_NKEY) {
          acc[0] -= S.tabD[0][nsat - 1];
          acc[1] = __builtin_fma(u, S.tabP[0][nsat - 1], acc[1]);
        } else {
          S.heavy_n[atomicAdd(&S.n_heavy_n, 1u)] = (uint16_t)r;
        }
      }
    }
    __syncthreads();
    // ---- B: scan
    {
      uint32_t total;
      const uint32_t v = tid < SRT_NKEY * SRT_REP ? S.hist[tid] : 0u;
      const uint32_t ex = srt_block_exscan(v, S.scan, &total);
      if (tid < SRT_NKEY * SRT_REP) S.hist[tid] = ex;
      if (tid == (SRT_NKEY - 1) * SRT_REP) S.segs.l0_end = ex;
      if (tid == 0) S.segs.h0_end = total;
    }
    __syncthreads();
    // ---- C: scatter
#pragma unroll
    for (int k = 0; k < SRT_RPT; ++k) {
      const uint32_t r = tid + k * SRT_THREADS;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (c[k][b] != 0) {
          const uint32_t key = (c[k][b] > SRT_NKEY ? SRT_NKEY : c[k][b]) - 1;
          S.items[S.hist[key * SRT_REP + rep] + rank[k][b]] = (uint16_t)((r << 3) | b);
        }
      }
    }
    __syncthreads();
    // ---- D: evaluate column items b < 4
    const uint32_t l0_end = S.segs.l0_end, h0_end = S.segs.h0_end;
    const uint32_t nh_n = S.n_heavy_n, nh_4 = S.n_heavy_4;
    const uint32_t ch_l0 = (l0_end + 63) >> 6;
    const uint32_t ch_h0 = ch_l0 + ((h0_end - l0_end + 63) >> 6);
    for (uint32_t ch = wave; ch < ch_h0; ch += SRT_WAVES) {
      const bool light = ch < ch_l0;
      const uint32_t idx = light ? (ch << 6) + lane : l0_end + ((ch - ch_l0) << 6) + lane;
      const bool on = idx < (light ? l0_end : h0_end);
      const uint32_t rec = on ? S.items[idx] : 0u;
      const uint32_t row = rec >> 3, off = row * 5 + (rec & 7u);
      const uint32_t ci = on ? S.trn[off] : 0u;
      // bear_ref.py:30-33 (Jukes-Cantor on the L1-normalised reference row), :63-68 (mix), bear_ref.py:106
      const double dev = __builtin_fma((double)S.ref[off] + eps, S.invR[row], -0.25);
      const double x = __builtin_fma(__builtin_fma(prm.E, dev, 0.25), VU, eps);
      bear_dp o;
      if (light) {
        o = srt_light(x, ci, srt_wave_max(ci), S.logtab);
      } else {
        o.D = 0.0;
        o.P = 0.0;
        if (on) o = bear_dm_item(x, (double)ci);
      }
      const double w1 = eps - x;
      acc[0] += o.D;
      acc[1] = __builtin_fma(w1, o.P, acc[1]);
      acc[2] = __builtin_fma(__builtin_fma(-tau, x, w2c), o.P, acc[2]);
      acc[3] = __builtin_fma(nwV * w1, o.P, acc[3]);
    }
    // ---- rare: contexts whose total / stop count needs the Stirling path
    for (uint32_t i = tid; i < nh_n; i += SRT_THREADS) {
      const uint32_t *cr = &S.trn[(uint32_t)S.heavy_n[i] * 5];
      const double n = (((double)cr[0] + (double)cr[1]) + ((double)cr[2] + (double)cr[3])) + (double)cr[4];
      const bear_dp o = bear_dm_item(A, n);
      acc[0] -= o.D;
      acc[1] = __builtin_fma(u, o.P, acc[1]);
    }
    for (uint32_t i = tid; i < nh_4; i += SRT_THREADS) {
      const bear_dp o = bear_dm_item(x4, (double)S.trn[(uint32_t)S.heavy_4[i] * 5 + 4]);
      acc[0] += o.D;
      acc[1] = __builtin_fma(eps - x4, o.P, acc[1]);
      acc[3] = __builtin_fma(VU * nwV, o.P, acc[3]);
    }
  }
  block_store_partials<4>(acc, partials);
}

// =========================================================================================
// Item-level entry (tests / diagnostics): D and P of independent (x, c) items through exactly the
// code paths the fused kernels use.  path 0: as the kernels choose (product path for c <= 31),
// 1: general routine bear_dm_item for every item.
// =========================================================================================
__global__ __launch_bounds__(256) void dm_items_kernel(const double *__restrict__ x, const uint32_t *__restrict__ c,
                                                       uint64_t n, int path, const double2 *__restrict__ logtab_g,
                                                       double *__restrict__ D, double *__restrict__ P) {
  __shared__ double2 logtab[BEAR_LOGTAB_N];
  if (threadIdx.x < BEAR_LOGTAB_N) logtab[threadIdx.x] = logtab_g[threadIdx.x];
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * 256;
  const uint64_t i = base + threadIdx.x;
  const bool on = i < n;
  const double xi = on ? x[i] : 1.0;
  const uint32_t ci = on ? c[i] : 0u;
  bear_dp o;
  o.D = 0.0;
  o.P = 0.0;
  const bool light = path == 0 && ci < SRT_NKEY;
  const uint32_t cl = light ? ci : 0u;
  const bear_dp ol = srt_light(xi, cl, srt_wave_max(cl), logtab);
  if (light) {
    o = ol;
  } else if (ci != 0) {
    o = (xi > 0.0) ? bear_dm_item(xi, (double)ci) : bear_dp{__builtin_nan(""), __builtin_nan("")};
  }
  if (on) {
    D[i] = o.D;
    P[i] = o.P;
  }
}

// kernels_eval.h -- held-out evaluation (bear_net.py:323-371, bear_ref.py:391-446) and the BMM marginal
// (dataloader.py:111-147) as one pass over the rows.
//
// Per context i with test counts t (n = sum t), optional training counts r and AR-function row f:
//   BEAR, one per h_j      conc = f / h_j + r + eps     LL = lbeta(conc + t) - lbeta(conc)   (core.py:73-74)
//   AR                     p    = f + eps               LL = sum_b t_b log p_b                (core.py:138-139)
//   vanilla, one per v_k   conc = r + v_k + eps         LL as BEAR                            (bear_net.py:328-343)
// and for every model the count of the test transitions that fall on the model's most likely next letter,
// argmax_b(conc_b + sigma z_b) with z ~ N(0,1), sigma = 100 eps (DM models, core.py:69-71) or eps (AR model,
// core.py:134-136).  The reference draws z from tf.random; here z is a counter-based hash of
// (seed, model, global row, letter) so that the oracle restates it bit for bit.  |z| < 8.6 for a 53-bit
// uniform, so the hash is only evaluated when the two largest entries are within 17.5 sigma of each other.
//
// Output vector [dev]: ll_ear[n_h], ll_arm, ll_van[n_van], cor_ear[n_h], cor_arm, cor_van[n_van], total_len
// (the 7 partial sums of bear_net.py:370-371, flattened).
#pragma once
#include "bear_common.h"
#include "kernels_rows.h"
#include "kernels_synth.h"

#define EVL_THREADS 256
#define EVL_WAVES (EVL_THREADS / 64)
#define EVL_MAX_MODELS 64                     // n_h + n_van
#define EVL_MAX_OUT (2 * EVL_MAX_MODELS + 3)
#define EVL_ID_ARM 1000u
#define EVL_ID_VAN 2000u

struct evl_args {
  int n_h, n_van, arm, has_train, has_prior;
  int has_rid;                                // planned kernel only: row i is row row_base + row_ids[i] of the table (compacted batch)
  double eps;
  uint64_t seed, row_base;                    // noise stream; global index of row 0 (rank shards / batches)
  double inv_h[EVL_MAX_MODELS];               // [0, n_h): 1 / h_j; [n_h, n_h + n_van): v_k
};

// Tie-breaking noise: one Box-Muller draw per (seed, model id, global row, letter) from ONE 64-bit hash word --
//   k = mix64(mix64(seed + model) ^ (row * 5 + letter)),  u1 = (k >> 32 + 1/2) 2^-32,  u2 = (k & 0xffffffff + 1/2) 2^-32,
//   z = sqrt(-2 ln u1) cos(2 pi u2)           (|z| < 6.8 for 32-bit uniforms; oracle/bear_oracle.py:eval_noise restates it)
// evaluated on the table log, hardware sqrt and cospi (exact range reduction): the oracle's number to a few ulp; an arg-max
// decision could only differ when two noisy values agree to ~1e-15 of sigma.
__device__ __forceinline__ uint64_t evl_key(uint64_t model_base, uint64_t cell) { return mix64(model_base ^ cell); }
__device__ __forceinline__ double evl_gauss(uint64_t k, const double2 *logtab) {
  const double u1 = ((double)(uint32_t)(k >> 32) + 0.5) * 0x1p-32, u2 = ((double)(uint32_t)k + 0.5) * 0x1p-32;
  return sqrt(-2.0 * bear_log_tab(u1, logtab)) * cospi(2.0 * u2);
}
// The same value in fp32 on the hardware transcendentals (v_log_f32, v_sqrt_f32, v_cos_f32: ~6 instructions instead of ~70).
// |error| <= 1e-5 whenever -2 ln u1 >= 1e-4; `*tiny` reports the other case (probability 5e-5), where only the exact form counts.
__device__ __forceinline__ float evl_gauss_f32(uint64_t k, bool *tiny) {
  const float u1 = ((float)(uint32_t)(k >> 32) + 0.5f) * 0x1p-32f, u2 = ((float)(uint32_t)k + 0.5f) * 0x1p-32f;
  const float r2 = -1.3862943611198906f * __builtin_amdgcn_logf(u1);     // -2 ln 2 log2(u1)
  *tiny = !(r2 >= 1e-4f);
  return __builtin_amdgcn_sqrtf(r2) * __builtin_amdgcn_cosf(u2);        // v_cos_f32 takes revolutions
}

// log of a probability that may be zero or (malformed prior rows) negative, on the table log
__device__ __forceinline__ double evl_log_any(double p, const double2 *logtab) {
  return p > 0.0 ? bear_log_tab(p, logtab) : (p == 0.0 ? -INFINITY : __builtin_nan(""));
}

// largest entry and whether it wins whatever the noise (|z| < 8.6: a gap above 17.5 sigma cannot be bridged)
__device__ __forceinline__ bool evl_argmax_clear(const double (&a)[5], double sigma, int &i1) {
  i1 = 0;
  double v1 = a[0], v2 = -INFINITY;
#pragma unroll
  for (int b = 1; b < 5; ++b) {
    if (a[b] > v1) {
      v2 = v1;
      v1 = a[b];
      i1 = b;
    } else if (a[b] > v2) {
      v2 = a[b];
    }
  }
  return v1 - v2 > 17.5 * sigma;
}

// arg-max of a[b] + sigma z_b over the letters that can still win (within 17.5 sigma of the largest entry); the noise of
// the others is never formed.  First index on exact ties, as argmax does.  A first pass in fp32 decides whenever the two
// best noisy values are more than 2e-4 sigma apart (20x its error bound); the rest (~0.05 % of the calls) repeat in fp64.
// The fp64 form is out of line with its arguments BY VALUE (registers): inlined, the constants of its cospi / sqrt / log
// expansions get hoisted into the caller's loops (60 registers, spilled); by reference the concentrations would travel through
// scratch memory.
__device__ __noinline__ int evl_argmax_exact(double a0, double a1, double a2, double a3, double a4, double top, double sigma,
                                             uint64_t base, uint64_t row, const double2 *logtab) {
  const double a[5] = {a0, a1, a2, a3, a4};
  int i1 = -1;
  double v1 = -INFINITY;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    if (top - a[b] <= 17.5 * sigma) {
      const double v = __builtin_fma(sigma, evl_gauss(evl_key(base, row * 5 + b), logtab), a[b]);
      if (v > v1) {
        v1 = v;
        i1 = b;
      }
    }
  }
  return i1 < 0 ? 0 : i1;
}
__device__ __forceinline__ int evl_argmax_noisy(const double (&a)[5], double sigma, uint64_t seed, uint32_t model, uint64_t row,
                                                const double2 *logtab) {
  double top = a[0];
#pragma unroll
  for (int b = 1; b < 5; ++b) top = a[b] > top ? a[b] : top;
  const uint64_t base = mix64(seed + (uint64_t)model);
  const float inv_sigma = (float)(1.0 / sigma);
  // branch-free: the five hash chains are independent and interleave; letters out of contention score -inf
  float v[5];
  bool unsure = false, any = false;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    const bool cont = top - a[b] <= 17.5 * sigma;
    bool tiny;
    const float z = evl_gauss_f32(evl_key(base, row * 5 + b), &tiny);
    v[b] = cont ? (float)(a[b] - top) * inv_sigma + z : -INFINITY;
    unsure |= cont && tiny;
    any |= cont;
  }
  if (!any) return 0;   // no contender only when the concentrations are NaN: index 0, as argmax of NaNs does
  int i1 = 0;
  float v1 = v[0], v2 = -INFINITY;
#pragma unroll
  for (int b = 1; b < 5; ++b) {
    const bool gt = v[b] > v1;
    v2 = gt ? v1 : (v[b] > v2 ? v[b] : v2);
    i1 = gt ? b : i1;
    v1 = gt ? v[b] : v1;
  }
  if (unsure || !(v1 - v2 > 2e-4f)) i1 = evl_argmax_exact(a[0], a[1], a[2], a[3], a[4], top, sigma, base, row, logtab);
  return i1;
}

// ------------------------------------------------------------------------------------------------ sorted formulation
// The row-per-thread kernel above runs every wave at the speed of its largest test count (4 % of the HBM roofline on
// the k=13 table).  Here a tile of 256 rows is split into what needs the whole row and what does not:
//   phase 1 (one thread per row): total length, the AR model, and for every DM model the arg-max accuracy -- a few FMAs
//            and compares per model, no lgamma;
//   phase 2: the cells with a non-zero test count, (row, letter) and (row, total), are counting-sorted by
//            min(count, 17) in LDS (LDS atomics on an 18-bin histogram, rank = the atomic's return value);
//   phase 3: one thread per sorted cell evaluates D(x, c) = lgamma(x + c) - lgamma(x) for every DM model of the launch:
//            lanes of a wave hold equal counts, so the product loops are wave-uniform (16 < c takes the Stirling form).
// A launch carries at most EVS_CHUNK DM models (register accumulators); more models are further launches.
#define EVS_THREADS 256
#define EVS_WAVES (EVS_THREADS / 64)
#define EVS_CHUNK 8
#define EVS_KMAX 17
#define EVS_NOUT (2 * EVS_CHUNK + 3)     // compact partial: ll[chunk], cor[chunk], ll_arm, cor_arm, total_len

// kept out of line: inlined twice it pushed the kernel to 225 VGPRs (2 waves per SIMD)
__device__ __noinline__ double evs_item_D(double x, double c, const double2 *tab) { return bear_dm_item_fast(x, c, tab).D; }

__global__ __launch_bounds__(EVS_THREADS) void eval_sorted_kernel(const uint32_t *__restrict__ test,
                                                                   const uint32_t *__restrict__ train,
                                                                   const double *__restrict__ prior, uint64_t n_rows,
                                                                   evl_args A, int m0, int m_cnt, int do_common,
                                                                   const double2 *__restrict__ logtab_g,
                                                                   double *__restrict__ partials) {
  __shared__ double2 s_log[BEAR_LOGTAB_N];
  __shared__ __attribute__((aligned(16))) uint32_t s_tst[EVS_THREADS * 5];
  __shared__ __attribute__((aligned(16))) uint32_t s_trn[EVS_THREADS * 5];
  __shared__ __attribute__((aligned(16))) double s_pri[EVS_THREADS * 5];
  __shared__ uint32_t s_hist[EVS_KMAX + 1], s_off[EVS_KMAX + 2];
  __shared__ uint16_t s_list[EVS_THREADS * 6];
  __shared__ uint16_t s_tie[EVS_THREADS * (EVS_CHUNK + 1)];   // (row, model slot) pairs whose arg-max needs the noise
  __shared__ uint32_t s_ntie;
  __shared__ double s_red[EVS_WAVES][EVS_NOUT];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid < BEAR_LOGTAB_N) s_log[tid] = logtab_g[tid];
  double acc_ll[EVS_CHUNK], acc_cor[EVS_CHUNK], acc_arm = 0.0, acc_carm = 0.0, acc_tot = 0.0;
#pragma unroll
  for (int k = 0; k < EVS_CHUNK; ++k) acc_ll[k] = acc_cor[k] = 0.0;
  const double eps = A.eps, sig_dm = 100.0 * A.eps;
  const uint64_t n_tiles = (n_rows + EVS_THREADS - 1) / EVS_THREADS;
  // concentration of DM model m (global index) for letter b of a staged row
  auto conc = [&](int m, uint32_t row, int b) -> double {
    const double w = A.inv_h[m];
    const double r = A.has_train ? (double)s_trn[row * 5 + b] : 0.0;
    if (m < A.n_h) return __builtin_fma(A.has_prior ? s_pri[row * 5 + b] : 1.0, w, r) + eps;
    return (r + w) + eps;
  };
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * EVS_THREADS;
    const uint32_t rows = (uint32_t)((n_rows - row0 < EVS_THREADS) ? (n_rows - row0) : EVS_THREADS);
    __syncthreads();                                    // previous tile's phase 3 is done with the staged rows / list
    stage_dwords(s_tst, test + row0 * 5, rows * 5);
    if (A.has_train) stage_dwords(s_trn, train + row0 * 5, rows * 5);
    if (A.has_prior)
      stage_dwords(reinterpret_cast<uint32_t *>(s_pri), reinterpret_cast<const uint32_t *>(prior + row0 * 5), rows * 10);
    if (tid <= EVS_KMAX) s_hist[tid] = 0u;
    if (tid == 0) s_ntie = 0u;
    __syncthreads();
    // ---- phase 1 + ranks
    const bool live = tid < rows;
    double t[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) t[b] = live ? (double)s_tst[tid * 5 + b] : 0.0;
    const double n = ((t[0] + t[1]) + (t[2] + t[3])) + t[4];
    uint32_t rank[6], key[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const double c = b < 5 ? t[b] : n;
      key[b] = c > 0.0 ? (c > 16.0 ? (uint32_t)EVS_KMAX : (uint32_t)c) : 0u;
      rank[b] = key[b] ? atomicAdd(&s_hist[key[b]], 1u) : 0u;
    }
    if (do_common) {
      acc_tot += n;
      if (A.arm) {
        double p[5], ll = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          p[b] = (live && A.has_prior ? s_pri[tid * 5 + b] : 1.0) + eps;
          if (t[b] != 0.0) ll = __builtin_fma(t[b], evl_log_any(p[b], s_log), ll);
        }
        acc_arm += ll;
        if (n != 0.0) {
          int im;
          if (evl_argmax_clear(p, eps, im))
            acc_carm += im == 0 ? t[0] : im == 1 ? t[1] : im == 2 ? t[2] : im == 3 ? t[3] : t[4];
          else
            s_tie[atomicAdd(&s_ntie, 1u)] = (uint16_t)(tid * 16u + (uint32_t)EVS_CHUNK);
        }
      }
    }
    if (n != 0.0) {
#pragma unroll 1
      for (int mi = 0; mi < m_cnt; ++mi) {
        const int m = m0 + mi;
        double a[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) a[b] = conc(m, tid, b);
        int im;
        if (evl_argmax_clear(a, sig_dm, im)) {
          // select without dynamic register indexing
          const double hit = im == 0 ? t[0] : im == 1 ? t[1] : im == 2 ? t[2] : im == 3 ? t[3] : t[4];
#pragma unroll
          for (int k = 0; k < EVS_CHUNK; ++k)
            if (k == mi) acc_cor[k] += hit;
        } else {
          s_tie[atomicAdd(&s_ntie, 1u)] = (uint16_t)(tid * 16u + (uint32_t)mi);
        }
      }
    }
    __syncthreads();
    // ---- phase 1b: the undecided (row, model) pairs, densely (ties are common on sparse tables -- equal counts -- and the
    //      noise costs ~500 instructions per pair; inside phase 1 every wave would pay for its unluckiest lane)
    for (uint32_t i = tid; i < s_ntie; i += EVS_THREADS) {
      const uint32_t e = s_tie[i], row = e >> 4, slot = e & 15u;
      double a[5];
      const bool is_arm = slot == (uint32_t)EVS_CHUNK;
      const int m = is_arm ? 0 : m0 + (int)slot;
#pragma unroll
      for (int b = 0; b < 5; ++b) a[b] = is_arm ? (A.has_prior ? s_pri[row * 5 + b] : 1.0) + eps : conc(m, row, b);
      const int im = evl_argmax_noisy(a, is_arm ? eps : sig_dm, A.seed,
                                      is_arm ? EVL_ID_ARM : (m < A.n_h ? (uint32_t)m : EVL_ID_VAN + (uint32_t)(m - A.n_h)),
                                      A.row_base + row0 + row, s_log);
      const double hit = (double)s_tst[row * 5 + im];
      if (slot == (uint32_t)EVS_CHUNK) acc_carm += hit;
#pragma unroll
      for (int k = 0; k < EVS_CHUNK; ++k)
        if ((uint32_t)k == slot) acc_cor[k] += hit;
    }
    // ---- phase 2: bucket offsets (18 bins: one thread), scatter
    if (tid == 0) {
      uint32_t run = 0;
      for (int k = 1; k <= EVS_KMAX; ++k) {
        s_off[k] = run;
        run += s_hist[k];
      }
      s_off[EVS_KMAX + 1] = run;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 6; ++b)
      if (key[b]) s_list[s_off[key[b]] + rank[b]] = (uint16_t)(tid * 8u + (uint32_t)b);
    __syncthreads();
    // ---- phase 3: one thread per sorted cell
    const uint32_t n_items = s_off[EVS_KMAX + 1];
    for (uint32_t i = tid; i < n_items; i += EVS_THREADS) {
      const uint32_t cell = s_list[i], row = cell >> 3, b = cell & 7u;
      // the row's letter (or, for the total, all five): test count, training count and prior entry, read once per cell
      double c, rr[5], ff[5];
      const bool tot = b >= 5u;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const uint32_t at = row * 5 + (tot ? (uint32_t)q : b);
        rr[q] = A.has_train ? (double)s_trn[at] : 0.0;
        ff[q] = A.has_prior ? s_pri[at] : 1.0;
      }
      if (!tot) {
        c = (double)s_tst[row * 5 + b];
      } else {
        c = (((double)s_tst[row * 5] + (double)s_tst[row * 5 + 1]) + ((double)s_tst[row * 5 + 2] + (double)s_tst[row * 5 + 3])) +
            (double)s_tst[row * 5 + 4];
      }
#pragma unroll 1
      for (int mi = 0; mi < m_cnt; ++mi) {
        const int m = m0 + mi;
        const double w = A.inv_h[m];
        double a[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) a[q] = m < A.n_h ? __builtin_fma(ff[q], w, rr[q]) + eps : (rr[q] + w) + eps;
        const double x = tot ? ((a[0] + a[1]) + (a[2] + a[3])) + a[4] : a[0];
        const double d = evs_item_D(x, c, s_log);
        const double v = tot ? -d : d;
#pragma unroll
        for (int k = 0; k < EVS_CHUNK; ++k)
          if (k == mi) acc_ll[k] += v;
      }
    }
  }
  // ---- block reduction -> compact partial
  double vals[EVS_NOUT];
#pragma unroll
  for (int k = 0; k < EVS_CHUNK; ++k) {
    vals[k] = acc_ll[k];
    vals[EVS_CHUNK + k] = acc_cor[k];
  }
  vals[2 * EVS_CHUNK] = acc_arm;
  vals[2 * EVS_CHUNK + 1] = acc_carm;
  vals[2 * EVS_CHUNK + 2] = acc_tot;
#pragma unroll
  for (int k = 0; k < EVS_NOUT; ++k) {
    const double v = bear_wave_sum(vals[k]);
    if (lane == 0) s_red[wave][k] = v;
  }
  __syncthreads();
  if (tid < EVS_NOUT) partials[(size_t)blockIdx.x * EVS_NOUT + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
}

// out[slot[k]] = fixed-order sum over blocks of compact partial k, for the k with slot[k] >= 0
struct evs_slots {
  int slot[EVS_NOUT];
};
__global__ __launch_bounds__(256) void eval_sorted_finalize_kernel(const double *__restrict__ partials, int n_blocks, evs_slots S,
                                                                   double *__restrict__ out) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= EVS_NOUT || S.slot[k] < 0) return;
  double s = 0.0;
  for (int b = lane; b < n_blocks; b += 64) s += partials[(size_t)b * EVS_NOUT + k];
  s = bear_wave_sum(s);
  if (lane == 0) out[S.slot[k]] = s;
}

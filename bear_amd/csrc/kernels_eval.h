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
  double eps;
  uint64_t seed, row_base;                    // noise stream; global index of row 0 (rank shards / batches)
  double inv_h[EVL_MAX_MODELS];               // [0, n_h): 1 / h_j; [n_h, n_h + n_van): v_k
};

__device__ __forceinline__ double evl_normal(uint64_t seed, uint32_t model, uint64_t cell) {
  return gauss(mix64(mix64(seed + (uint64_t)model) ^ cell));
}

// index of the largest of a[b] + sigma z_b (first index on exact ties, as argmax does)
__device__ __forceinline__ int evl_argmax(const double (&a)[5], double sigma, uint64_t seed, uint32_t model, uint64_t row) {
  int i1 = 0;
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
  if (v1 - v2 > 17.5 * sigma) return i1;
  i1 = 0;
  v1 = __builtin_fma(sigma, evl_normal(seed, model, row * 5), a[0]);
#pragma unroll
  for (int b = 1; b < 5; ++b) {
    const double v = __builtin_fma(sigma, evl_normal(seed, model, row * 5 + b), a[b]);
    if (v > v1) {
      v1 = v;
      i1 = b;
    }
  }
  return i1;
}

// DM log-likelihood of test counts t (double, n = sum) under concentrations a
__device__ __forceinline__ double evl_dm_ll(const double (&t)[5], double n, const double (&a)[5], const double2 *logtab) {
  double ll = 0.0;
#pragma unroll
  for (int b = 0; b < 5; ++b)
    if (t[b] != 0.0) ll += bear_dm_item_fast(a[b], t[b], logtab).D;
  if (n != 0.0) ll -= bear_dm_item_fast(((a[0] + a[1]) + (a[2] + a[3])) + a[4], n, logtab).D;
  return ll;
}

__global__ __launch_bounds__(EVL_THREADS) void eval_kernel(const uint32_t *__restrict__ test,
                                                            const uint32_t *__restrict__ train,
                                                            const double *__restrict__ prior, uint64_t n_rows,
                                                            evl_args A, const double2 *__restrict__ logtab_g,
                                                            double *__restrict__ partials) {
  __shared__ double2 s_log[BEAR_LOGTAB_N];
  __shared__ __attribute__((aligned(16))) uint32_t s_tst[EVL_THREADS * 5];
  __shared__ __attribute__((aligned(16))) uint32_t s_trn[EVL_THREADS * 5];
  __shared__ __attribute__((aligned(16))) double s_pri[EVL_THREADS * 5];
  __shared__ double s_acc[EVL_WAVES][EVL_MAX_OUT];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid < BEAR_LOGTAB_N) s_log[tid] = logtab_g[tid];
  const int n_models = A.n_h + A.n_van;
  const int n_out = 2 * n_models + 3;
  for (int k = tid; k < EVL_WAVES * EVL_MAX_OUT; k += EVL_THREADS) (&s_acc[0][0])[k] = 0.0;
  const int o_arm = A.n_h, o_van = A.n_h + 1, o_cor = n_models + 1, o_tot = 2 * n_models + 2;
  const double eps = A.eps, sig_dm = 100.0 * A.eps;
  auto add = [&](int slot, double v) {  // one adder per (wave, slot): deterministic order
    v = bear_wave_sum(v);
    if (lane == 0) s_acc[wave][slot] += v;
  };
  const uint64_t n_tiles = (n_rows + EVL_THREADS - 1) / EVL_THREADS;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * EVL_THREADS;
    const uint32_t rows = (uint32_t)((n_rows - row0 < EVL_THREADS) ? (n_rows - row0) : EVL_THREADS);
    __syncthreads();
    stage_dwords(s_tst, test + row0 * 5, rows * 5);
    if (A.has_train) stage_dwords(s_trn, train + row0 * 5, rows * 5);
    if (A.has_prior)
      stage_dwords(reinterpret_cast<uint32_t *>(s_pri), reinterpret_cast<const uint32_t *>(prior + row0 * 5), rows * 10);
    __syncthreads();
    const bool live = tid < rows;
    double t[5], r[5], f[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      t[b] = live ? (double)s_tst[tid * 5 + b] : 0.0;
      r[b] = (live && A.has_train) ? (double)s_trn[tid * 5 + b] : 0.0;
      f[b] = (live && A.has_prior) ? s_pri[tid * 5 + b] : 1.0;
    }
    const double n = ((t[0] + t[1]) + (t[2] + t[3])) + t[4];
    const uint64_t grow = A.row_base + row0 + tid;
    add(o_tot, n);
    // dead lanes carry t = 0: every term below is then exactly 0 and they only take part in the wave sums
#pragma unroll 1
    for (int m = 0; m < n_models; ++m) {
      const bool ear = m < A.n_h;
      const double w = A.inv_h[m];
      double a[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) a[b] = ear ? (__builtin_fma(f[b], w, r[b]) + eps) : ((r[b] + w) + eps);
      const double ll = evl_dm_ll(t, n, a, s_log);
      double cor = 0.0;
      if (n != 0.0) {
        const int im = evl_argmax(a, sig_dm, A.seed, ear ? (uint32_t)m : EVL_ID_VAN + (uint32_t)(m - A.n_h), grow);
        cor = t[im];
      }
      const int slot = ear ? m : o_van + (m - A.n_h);
      add(slot, ll);
      add(o_cor + slot, cor);
    }
    if (A.arm) {
      double p[5], ll = 0.0, cor = 0.0;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        p[b] = f[b] + eps;
        if (t[b] != 0.0) ll = __builtin_fma(t[b], p[b] > 0.0 ? bear_log_tab(p[b], s_log) : bear_log(p[b]), ll);
      }
      if (n != 0.0) cor = t[evl_argmax(p, eps, A.seed, EVL_ID_ARM, grow)];
      add(o_arm, ll);
      add(o_cor + o_arm, cor);
    }
  }
  __syncthreads();
  for (int k = tid; k < n_out; k += EVL_THREADS)
    partials[(size_t)blockIdx.x * EVL_MAX_OUT + k] = (s_acc[0][k] + s_acc[1][k]) + (s_acc[2][k] + s_acc[3][k]);
}

// fixed-order sum of the block partials: one wave per output (lane-strided partial sums, then the shuffle tree)
__global__ __launch_bounds__(256) void eval_finalize_kernel(const double *__restrict__ partials, int n_blocks, int n_out,
                                                            double *__restrict__ out) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_out) return;
  double s = 0.0;
  for (int b = lane; b < n_blocks; b += 64) s += partials[(size_t)b * EVL_MAX_OUT + k];
  s = bear_wave_sum(s);
  if (lane == 0) out[k] = s;
}

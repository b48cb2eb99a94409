// kernels_sample.h -- posterior sampling of transition probabilities.
//
// Replaces, for a batch of k-mer contexts, the sampling half of get_var_probs.get_pdf
// (bear_model/get_var_probs.py:91-194) and the sampler it calls, log_gamma.log_gamma
// (bear_model/log_gamma.py:17-76):
//   concs[m, k, :] = ar_vals[k, :] / h_m + counts[k, :]      (BEAR models, get_var_probs.py:141-149)
//                  = van_m + counts[k, :]                    (vanilla models, :138, :149)
//                  = ar_vals[k, :]                           (AR model, MAP only, :150-153)
//   sampling:  g ~ logGamma(concs) independently, log_probs = g - logsumexp_b g   (:176-178)
//              = the log of a Dirichlet(concs) draw, taken in log space so that tiny
//              concentrations do not underflow (the point of log_gamma.py)
//   MAP:       log_probs = log(concs / sum_b concs)                               (:174-175)
// written in the reference's output='numpy' layout [k-mer, letter, model, mc_sample] (:181-183).
//
// The reference draws from numpy's global Mersenne twister by vectorised rejection rounds; a GPU has no
// such stream.  Here every draw is a pure function of (seed, model, sample, global row, letter):
// Marsaglia-Tsang squeeze-free rejection for shape >= 1 evaluated in log space, and for shape a < 1 the
// boost  log G_a = log G_{a+1} + log(U) / a,  which is exact and, like the reference's scheme, never forms
// exp() of a large negative number.  Same distribution as the reference (tests: the reference's own KS
// criterion and its closed-form Beta expectations); the oracle restates this hash sampler bit for bit.
#pragma once
#include "bear_common.h"
#include "kernels_synth.h"

#define SMP_THREADS 256
#define SMP_MAX_MODELS 64
#define SMP_MAX_ROUNDS 64   // acceptance >= 0.95 per round: 64 misses in a row has probability < 1e-83

struct smp_args {
  int n_h, n_van, arm, has_counts, has_prior, map;
  uint32_t mc;
  uint64_t seed, row_base;
  double w[SMP_MAX_MODELS];   // [0, n_h): 1 / h_j ; [n_h, n_h + n_van): van_k
};

// one logGamma(a) draw, a > 0, from the counter stream `key`
__device__ __forceinline__ double smp_log_gamma(double a, uint64_t key) {
  const bool small = a < 1.0;
  const double a1 = small ? a + 1.0 : a;
  const double d = a1 - 1.0 / 3.0;
  const double c = 1.0 / sqrt(9.0 * d);
  const double ld = log(d);
  double lg = ld;
  for (uint32_t t = 0; t < SMP_MAX_ROUNDS; ++t) {
    const double x = gauss(key + 2ull * t + 1ull);
    const double u = u01(mix64(key + 2ull * t + 2ull));
    const double w = __builtin_fma(c, x, 1.0);
    if (w <= 0.0) continue;
    const double v = w * w * w;
    const double lv = log(v);
    lg = ld + lv;
    if (log(u) < __builtin_fma(0.5 * x, x, d) - d * v + d * lv) break;
  }
  if (small) lg += log(u01(mix64(key))) / a;
  return lg;
}

__device__ __forceinline__ uint64_t smp_cell_key(uint64_t seed, uint32_t model, uint64_t sample, uint64_t cell) {
  return mix64(mix64(mix64(seed + (uint64_t)model) + sample) ^ cell);
}

// log_gamma.log_gamma(concs, size): out[s * n + i] = logGamma(conc[i]) draw, s < n_samples
__global__ __launch_bounds__(SMP_THREADS) void log_gamma_kernel(const double *__restrict__ conc, uint64_t n,
                                                                uint64_t n_samples, uint64_t seed,
                                                                double *__restrict__ out) {
  const uint64_t total = n * n_samples;
  for (uint64_t e = (uint64_t)blockIdx.x * SMP_THREADS + threadIdx.x; e < total; e += (uint64_t)gridDim.x * SMP_THREADS) {
    const uint64_t s = e / n, i = e - s * n;
    out[e] = smp_log_gamma(conc[i], smp_cell_key(seed, 0u, s, i));
  }
}

// one thread per (row, model, sample): 5 draws, normalise, 5 strided stores (consecutive lanes = consecutive samples)
__global__ __launch_bounds__(SMP_THREADS) void logdir_sample_kernel(const uint32_t *__restrict__ counts,
                                                                    const double *__restrict__ prior, uint64_t n_rows,
                                                                    smp_args A, double *__restrict__ out) {
  const uint32_t M = (uint32_t)(A.arm + A.n_h + A.n_van);
  const uint64_t per_row = (uint64_t)M * A.mc;
  const uint64_t total = n_rows * per_row;
  for (uint64_t e = (uint64_t)blockIdx.x * SMP_THREADS + threadIdx.x; e < total; e += (uint64_t)gridDim.x * SMP_THREADS) {
    const uint64_t k = e / per_row;
    const uint32_t rem = (uint32_t)(e - k * per_row);
    const uint32_t m = rem / A.mc, s = rem - m * A.mc;
    const int j = (int)m - A.arm;           // -1: the AR model; [0, n_h): BEAR; then vanilla
    double a[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const double cnt = A.has_counts ? (double)counts[k * 5 + b] : 0.0;
      const double f = A.has_prior ? prior[k * 5 + b] : 0.0;
      a[b] = j < 0 ? f : (j < A.n_h ? __builtin_fma(f, A.w[j], cnt) : A.w[j] + cnt);
    }
    double g[5];
    if (A.map) {
      const double tot = ((a[0] + a[1]) + (a[2] + a[3])) + a[4];
#pragma unroll
      for (int b = 0; b < 5; ++b) g[b] = log(a[b] / tot);
    } else {
      const uint64_t grow = A.row_base + k;
      double mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        g[b] = smp_log_gamma(a[b], smp_cell_key(A.seed, m, s, grow * 5 + b));
        mx = g[b] > mx ? g[b] : mx;
      }
      double se = 0.0;
#pragma unroll
      for (int b = 0; b < 5; ++b) se += exp(g[b] - mx);
      const double lse = mx + log(se);
#pragma unroll
      for (int b = 0; b < 5; ++b) g[b] -= lse;
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) out[((k * 5 + b) * M + m) * A.mc + s] = g[b];
  }
}

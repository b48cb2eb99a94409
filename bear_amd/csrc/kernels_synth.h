// kernels_synth.h -- synthetic count-table / prior generators (measurement tooling).
#pragma once
#include "bear_common.h"

// ------------------------------------------------------------------ synthetic table
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t h) { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
__device__ __forceinline__ double gauss(uint64_t k) {
  return sqrt(-2.0 * log(u01(mix64(k)))) * cos(6.283185307179586 * u01(mix64(k ^ 0x5851F42D4C957F2Dull)));
}
__device__ uint32_t poisson(double mu, uint64_t k) {
  if (!(mu > 0.0)) return 0u;
  if (mu < 12.0) {
    double u = u01(mix64(k)), p = exp(-mu), s = p;
    uint32_t n = 0;
    while (u > s && n < 200u) {
      ++n;
      p *= mu / (double)n;
      s += p;
    }
    return n;
  }
  double v = floor(mu + sqrt(mu) * gauss(k) + 0.5);
  return v > 0.0 ? (v < 4.0e9 ? (uint32_t)v : 4000000000u) : 0u;
}

__global__ void synth_counts_kernel(uint64_t seed, uint64_t row0, uint64_t n_rows, int dense, uint32_t *train,
                                    uint32_t *test, uint32_t *ref) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const uint64_t key = mix64(seed ^ mix64(row0 + i));
  double lam;
  if (dense) {
    lam = 1.0e4 * exp(u01(mix64(key + 1)) * 3.4011973816621555);  // 1e4 .. 3e5 (ysd1-like)
  } else {
    lam = exp(0.5 + 1.5 * gauss(key + 1));  // "k=13 sparse": median 1.65 transitions per context
  }
  double w[4], ws = 0.0;
  for (int b = 0; b < 4; ++b) {
    // ~Gamma(0.3) weights: spiky next-base distributions
    w[b] = -log(u01(mix64(key + 10 + b))) * pow(u01(mix64(key + 20 + b)), 10.0 / 3.0);
    if (dense) w[b] += 0.15;
    ws += w[b];
  }
  double p[5];
  for (int b = 0; b < 4; ++b) p[b] = w[b] / ws * (1.0 - 1.0 / 150.0);
  p[4] = 1.0 / 150.0;  // read length 150 (docs/usage.rst:289-291)
  for (int b = 0; b < 5; ++b) {
    if (train) train[i * 5 + b] = poisson(lam * p[b], key + 100 + b);
    if (test) test[i * 5 + b] = poisson(lam * p[b] / 3.0, key + 200 + b);
    if (ref) ref[i * 5 + b] = b < 4 ? poisson((dense ? 0.001 : 0.02) * lam * p[b], key + 300 + b) : 0u;
  }
}

__global__ void synth_prior_kernel(uint64_t seed, uint64_t row0, uint64_t n_rows, double *prior) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const uint64_t key = mix64(~seed ^ mix64(row0 + i));
  double e[5], s = 0.0;
  for (int b = 0; b < 5; ++b) {
    double z = 2.0 * (u01(mix64(key + 400 + b)) - 0.5) - (b == 4 ? 3.0 : 0.0);
    e[b] = exp(z);
    s += e[b];
  }
  for (int b = 0; b < 5; ++b) prior[i * 5 + b] = e[b] / s;
}


// ------------------------------------------------------------------ read-only stream (measurement)
// What a pure read of `n16` 16-byte words costs on this device: the measured ceiling next to the 8 TB/s spec peak.
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4 *__restrict__ src, uint64_t n16, uint32_t *__restrict__ sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    const uint4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9E3779B9u) sink[0] = acc;   // never true in practice: keeps the loads alive without a store per thread
}

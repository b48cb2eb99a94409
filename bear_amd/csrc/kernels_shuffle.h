// kernels_shuffle.h -- on-device row shuffle of a resident count table.
//
// The reference asks for tables shuffled on disk with `shuf` before training (docs/usage.rst:191-200: rows of one
// summarize.py file come out in k-mer order per bin, and a batch of neighbouring k-mers is not a sample of the
// table).  Here the permutation is a keyed bijection of [0, n): a 4-round Feistel network on the smallest even
// number of bits covering n, cycle-walked back into range (expected < 4 evaluations), so any rank can compute any
// row's source without a permutation table, and the oracle restates it (bear_oracle.py:shuffle_perm).
// dst[i] = src[perm(i)] as one gather pass: rows of `row_bytes` bytes (20 for a count slab, 8 for packed k-mers,
// lag for k-mer bytes).
#pragma once
#include "bear_common.h"
#include "kernels_synth.h"

__host__ __device__ __forceinline__ uint64_t shf_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// half_bits: bits per Feistel half (domain 2^(2 half_bits) >= n)
__host__ __device__ __forceinline__ uint64_t shf_perm(uint64_t i, uint64_t n, uint32_t half_bits, uint64_t seed) {
  const uint64_t mask = (1ull << half_bits) - 1ull;
  uint64_t x = i;
  do {
    uint64_t l = x >> half_bits, r = x & mask;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint64_t f = shf_mix64(seed ^ (r + ((uint64_t)(k + 1) << 58))) & mask;
      const uint64_t t = l ^ f;
      l = r;
      r = t;
    }
    x = (l << half_bits) | r;
  } while (x >= n);
  return x;
}

static inline uint32_t shf_half_bits(uint64_t n) {
  uint32_t b = 1;
  while (b < 32 && (1ull << (2 * b)) < n) ++b;
  return b;
}

// one thread per 4-byte word of the destination (row_bytes % 4 == 0) -- coalesced stores, gathered loads
__global__ __launch_bounds__(256) void shuffle_words_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                            uint64_t n_rows, uint32_t row_words, uint32_t half_bits,
                                                            uint64_t seed) {
  const uint64_t total = n_rows * row_words;
  for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
    const uint64_t i = e / row_words;
    const uint32_t w = (uint32_t)(e - i * row_words);
    dst[e] = src[shf_perm(i, n_rows, half_bits, seed) * row_words + w];
  }
}

__global__ __launch_bounds__(256) void shuffle_bytes_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                            uint64_t n_rows, uint32_t row_bytes, uint32_t half_bits,
                                                            uint64_t seed) {
  const uint64_t total = n_rows * row_bytes;
  for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
    const uint64_t i = e / row_bytes;
    const uint32_t w = (uint32_t)(e - i * row_bytes);
    dst[e] = src[shf_perm(i, n_rows, half_bits, seed) * row_bytes + w];
  }
}

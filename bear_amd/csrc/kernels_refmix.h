// kernels_refmix.h -- bear_ref's prior rows for a parametrised net function, forward and backward (bear_ref.py:9-33, 63-68):
//
//   f_i = (nw g_i + jc_i) / (nw + 1),   jc_ib = shape_b / 4 + exp(-tau) (r_ib / sum_c |r_ic| - shape_b / 4),   shape = (1,1,1,1,0)
//
// g_i = the net function's row (linear, cnn), r_i = the reference-count row as the driver hands it over ((counts + eps) with the
// stop column zeroed, bear_ref.py:332-337), tau = exp(tau_signed), nw = exp(net_weight_signed).  With the stop net function the
// whole of this lives inside the mode-R DM kernels; with a net function that has parameters the rows exist, and as torch ops the
// mixing and its autograd cost 30 ms per 1e8 contexts (a dozen passes over [n, 5] temporaries) next to a 1.5 ms DM step.
//   forward : one pass, 40 + 40 B read and 40 B written per context; rows enter and leave a wave through LDS as 16-byte pieces;
//   backward: Q_i = d L / d f_i in, d L / d g_i = Q_i nw / (nw + 1) out, and the two sums
//             A = sum_ib Q_ib (g_ib - jc_ib),  B = sum_ib Q_ib (r_ib / sum |r_i| - shape_b / 4)
//             from which d L / d net_weight_signed = nw A / (nw + 1)^2 and d L / d tau_signed = -tau exp(-tau) B / (nw + 1)
//             (the last block to finish writes both).
// The parameters are read from device memory (the optimizer's tensors): no host round trip.
#pragma once
#include "bear_common.h"

#define RMX_THREADS 1024
#define RMX_WAVES (RMX_THREADS / 64)

struct rmx_consts {
  double nw, V, E, tau;
};
__device__ __forceinline__ rmx_consts rmx_load(const double *__restrict__ tau_signed, const double *__restrict__ nw_signed) {
  rmx_consts c;
  c.tau = bear_uniform_f64(exp(tau_signed[0]));
  c.nw = bear_uniform_f64(exp(nw_signed[0]));
  c.E = bear_uniform_f64(exp(-c.tau));
  c.V = bear_uniform_f64(1.0 / (c.nw + 1.0));
  return c;
}
// d_b = r_b / sum |r| - shape_b / 4: the part of the Jukes-Cantor row that exp(-tau) scales
__device__ __forceinline__ void rmx_dev(const double (&r)[5], double (&d)[5]) {
  const double l1 = ((__builtin_fabs(r[0]) + __builtin_fabs(r[1])) + (__builtin_fabs(r[2]) + __builtin_fabs(r[3]))) + __builtin_fabs(r[4]);
#pragma unroll
  for (int b = 0; b < 5; ++b) d[b] = r[b] / l1 - (b < 4 ? 0.25 : 0.0);
}

__global__ __launch_bounds__(RMX_THREADS) void ref_mix_forward_kernel(const double *__restrict__ net_rows, const double *__restrict__ ref_rows,
                                                                      const double *__restrict__ tau_signed,
                                                                      const double *__restrict__ nw_signed, uint64_t n,
                                                                      double *__restrict__ prior) {
  __shared__ __attribute__((aligned(16))) double rows[RMX_WAVES][64 * 5];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const rmx_consts C = rmx_load(tau_signed, nw_signed);
  const uint64_t n_chunks = (n + 63u) >> 6;
  for (uint64_t c = (uint64_t)blockIdx.x * RMX_WAVES + wave; c < n_chunks; c += (uint64_t)gridDim.x * RMX_WAVES) {
    const uint64_t i0 = c << 6;
    const uint32_t valid = n - i0 < 64u ? (uint32_t)(n - i0) : 64u;
    double g[5], r[5], d[5], f[5];
    bear_wave_load_rows5(rows[wave], net_rows, i0, valid, lane, g);   // (as 8-byte strided loads: 2.38 instead of 2.21 ms per 1e8)
    bear_wave_load_rows5(rows[wave], ref_rows, i0, valid, lane, r);
    rmx_dev(r, d);
#pragma unroll
    for (int b = 0; b < 5; ++b) f[b] = (C.nw * g[b] + ((b < 4 ? 0.25 : 0.0) + C.E * d[b])) * C.V;
    bear_wave_store_rows5(rows[wave], f, prior, i0, valid, lane);
  }
}

__global__ __launch_bounds__(RMX_THREADS) void ref_mix_backward_kernel(const double *__restrict__ net_rows, const double *__restrict__ ref_rows,
                                                                       const double *__restrict__ grad_prior,
                                                                       const double *__restrict__ tau_signed,
                                                                       const double *__restrict__ nw_signed, uint64_t n,
                                                                       double *__restrict__ grad_net_rows, double *__restrict__ partials,
                                                                       const bear_arrival arrive, double *__restrict__ grad_scalars) {
  __shared__ __attribute__((aligned(16))) double rows[RMX_WAVES][64 * 5];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const rmx_consts C = rmx_load(tau_signed, nw_signed);
  const double nwV = C.nw * C.V;
  double acc[2] = {0.0, 0.0};
  const uint64_t n_chunks = (n + 63u) >> 6;
  for (uint64_t c = (uint64_t)blockIdx.x * RMX_WAVES + wave; c < n_chunks; c += (uint64_t)gridDim.x * RMX_WAVES) {
    const uint64_t i0 = c << 6;
    const uint32_t valid = n - i0 < 64u ? (uint32_t)(n - i0) : 64u;
    const bool in = lane < valid;
    double g[5], r[5], q[5], d[5], dg[5];
    bear_wave_load_rows5(rows[wave], net_rows, i0, valid, lane, g);   // (as 8-byte strided loads: 3.41 instead of 3.03 ms per 1e8)
    bear_wave_load_rows5(rows[wave], ref_rows, i0, valid, lane, r);
    bear_wave_load_rows5(rows[wave], grad_prior, i0, valid, lane, q);
#pragma unroll
    for (int b = 0; b < 5; ++b) q[b] = in ? q[b] : 0.0;
    rmx_dev(r, d);
    double a = 0.0, bsum = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const double jc = (b < 4 ? 0.25 : 0.0) + C.E * d[b];
      a = __builtin_fma(q[b], g[b] - jc, a);
      bsum = __builtin_fma(q[b], d[b], bsum);
      dg[b] = q[b] * nwV;
    }
    // rows whose gradient is all zero (contexts without counts) add exact zeros -- also when their reference row is degenerate
    const bool any = (q[0] != 0.0) | (q[1] != 0.0) | (q[2] != 0.0) | (q[3] != 0.0) | (q[4] != 0.0);
    acc[0] += any ? a : 0.0;
    acc[1] += any ? bsum : 0.0;
    bear_wave_store_rows5(rows[wave], dg, grad_net_rows, i0, valid, lane);
  }
  block_store_partials<2, true>(acc, partials);
  if (!bear_arrive_last(arrive)) return;
  // the fixed-order sum of bear_finalize_in_block, then the chain rule of the two signed parameters
  __shared__ double sums[2];
  bear_finalize_in_block(partials, 2, sums, arrive);
  __syncthreads();
  if (tid == 0) {
    grad_scalars[0] = -C.tau * C.E * C.V * sums[1];   // d L / d tau_signed
    grad_scalars[1] = C.nw * C.V * C.V * sums[0];     // d L / d net_weight_signed
  }
}

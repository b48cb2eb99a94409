// bear_common.h -- workspace, launch parameters and block-level reduction shared by all kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/bear_hip.h"
#include "bear_math.h"

#define BEAR_THREADS 256
#define BEAR_TILE_ROWS 1024
#define BEAR_ROWS_PER_THREAD (BEAR_TILE_ROWS / BEAR_THREADS)
#define BEAR_MAX_OUT 4

static thread_local int g_last_hip_error = 0;

#define HIP_TRY(expr)                      \
  do {                                     \
    hipError_t _e = (expr);                \
    if (_e != hipSuccess) {                \
      g_last_hip_error = (int)_e;          \
      return BEAR_ERR_HIP;                 \
    }                                      \
  } while (0)

struct bear_params;
struct bear_ws {
  int device;
  int num_cu;
  int max_blocks;
  double *partials;  // [max_blocks][BEAR_MAX_OUT]
  double *logtab;    // [BEAR_LOGTAB_N][2] = {r_i, -log r_i} (bear_math.h, bear_log_tab)
  unsigned long long *dbg;  // developer timing buffer [max_blocks][8][6]
  double *eval_partials;    // [eval_blocks][EVL_MAX_OUT] (kernels_eval.h)
  double *eval_out;         // [EVL_MAX_OUT] scratch result vector (bear_bmm_f64)
  int eval_blocks;
  double *lin_partials;     // [num_cu][LIN_MAX_GRAD] d/d mat partials (kernels_linear.h)
  struct bear_params *ref_prm;     // device copy of the mode-R constants (bear_ref_train_step_f64: graph replay)
  double *cnn_partials;     // [cnn_blocks][cnn total] parameter-gradient partials (kernels_cnn.h), grown on demand
  size_t cnn_partials_cap;  // doubles
};

struct bear_params {
  double inv_h;   // 1 / exp(h_signed)
  double eps;
  // mode R (bear_ref.py:63-68 with the stop net function)
  double E;       // exp(-tau)
  double tauE;    // tau * exp(-tau)
  double tau;     // exp(tau_signed)
  double V;       // 1 / (nw + 1)
  double nw;      // exp(net_weight_signed)
};

// ------------------------------------------------------------------ block reduction
// Sums NOUT per-thread accumulators over the block (any block size that is a multiple of 64, up to
// 1024) and stores one partial per block.
template <int NOUT>
__device__ __forceinline__ void block_store_partials(double (&acc)[NOUT], double *partials) {
  __shared__ double red[16][BEAR_MAX_OUT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    double v = bear_wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NOUT) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w][threadIdx.x];
    partials[(size_t)blockIdx.x * BEAR_MAX_OUT + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------ finalize: fixed-order sum of block partials
__global__ __launch_bounds__(256) void finalize_kernel(const double *__restrict__ partials, int n_blocks,
                                                       int n_out, double *__restrict__ out) {
  __shared__ double red[4][BEAR_MAX_OUT];
  double acc[BEAR_MAX_OUT] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < n_blocks; b += 256)
#pragma unroll
    for (int k = 0; k < BEAR_MAX_OUT; ++k)
      if (k < n_out) acc[k] += partials[(size_t)b * BEAR_MAX_OUT + k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < BEAR_MAX_OUT; ++k) {
    double v = bear_wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < n_out) out[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}


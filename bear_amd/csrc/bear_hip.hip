// bear_hip.hip -- kernels + C ABI (include/bear_hip.h) for the BEAR training hot path on gfx950.
//
// v1 layout of one launch ("row-per-thread"):
//   grid  = min(#tiles, 2 * #CU) persistent blocks of 256 threads, grid-stride over tiles
//   tile  = TILE_ROWS consecutive k-mer contexts; its count rows (20 B each) and prior rows
//           (40 B each) are fetched as one flat, fully coalesced stream of 16-byte lane loads
//           into LDS, then each thread reads whole rows back (stride 5 dwords / 5 doubles:
//           conflict-free, gcd(5,32) = 1)
//   math  = bear_math.h (product / Stirling evaluation of the lgamma and digamma differences)
//   sums  = per-thread fp64 accumulators -> wave shuffle -> LDS -> one partial per block in the
//           workspace -> single-block finalize kernel in fixed order (no atomics: bitwise
//           reproducible for a given grid).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
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

struct bear_ws {
  int device;
  int num_cu;
  int max_blocks;
  double *partials;  // [max_blocks][BEAR_MAX_OUT]
};

struct bear_params {
  double inv_h;   // 1 / exp(h_signed)
  double eps;
  // mode R (bear_ref.py:63-68 with the stop net function)
  double E;       // exp(-tau)
  double tauE;    // tau * exp(-tau)
  double V;       // 1 / (nw + 1)
  double nw;      // exp(net_weight_signed)
};

// ------------------------------------------------------------------ tile staging
// Copies `n_dwords` dwords starting at src (16-byte aligned) into LDS with 16-byte lane
// loads; the (< 4 dword) tail and anything beyond `n_dwords` is handled dword-wise.
__device__ __forceinline__ void stage_dwords(uint32_t *lds, const uint32_t *src, uint32_t n_dwords) {
  const uint32_t n_vec = n_dwords >> 2;
  const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
  uint4 *d4 = reinterpret_cast<uint4 *>(lds);
  for (uint32_t i = threadIdx.x; i < n_vec; i += BEAR_THREADS) d4[i] = s4[i];
  for (uint32_t i = (n_vec << 2) + threadIdx.x; i < n_dwords; i += BEAR_THREADS) lds[i] = src[i];
}

// ------------------------------------------------------------------ row math
// BEAR mode: LL_i and g_b = dLL_i/dalpha_b from counts c[5] and concentrations a[5].
__device__ __forceinline__ double dm_row(const uint32_t (&c)[5], const double (&a)[5], double (&g)[5]) {
  uint32_t n = c[0] + c[1] + c[2] + c[3] + c[4];
  double ll = 0.0;
#pragma unroll
  for (int b = 0; b < 5; ++b) g[b] = 0.0;
  if (n == 0) return 0.0;
  double A = ((a[0] + a[1]) + (a[2] + a[3])) + a[4];
  bear_dp tn = bear_dm_item(A, n);
  ll = -tn.D;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    g[b] = -tn.P;
    if (c[b] != 0) {
      bear_dp tb = bear_dm_item(a[b], c[b]);
      ll += tb.D;
      g[b] += tb.P;
    }
  }
  return ll;
}

template <int NOUT>
__device__ __forceinline__ void block_store_partials(double (&acc)[NOUT], double *partials) {
  __shared__ double red[BEAR_THREADS / 64][BEAR_MAX_OUT];
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
    for (int w = 0; w < BEAR_THREADS / 64; ++w) s += red[w][threadIdx.x];
    partials[(size_t)blockIdx.x * BEAR_MAX_OUT + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------ mode N: counts + prior rows
template <bool AR, bool GRAD>
__global__ __launch_bounds__(BEAR_THREADS) void dm_prior_kernel(const uint32_t *__restrict__ counts,
                                                                 const double *__restrict__ prior,
                                                                 uint64_t n_rows, bear_params prm,
                                                                 double *__restrict__ grad_prior,
                                                                 double *__restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint32_t s_cnt[BEAR_TILE_ROWS * 5];
  __shared__ __attribute__((aligned(16))) double s_pri[BEAR_TILE_ROWS * 5];
  const uint64_t n_tiles = (n_rows + BEAR_TILE_ROWS - 1) / BEAR_TILE_ROWS;
  double acc[2] = {0.0, 0.0};
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * BEAR_TILE_ROWS;
    const uint32_t rows = (uint32_t)((n_rows - row0 < BEAR_TILE_ROWS) ? (n_rows - row0) : BEAR_TILE_ROWS);
    __syncthreads();  // previous tile fully consumed
    stage_dwords(s_cnt, counts + row0 * 5, rows * 5);
    stage_dwords(reinterpret_cast<uint32_t *>(s_pri), reinterpret_cast<const uint32_t *>(prior + row0 * 5),
                 rows * 10);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < BEAR_ROWS_PER_THREAD; ++k) {
      const uint32_t r = threadIdx.x + k * BEAR_THREADS;
      if (r >= rows) break;
      uint32_t c[5];
      double f[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        c[b] = s_cnt[r * 5 + b];
        f[b] = s_pri[r * 5 + b];
      }
      if (AR) {
        // core.py:138-139 with probs = prior + eps (bear_net.py:68)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          double p = f[b] + prm.eps;
          double cb = (double)c[b];
          if (c[b] != 0) acc[0] += cb * bear_log(p);
          if (GRAD) grad_prior[(row0 + r) * 5 + b] = c[b] != 0 ? cb * bear_rcp(p) : 0.0;
        }
      } else {
        double a[5], g[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) a[b] = __builtin_fma(f[b], prm.inv_h, prm.eps);
        acc[0] += dm_row(c, a, g);
        double dh = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          dh = __builtin_fma(g[b], f[b], dh);
          if (GRAD) grad_prior[(row0 + r) * 5 + b] = g[b] * prm.inv_h;
        }
        acc[1] -= dh * prm.inv_h;  // d alpha_b / d h_signed = -f_b / h
      }
    }
  }
  block_store_partials<2>(acc, partials);
}

// ------------------------------------------------------------------ mode R: train + reference counts
template <bool AR>
__global__ __launch_bounds__(BEAR_THREADS) void dm_ref_kernel(const uint32_t *__restrict__ train,
                                                               const uint32_t *__restrict__ ref,
                                                               uint64_t n_rows, bear_params prm,
                                                               double *__restrict__ partials) {
  __shared__ __attribute__((aligned(16))) uint32_t s_trn[BEAR_TILE_ROWS * 5];
  __shared__ __attribute__((aligned(16))) uint32_t s_ref[BEAR_TILE_ROWS * 5];
  const uint64_t n_tiles = (n_rows + BEAR_TILE_ROWS - 1) / BEAR_TILE_ROWS;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t row0 = tile * BEAR_TILE_ROWS;
    const uint32_t rows = (uint32_t)((n_rows - row0 < BEAR_TILE_ROWS) ? (n_rows - row0) : BEAR_TILE_ROWS);
    __syncthreads();
    stage_dwords(s_trn, train + row0 * 5, rows * 5);
    stage_dwords(s_ref, ref + row0 * 5, rows * 5);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < BEAR_ROWS_PER_THREAD; ++k) {
      const uint32_t r = threadIdx.x + k * BEAR_THREADS;
      if (r >= rows) break;
      uint32_t c[5];
      double rr[4];
#pragma unroll
      for (int b = 0; b < 5; ++b) c[b] = s_trn[r * 5 + b];
#pragma unroll
      for (int b = 0; b < 4; ++b) rr[b] = (double)s_ref[r * 5 + b] + prm.eps;  // bear_ref.py:335-337
      // bear_ref.py:30-33: L1-normalise, Jukes-Cantor; bear_ref.py:63-68: mix with the stop net
      const double invR = bear_rcp((rr[0] + rr[1]) + (rr[2] + rr[3]));
      double f[5], dft[5], dfn[5];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double dev = __builtin_fma(rr[b], invR, -0.25);  // norm_b - 1/4
        f[b] = __builtin_fma(prm.E, dev, 0.25) * prm.V;
        dft[b] = -prm.tauE * dev * prm.V;                 // d f_b / d tau_signed
        dfn[b] = -prm.nw * f[b] * prm.V;                  // d f_b / d nu_signed (g_net = 0)
      }
      f[4] = prm.nw * prm.V;
      dft[4] = 0.0;
      dfn[4] = prm.nw * (1.0 - f[4]) * prm.V;
      double dLdf[5];
      if (AR) {
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          double p = f[b] + prm.eps;
          double cb = (double)c[b];
          dLdf[b] = 0.0;
          if (c[b] != 0) {
            acc[0] += cb * bear_log(p);
            dLdf[b] = cb * bear_rcp(p);
          }
        }
      } else {
        double a[5], g[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) a[b] = __builtin_fma(f[b], prm.inv_h, prm.eps);
        acc[0] += dm_row(c, a, g);
        double dh = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          dLdf[b] = g[b] * prm.inv_h;
          dh = __builtin_fma(dLdf[b], f[b], dh);
        }
        acc[1] -= dh;
      }
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        acc[2] = __builtin_fma(dLdf[b], dft[b], acc[2]);
        acc[3] = __builtin_fma(dLdf[b], dfn[b], acc[3]);
      }
    }
  }
  block_store_partials<4>(acc, partials);
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

// ------------------------------------------------------------------ C ABI
extern "C" {

int bear_abi_version(void) { return BEAR_ABI_VERSION; }

const char *bear_strerror(int status) {
  switch (status) {
    case BEAR_OK: return "ok";
    case BEAR_ERR_INVALID_ARG: return "invalid argument (null or misaligned pointer, bad flag)";
    case BEAR_ERR_NO_DEVICE: return "no usable HIP device";
    case BEAR_ERR_WRONG_DEVICE: return "current HIP device differs from the workspace device";
    case BEAR_ERR_HIP: return "HIP runtime call failed (see bear_last_hip_error)";
    case BEAR_ERR_NOMEM: return "out of memory";
    case BEAR_ERR_IO: return "file could not be opened or read";
    case BEAR_ERR_PARSE: return "malformed count-table row";
    default: return "unknown bear status";
  }
}

int bear_last_hip_error(void) { return g_last_hip_error; }

int bear_ws_create(int device, bear_ws **out) {
  if (!out) return BEAR_ERR_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return BEAR_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return BEAR_ERR_INVALID_ARG;
  int prev = 0;
  HIP_TRY(hipGetDevice(&prev));
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  bear_ws *ws = nullptr;
  if (e == hipSuccess) {
    ws = new (std::nothrow) bear_ws();
    if (ws) {
      ws->device = device;
      ws->num_cu = prop.multiProcessorCount;
      ws->max_blocks = ws->num_cu * 8;
      e = hipMalloc(&ws->partials, sizeof(double) * BEAR_MAX_OUT * (size_t)ws->max_blocks);
    }
  }
  (void)hipSetDevice(prev);
  if (!ws) return e == hipSuccess ? BEAR_ERR_NOMEM : (g_last_hip_error = (int)e, BEAR_ERR_HIP);
  if (e != hipSuccess) {
    delete ws;
    g_last_hip_error = (int)e;
    return BEAR_ERR_HIP;
  }
  *out = ws;
  return BEAR_OK;
}

int bear_ws_destroy(bear_ws *ws) {
  if (!ws) return BEAR_OK;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(ws->device);
  (void)hipFree(ws->partials);
  (void)hipSetDevice(prev);
  delete ws;
  return BEAR_OK;
}

static int check_ws(const bear_ws *ws) {
  if (!ws) return BEAR_ERR_INVALID_ARG;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return BEAR_ERR_NO_DEVICE;
  if (dev != ws->device) return BEAR_ERR_WRONG_DEVICE;
  return BEAR_OK;
}

static inline bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

static int grid_for(const bear_ws *ws, uint64_t n_rows) {
  uint64_t tiles = (n_rows + BEAR_TILE_ROWS - 1) / BEAR_TILE_ROWS;
  uint64_t g = (uint64_t)ws->num_cu * 2;
  if (g > (uint64_t)ws->max_blocks) g = ws->max_blocks;
  if (tiles < g) g = tiles;
  return g < 1 ? 1 : (int)g;
}

int bear_dm_prior_f64(bear_ws *ws, const uint32_t *counts, const double *prior, uint64_t n_rows,
                      double h_signed, double eps, int train_ar, double *out, double *grad_prior,
                      void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!out || (n_rows && (!counts || !prior))) return BEAR_ERR_INVALID_ARG;
  if (misaligned(counts) || misaligned(prior) || misaligned(grad_prior) || (reinterpret_cast<uintptr_t>(out) & 7u))
    return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.inv_h = 1.0 / exp(h_signed);
  prm.eps = eps;
  const int grid = grid_for(ws, n_rows);
  if (train_ar) {
    if (grad_prior)
      hipLaunchKernelGGL((dm_prior_kernel<true, true>), dim3(grid), dim3(BEAR_THREADS), 0, s, counts, prior, n_rows, prm, grad_prior, ws->partials);
    else
      hipLaunchKernelGGL((dm_prior_kernel<true, false>), dim3(grid), dim3(BEAR_THREADS), 0, s, counts, prior, n_rows, prm, grad_prior, ws->partials);
  } else {
    if (grad_prior)
      hipLaunchKernelGGL((dm_prior_kernel<false, true>), dim3(grid), dim3(BEAR_THREADS), 0, s, counts, prior, n_rows, prm, grad_prior, ws->partials);
    else
      hipLaunchKernelGGL((dm_prior_kernel<false, false>), dim3(grid), dim3(BEAR_THREADS), 0, s, counts, prior, n_rows, prm, grad_prior, ws->partials);
  }
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, ws->partials, grid, 2, out);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_dm_ref_f64(bear_ws *ws, const uint32_t *train, const uint32_t *ref, uint64_t n_rows,
                    double h_signed, double tau_signed, double nu_signed, double eps, int train_ar,
                    double *out, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!out || (n_rows && (!train || !ref))) return BEAR_ERR_INVALID_ARG;
  if (misaligned(train) || misaligned(ref) || (reinterpret_cast<uintptr_t>(out) & 7u)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_params prm;
  const double tau = exp(tau_signed), nw = exp(nu_signed);
  prm.inv_h = 1.0 / exp(h_signed);
  prm.eps = eps;
  prm.E = exp(-tau);
  prm.tauE = tau * prm.E;
  prm.V = 1.0 / (nw + 1.0);
  prm.nw = nw;
  const int grid = grid_for(ws, n_rows);
  if (train_ar)
    hipLaunchKernelGGL((dm_ref_kernel<true>), dim3(grid), dim3(BEAR_THREADS), 0, s, train, ref, n_rows, prm, ws->partials);
  else
    hipLaunchKernelGGL((dm_ref_kernel<false>), dim3(grid), dim3(BEAR_THREADS), 0, s, train, ref, n_rows, prm, ws->partials);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, ws->partials, grid, 4, out);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_synth_counts_u32(uint64_t seed, uint64_t row0, uint64_t n_rows, int dense, uint32_t *train,
                          uint32_t *test, uint32_t *ref, void *stream) {
  if (n_rows == 0) return BEAR_OK;
  if (!train && !test && !ref) return BEAR_ERR_INVALID_ARG;
  uint64_t blocks = (n_rows + 255) / 256;
  if (blocks > 0x7fffffffull) return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(synth_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), seed, row0, n_rows, dense, train, test, ref);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_synth_prior_f64(uint64_t seed, uint64_t row0, uint64_t n_rows, double *prior, void *stream) {
  if (n_rows == 0) return BEAR_OK;
  if (!prior) return BEAR_ERR_INVALID_ARG;
  uint64_t blocks = (n_rows + 255) / 256;
  if (blocks > 0x7fffffffull) return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(synth_prior_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), seed, row0, n_rows, prior);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

}  // extern "C"

// kernels_cnn.h -- the convolutional AR function of bear_net fused end to end (BASELINE configs[4]).
//
// Replaces make_ar_func_cnn's ar_func (bear_model/ar_funcs.py:49-99) and its backward pass (grad_tape.gradient,
// bear_model/bear_net.py:193) for 4-letter alphabets, num_filters = 30, kmer_layer1_width = 16 (the values of every
// reference config, models/config_files/bear_cnn_*.cfg:65) and any lag <= 21, filter_width <= lag:
//
//   conv[p][f] = sum_w filters[w][a_{p+w}][f]                (conv1d VALID over a one-hot input = table look-ups)
//   y0 = scale0 * LN_f(conv) + intercept0 ; e0 = elu(y0)     (_normalize_layer over the filter axis, ar_funcs.py:18-19,78)
//   t1[j] = sum_{p,f} e0[p][f] weights1[p][f][j]             (tensordot, ar_funcs.py:79-81)
//   y1 = scale1 * LN_j(t1) + intercept1 ; e1 = elu(y1)
//   z[b] = intercept2[b] + sum_j e1[j] weights2[j][b] ; prior = softmax(z)          (ar_funcs.py:82-83)
//
// The torch formulation materialises [N, P, nf] fp64 activations (1.4 kB per context and layer) and runs ~40 passes
// over them: 2.4 s per 1e7 contexts.  Here a context is one thread: it reads its 8-byte packed k-mer, keeps one
// position's 30 filter responses in registers, takes the filter rows from an LDS table (6 rows per tap: A, C, G, T,
// '[' and a zero row for any other character) and every other weight from SGPRs (uniform scalar loads), and
// writes its 5 prior rows plus the 16 pre-normalisation layer-1 sums t1 (so that the backward kernel does not redo
// the first tensordot).  Backward recomputes the cheap per-position quantities and accumulates the parameter
// gradients per block (LDS), one partial vector per block, fixed-order finalize.
//
// Packed parameter vector (doubles), the reference's parameter order (ar_funcs.py:98-99):
//   filters [fw][5][nf] | intercept0 [P][nf] | weights1 [P][nf][l1] | intercept1 [l1] | weights2 [l1][5] |
//   intercept2 [5] | scale0 [P][nf] | scale1 [l1]
#pragma once
#include "bear_common.h"

#define CNN_NF 30
#define CNN_L1 16
#define CNN_THREADS 256
#define CNN_MAX_LAG 21
#define CNN_LN_EPS 1e-5

struct cnn_dims {
  int lag, fw, P;
  int oF, ob0, oW1, ob1, oW2, ob2, os0, os1, total;   // offsets in doubles
};

static inline cnn_dims cnn_make_dims(int lag, int fw) {
  cnn_dims d;
  d.lag = lag;
  d.fw = fw;
  d.P = lag - fw + 1;
  d.oF = 0;
  d.ob0 = d.oF + fw * 5 * CNN_NF;
  d.oW1 = d.ob0 + d.P * CNN_NF;
  d.ob1 = d.oW1 + d.P * CNN_NF * CNN_L1;
  d.oW2 = d.ob1 + CNN_L1;
  d.ob2 = d.oW2 + CNN_L1 * 5;
  d.os0 = d.ob2 + 5;
  d.os1 = d.os0 + d.P * CNN_NF;
  d.total = d.os1 + CNN_L1;
  return d;
}

// LDS image of the filter bank: 6 letter rows per tap (row 5 = zeros: characters outside the alphabet, core.py:173)
__device__ __forceinline__ void cnn_stage_filters(double *Fs, const double *__restrict__ params, const cnn_dims &D) {
  for (int k = threadIdx.x; k < D.fw * 6 * CNN_NF; k += CNN_THREADS) {
    const int w = k / (6 * CNN_NF), r = k - w * 6 * CNN_NF, a = r / CNN_NF, f = r - a * CNN_NF;
    Fs[k] = a < 5 ? params[D.oF + (w * 5 + a) * CNN_NF + f] : 0.0;
  }
}

// elu(y) and its derivative: y > 0 ? (y, 1) : (exp(y) - 1, exp(y))     (tf.nn.elu, alpha = 1)
__device__ __forceinline__ double cnn_elu(double y, const double *exptab, double &deriv) {
  const double ex = bear_exp_tab(y < 0.0 ? y : 0.0, exptab);
  deriv = y > 0.0 ? 1.0 : ex;
  return y > 0.0 ? y : ex - 1.0;
}

__device__ __forceinline__ double cnn_rsqrt(double v) {
  const double s = sqrt(v);
  return bear_rcp(s);
}

// conv row of position p, then the layer norm over the filter axis: x <- normalised row, returns 1/sqrt(var + eps)
__device__ __forceinline__ double cnn_conv_norm(const double *Fs, unsigned long long code, int p, int fw, double (&x)[CNN_NF]) {
#pragma unroll
  for (int f = 0; f < CNN_NF; ++f) x[f] = 0.0;
  unsigned long long c = code >> (3 * p);
  for (int w = 0; w < fw; ++w) {
    const int a = (int)(c & 7ull);
    const double2 *row = reinterpret_cast<const double2 *>(Fs + (w * 6 + (a < 5 ? a : 5)) * CNN_NF);
    c >>= 3;
#pragma unroll
    for (int f2 = 0; f2 < CNN_NF / 2; ++f2) {
      const double2 v = row[f2];
      x[2 * f2] += v.x;
      x[2 * f2 + 1] += v.y;
    }
  }
  double mu = 0.0;
#pragma unroll
  for (int f = 0; f < CNN_NF; ++f) mu += x[f];
  mu *= 1.0 / CNN_NF;
  double var = 0.0;
#pragma unroll
  for (int f = 0; f < CNN_NF; ++f) {
    x[f] -= mu;
    var = __builtin_fma(x[f], x[f], var);
  }
  const double r = cnn_rsqrt(var * (1.0 / CNN_NF) + CNN_LN_EPS);
#pragma unroll
  for (int f = 0; f < CNN_NF; ++f) x[f] *= r;
  return r;
}

// layer 1 onwards: t1 -> normalised n1, e1 (+ derivative), 1/sigma
__device__ __forceinline__ double cnn_layer1(const double (&t1)[CNN_L1], const double *__restrict__ params, const cnn_dims &D,
                                             const double *exptab, double (&n1)[CNN_L1], double (&e1)[CNN_L1],
                                             double (&d1)[CNN_L1]) {
  double mu = 0.0;
#pragma unroll
  for (int j = 0; j < CNN_L1; ++j) mu += t1[j];
  mu *= 1.0 / CNN_L1;
  double var = 0.0;
#pragma unroll
  for (int j = 0; j < CNN_L1; ++j) {
    n1[j] = t1[j] - mu;
    var = __builtin_fma(n1[j], n1[j], var);
  }
  const double r = cnn_rsqrt(var * (1.0 / CNN_L1) + CNN_LN_EPS);
#pragma unroll
  for (int j = 0; j < CNN_L1; ++j) {
    n1[j] *= r;
    e1[j] = cnn_elu(__builtin_fma(params[D.os1 + j], n1[j], params[D.ob1 + j]), exptab, d1[j]);
  }
  return r;
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(CNN_THREADS) void cnn_forward_kernel(const unsigned long long *__restrict__ codes, uint64_t n_rows,
                                                                   cnn_dims D, const double *__restrict__ params,
                                                                   double *__restrict__ prior, double *__restrict__ t1_save) {
  extern __shared__ __attribute__((aligned(16))) double cnn_lds[];
  double *exptab = cnn_lds;                       // [128]
  double *Fs = cnn_lds + BEAR_EXPTAB_N;           // [fw][6][nf]
  if (threadIdx.x < BEAR_EXPTAB_N) exptab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / BEAR_EXPTAB_N));
  cnn_stage_filters(Fs, params, D);
  __syncthreads();
  for (uint64_t base = (uint64_t)blockIdx.x * CNN_THREADS; base < n_rows; base += (uint64_t)gridDim.x * CNN_THREADS) {
    const uint64_t i = base + threadIdx.x;
    const bool live = i < n_rows;
    const unsigned long long code = live ? codes[i] : ~0ull;
    double t1[CNN_L1];
#pragma unroll
    for (int j = 0; j < CNN_L1; ++j) t1[j] = 0.0;
    for (int p = 0; p < D.P; ++p) {
      double x[CNN_NF];
      cnn_conv_norm(Fs, code, p, D.fw, x);
      const double *__restrict__ s0 = params + D.os0 + p * CNN_NF, *__restrict__ b0 = params + D.ob0 + p * CNN_NF;
      const double *__restrict__ W1 = params + D.oW1 + p * CNN_NF * CNN_L1;
#pragma unroll
      for (int f = 0; f < CNN_NF; ++f) {
        double dv;
        const double e = cnn_elu(__builtin_fma(s0[f], x[f], b0[f]), exptab, dv);
#pragma unroll
        for (int j = 0; j < CNN_L1; ++j) t1[j] = __builtin_fma(e, W1[f * CNN_L1 + j], t1[j]);
      }
    }
    double n1[CNN_L1], e1[CNN_L1], d1[CNN_L1];
    cnn_layer1(t1, params, D, exptab, n1, e1, d1);
    double z[5], m = -INFINITY;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      double s = params[D.ob2 + b];
#pragma unroll
      for (int j = 0; j < CNN_L1; ++j) s = __builtin_fma(e1[j], params[D.oW2 + j * 5 + b], s);
      z[b] = s;
      m = s > m ? s : m;
    }
    double tot = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      z[b] = bear_exp_tab(z[b] - m, exptab);
      tot += z[b];
    }
    const double rt = bear_rcp(tot);
    if (live) {
#pragma unroll
      for (int b = 0; b < 5; ++b) prior[i * 5 + b] = z[b] * rt;
      if (t1_save) {
        double2 *o = reinterpret_cast<double2 *>(t1_save + i * CNN_L1);
#pragma unroll
        for (int j = 0; j < CNN_L1 / 2; ++j) o[j] = make_double2(t1[2 * j], t1[2 * j + 1]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// grad_prior [n,5] = d loss / d prior rows (what the DM kernel returns, already scaled by the caller or not: linear).
// Accumulates d loss / d params: per block in LDS (fp64 LDS atomics), one partial vector per block.
__device__ __forceinline__ void cnn_lds_add(double *addr, double v) { atomicAdd(addr, v); }

__global__ __launch_bounds__(CNN_THREADS) void cnn_backward_kernel(const unsigned long long *__restrict__ codes, uint64_t n_rows,
                                                                    cnn_dims D, const double *__restrict__ params,
                                                                    const double *__restrict__ t1_save,
                                                                    const double *__restrict__ prior,
                                                                    const double *__restrict__ grad_prior,
                                                                    double *__restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) double cnn_lds[];
  double *exptab = cnn_lds;
  double *Fs = cnn_lds + BEAR_EXPTAB_N;
  double *G = Fs + D.fw * 6 * CNN_NF;             // [total] block gradient accumulators, parameter layout
  if (threadIdx.x < BEAR_EXPTAB_N) exptab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / BEAR_EXPTAB_N));
  cnn_stage_filters(Fs, params, D);
  for (int k = threadIdx.x; k < D.total; k += CNN_THREADS) G[k] = 0.0;
  __syncthreads();
  for (uint64_t base = (uint64_t)blockIdx.x * CNN_THREADS; base < n_rows; base += (uint64_t)gridDim.x * CNN_THREADS) {
    const uint64_t i = base + threadIdx.x;
    if (i >= n_rows) continue;
    const unsigned long long code = codes[i];
    double t1[CNN_L1], n1[CNN_L1], e1[CNN_L1], d1[CNN_L1];
    {
      const double2 *src = reinterpret_cast<const double2 *>(t1_save + i * CNN_L1);
#pragma unroll
      for (int j = 0; j < CNN_L1 / 2; ++j) {
        const double2 v = src[j];
        t1[2 * j] = v.x;
        t1[2 * j + 1] = v.y;
      }
    }
    const double r1 = cnn_layer1(t1, params, D, exptab, n1, e1, d1);
    // softmax backward
    double dz[5], sg = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      dz[b] = prior[i * 5 + b];
      sg = __builtin_fma(dz[b], grad_prior[i * 5 + b], sg);
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      dz[b] *= grad_prior[i * 5 + b] - sg;
      cnn_lds_add(G + D.ob2 + b, dz[b]);
    }
    // layer 2 / layer-1 norm backward -> dt1 (kept in t1)
    double ma = 0.0, mb = 0.0;
#pragma unroll
    for (int j = 0; j < CNN_L1; ++j) {
      double de = 0.0;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        de = __builtin_fma(params[D.oW2 + j * 5 + b], dz[b], de);
        cnn_lds_add(G + D.oW2 + j * 5 + b, e1[j] * dz[b]);
      }
      const double dy = de * d1[j];
      cnn_lds_add(G + D.os1 + j, dy * n1[j]);
      cnn_lds_add(G + D.ob1 + j, dy);
      const double dn = dy * params[D.os1 + j];
      t1[j] = dn;
      ma += dn;
      mb = __builtin_fma(dn, n1[j], mb);
    }
    ma *= 1.0 / CNN_L1;
    mb *= 1.0 / CNN_L1;
#pragma unroll
    for (int j = 0; j < CNN_L1; ++j) t1[j] = r1 * (t1[j] - ma - n1[j] * mb);
    // positions
    for (int p = 0; p < D.P; ++p) {
      double x[CNN_NF], dn0[CNN_NF];
      const double r0 = cnn_conv_norm(Fs, code, p, D.fw, x);
      const double *__restrict__ s0 = params + D.os0 + p * CNN_NF, *__restrict__ b0 = params + D.ob0 + p * CNN_NF;
      const double *__restrict__ W1 = params + D.oW1 + p * CNN_NF * CNN_L1;
      double *gW1 = G + D.oW1 + p * CNN_NF * CNN_L1, *gs0 = G + D.os0 + p * CNN_NF, *gb0 = G + D.ob0 + p * CNN_NF;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int f = 0; f < CNN_NF; ++f) {
        double dv;
        const double e = cnn_elu(__builtin_fma(s0[f], x[f], b0[f]), exptab, dv);
        double de = 0.0;
#pragma unroll
        for (int j = 0; j < CNN_L1; ++j) {
          de = __builtin_fma(W1[f * CNN_L1 + j], t1[j], de);
          cnn_lds_add(gW1 + f * CNN_L1 + j, e * t1[j]);
        }
        const double dy = de * dv;
        cnn_lds_add(gs0 + f, dy * x[f]);
        cnn_lds_add(gb0 + f, dy);
        const double dn = dy * s0[f];
        dn0[f] = dn;
        a0 += dn;
        a1 = __builtin_fma(dn, x[f], a1);
      }
      a0 *= 1.0 / CNN_NF;
      a1 *= 1.0 / CNN_NF;
#pragma unroll
      for (int f = 0; f < CNN_NF; ++f) dn0[f] = r0 * (dn0[f] - a0 - x[f] * a1);
      unsigned long long c = code >> (3 * p);
      for (int w = 0; w < D.fw; ++w) {
        const int a = (int)(c & 7ull);
        c >>= 3;
        if (a < 5) {
          double *gF = G + D.oF + (w * 5 + a) * CNN_NF;
#pragma unroll
          for (int f = 0; f < CNN_NF; ++f) cnn_lds_add(gF + f, dn0[f]);
        }
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < D.total; k += CNN_THREADS) partials[(size_t)blockIdx.x * D.total + k] = G[k];
}

// fixed-order sum of the block partial vectors: one wave per parameter (lane-strided partial sums, then the shuffle tree)
__global__ __launch_bounds__(256) void cnn_finalize_kernel(const double *__restrict__ partials, int n_blocks, int total,
                                                           double *__restrict__ out) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= total) return;
  double s = 0.0;
  for (int b = lane; b < n_blocks; b += 64) s += partials[(size_t)b * total + k];
  s = bear_wave_sum(s);
  if (lane == 0) out[k] = s;
}

// kernels_linear.h -- the whole bear_net training step for the linear AR function, fused on the plan.
//
//   f_i = softmax(sum_l mat[l, kmer_i[l], :])                      (ar_funcs.py:41-45)
//   sum LL, d sum LL / d h_signed                                   (bear_net.py:177-191, core.py:73-74)
//   d sum LL / d mat[l, a, b]                                       (bear_net.py:193 through the softmax)
//
// Contexts arrive as packed k-mers (3 bits per letter: 0..A-1, A = start symbol, 5 = unknown letter = all-zero
// one-hot row, core.py:173), 8 bytes per context instead of the 40-byte prior row, and nothing is written per
// context.  Per tile of the plan:
//   A  one thread per context: logits from PAIR tables T[g][a_2g, a_2g+1][b] = mat[2g][a][b] + mat[2g+1][a'][b]
//      (7 LDS rows instead of 13 for lag 13), softmax, row into LDS;
//   B  ticketed item units as in dm_prior_plan_kernel: D, P per item; with q = dLL/df at the item's cell and
//      w = f q the softmax backward is  g_logit[b] = f_b (q_b - s),  s = sum over the context's items of w.
//      The common base -u P(A,n) of all five cells drops out because the softmax row sums to one, so contexts
//      only need s (LDS fp64 atomics per item) and the item adds +w to the gradient pair table at its own cell;
//   C  one thread per context with s != 0: -f_b s into the gradient pair tables (LDS fp64 atomics).
// The context terms -D(A, n) come from the plan's histogram (A = u + 5 eps: softmax rows are normalised).
// After the last tile the pair tables fold into d/d mat partials; a finalize kernel sums the blocks in fixed order.
// Measured at 1e8 contexts, lag 13 (3.9 ms): pass C 1.6 ms and the item scatter 0.6 ms run at the LDS fp64-atomic
// rate (~2.5 lane-atomics per clock per CU, scripts/dev/lds_atomic_bench.hip), the softmax exponentials 0.2 ms.
// Note: LDS floating-point atomics make the summation order inside a block run-dependent (last-bit jitter in
// grad_mat); the ELBO and d/dh sums keep the fixed-order reduction of the other kernels.
#pragma once
#include "kernels_plan.h"

#define LIN_MAX_LAG 21
#define LIN_MAX_GROUPS ((LIN_MAX_LAG + 1) / 2)
#define LIN_COMBOS 36
#define LIN_GSTRIDE (LIN_COMBOS * 5)
#define LIN_MAX_GRAD (LIN_MAX_LAG * 25)

struct lin_buf {
  __attribute__((aligned(16))) unsigned long long codes[PLN_RMAX + 2];
  __attribute__((aligned(16))) unsigned char blk[PLN_BLOCK_MAX];
};
struct pln_lds_lin {
  double pri[PLN_RMAX * 5 + 2];  // [PLN_SENTINEL] = 1.0
  double srow[PLN_RMAX + 2];     // [PLN_RMAX] = sink of the sentinel lane
  lin_buf buf[2];
  double T[LIN_MAX_GROUPS * LIN_GSTRIDE];
  double GT[LIN_MAX_GROUPS * LIN_GSTRIDE];
  double2 logtab[BEAR_LOGTAB_N];
  double tabD[SRT_NKEY];
  double tabP[SRT_NKEY];
  double exptab[BEAR_EXPTAB_N];
  uint32_t ticket[2];
};
static_assert(sizeof(pln_lds_lin) <= 160 * 1024, "linear-head kernel: LDS budget");

// int8 codes [n, lag] (core.encode_kmers: 0..A letters / start symbol, anything else unknown) -> packed words
__global__ void pack_kmers_kernel(const int8_t *__restrict__ codes, uint64_t n, int lag, unsigned long long *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long w = 0;
  for (int l = 0; l < LIN_MAX_LAG + 1; ++l) {
    unsigned long long v = 5;
    if (l < lag) {
      const int c = codes[i * lag + l];
      v = (c >= 0 && c <= 4) ? (unsigned long long)c : 5ull;
    }
    w |= v << (3 * l);
  }
  out[i] = w;
}

// ASCII k-mers as they sit in the count file -> int8 letter codes (core.tf_one_hot's alphabet order, core.py:146-153:
// 0..3 letters, 4 = start symbol '[', -1 = anything else) -- the host LUT of a 1e9-row table took longer than an epoch.
__global__ void encode_kmers_kernel(const uint8_t *__restrict__ ascii, uint64_t n_bytes, int rna, int8_t *__restrict__ codes) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t ch = ascii[i];
    int8_t c = -1;
    if (ch == 'A') c = 0;
    else if (ch == 'C') c = 1;
    else if (ch == 'G') c = 2;
    else if (ch == (rna ? 'U' : 'T')) c = 3;
    else if (ch == '[') c = 4;
    codes[i] = c;
  }
}

__device__ __forceinline__ uint32_t lin_combo(unsigned long long code, int g) {
  const uint32_t field = (uint32_t)(code >> (6 * g)) & 63u;
  return (field & 7u) * 6u + (field >> 3);
}

// softmax row of one context from the pair tables
__device__ __forceinline__ void lin_row(const double *T, const double *exptab, unsigned long long code, int ng, double (&f)[5]) {
  double z[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int g = 0; g < ng; ++g) {
    const double *t = T + g * LIN_GSTRIDE + lin_combo(code, g) * 5u;
#pragma unroll
    for (int b = 0; b < 5; ++b) z[b] += t[b];
  }
  double m = z[0];
#pragma unroll
  for (int b = 1; b < 5; ++b) m = z[b] > m ? z[b] : m;
  double s = 0.0;
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    f[b] = bear_exp_tab(z[b] - m, exptab);
    s += f[b];
  }
  const double r = bear_rcp(s);
#pragma unroll
  for (int b = 0; b < 5; ++b) f[b] *= r;
}

template <bool AR>
__global__ __launch_bounds__(PLN_THREADS, PLN_WAVES / 4) void dm_linear_plan_kernel(
    const unsigned long long *__restrict__ kmer_code, const double *__restrict__ mat, int lag, bear_params prm_arg, pln_view pv,
    const double2 *__restrict__ logtab_g, double *__restrict__ partials, double *__restrict__ grad_partials,
    const bear_params *__restrict__ prm_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_lin &S = *reinterpret_cast<pln_lds_lin *>(srt_smem);
  const bear_params prm = prm_dev ? *prm_dev : prm_arg;   // device-resident parameters for HIP-graph replay (bear_net_linear_train_step_f64)
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  const int ng = (lag + 1) >> 1;
  double acc[2] = {0.0, 0.0};

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < SRT_NKEY) {
    const bear_dp o = srt_general_fast(u + eps5, (double)(tid + 1), logtab_g);
    S.tabD[tid] = o.D;
    S.tabP[tid] = o.P;
  }
  if (tid == 0) {
    S.pri[PLN_SENTINEL] = 1.0;
    S.ticket[0] = 0;
    S.ticket[1] = 0;
  }
  if (tid < BEAR_EXPTAB_N) S.exptab[tid] = exp2((double)tid * (1.0 / BEAR_EXPTAB_N));
  // pair tables: T[g][a * 6 + a'][b] = mat[2g][a][b] (a < 5) + mat[2g+1][a'][b] (a' < 5, position inside the lag)
  for (int k = tid; k < ng * LIN_GSTRIDE; k += PLN_THREADS) {
    const int g = k / LIN_GSTRIDE, r = k - g * LIN_GSTRIDE, combo = r / 5, b = r - combo * 5;
    const int a0 = combo / 6, a1 = combo - a0 * 6, l0 = 2 * g, l1 = 2 * g + 1;
    double v = 0.0;
    if (a0 < 5) v += mat[(l0 * 5 + a0) * 5 + b];
    if (a1 < 5 && l1 < lag) v += mat[(l1 * 5 + a1) * 5 + b];
    S.T[k] = v;
    S.GT[k] = 0.0;
  }
  __syncthreads();

  auto stage = [&](const pln_tile &ti, uint32_t b) {
    const uint32_t rows = ti.rows_items >> 16;
    if (rows == 0) return;
    const uint32_t cbytes = rows * 8u;
    pln_dma(S.buf[b].codes, kmer_code + ti.row0, cbytes & ~15u, wave, lane, 0);
    if (cbytes & 15u) {  // odd row count: trailing word through the scalar path (see dm_prior_plan_kernel)
      const __attribute__((address_space(4))) unsigned long long *tail =
          (const __attribute__((address_space(4))) unsigned long long *)(uintptr_t)(kmer_code + ti.row0 + rows - 1);
      const unsigned long long v = *tail;
      if (tid == 0) S.buf[b].codes[rows - 1] = v;
    }
    pln_dma(S.buf[b].blk, pv.stream + (size_t)ti.off16 * 16, ti.blk16 * 16u, wave, lane, (cbytes + 1023u) >> 10);
  };
  // +w (item's own cell) or -f_b s (all cells of a context) into the gradient pair tables
  // Lanes walk the groups in rotated order (lane i starts at group i mod ng): at any moment the 64 atomics of
  // a wave spread over all ng tables instead of colliding inside one.
  const int g_rot = (int)(lane % (uint32_t)ng);
  // Only letters b < 4 are accumulated: the softmax gradient of a context sums to zero over b, so the last
  // column is minus the sum of the others (restored in the fold below).
  auto scatter1 = [&](unsigned long long code, uint32_t b, double w) {
    if (b == 4u) return;
    int g = g_rot;
    for (int k = 0; k < ng; ++k) {
      atomicAdd(&S.GT[g * LIN_GSTRIDE + lin_combo(code, g) * 5u + b], w);
      g = g + 1 == ng ? 0 : g + 1;
    }
  };

  const uint64_t G = gridDim.x;
  pln_tile cur = pln_load_tile(pv, blockIdx.x), nxt = pln_load_tile(pv, blockIdx.x + G);
  stage(cur, 0);
  uint32_t slot = 0;
  for (uint64_t t = blockIdx.x; t < pv.n_tiles; t += G) {
    srt_wait_dma();
    srt_sync();  // current tile landed; previous tile fully consumed
    stage(nxt, slot ^ 1u);
    const lin_buf &B = S.buf[slot];
    const uint32_t rows = cur.rows_items >> 16, n_light = cur.rows_items & 0xffffu;
    const uint32_t hc = cur.hc_hr >> 16, hr = cur.hc_hr & 0xffffu;
    const pln_layout L = pln_block_layout(rows, n_light, hc, hr);
    const uint16_t *E = reinterpret_cast<const uint16_t *>(B.blk);
    const uint16_t *items = reinterpret_cast<const uint16_t *>(B.blk + L.items);
    if (tid == 0) S.ticket[slot ^ 1u] = 0;
    // ---- A: softmax rows of the tile
    for (uint32_t row = tid; row < rows; row += PLN_THREADS) {
      double f[5];
      lin_row(S.T, S.exptab, B.codes[row], ng, f);
#pragma unroll
      for (int b = 0; b < 5; ++b) S.pri[row * 5 + b] = f[b];
      S.srow[row] = 0.0;
    }
    srt_sync();
    // ---- B: items (tickets, dearest first)
    auto item = [&](uint32_t off, double D, double P, double x, double cnt) {
      // D, P of the item (AR: unused), x its concentration; accumulates ELBO / d/dh and the softmax backward
      const double fb = S.pri[off];
      double q;
      if (AR) {
        const double pp = fb + eps;
        acc[0] = __builtin_fma(cnt, bear_log_tab(pp, S.logtab), acc[0]);
        q = cnt * bear_rcp(pp);
      } else {
        acc[0] += D;
        acc[1] = __builtin_fma(eps - x, P, acc[1]);
        q = u * P;
      }
      if (cnt != 0.0) {
        const uint32_t row = off / 5u, b = off - row * 5u;
        const double w = fb * q;
        atomicAdd(&S.srow[row], w);
        scatter1(B.codes[row], b, w);
      }
    };
    const uint32_t n_hcu = (hc + 63u) >> 6, n_hru = AR ? 0u : (hr + 63u) >> 6, n_units = (n_light + 63u) >> 6;
    const uint32_t n_work = n_hcu + n_hru + n_units;
    for (uint32_t w = pln_ticket(&S.ticket[slot], lane); w < n_work; w = pln_ticket(&S.ticket[slot], lane)) {
      if (w < n_hcu) {
        const uint32_t i = w * 64u + lane;
        if (i < hc) {
          const uint32_t off = reinterpret_cast<const uint16_t *>(B.blk + L.hoff)[i];
          const double cnt = (double)reinterpret_cast<const uint32_t *>(B.blk + L.hcnt)[i];
          const double x = __builtin_fma(S.pri[off], u, eps);
          bear_dp o = {0.0, 0.0};
          if (!AR) o = srt_general_fast(x, cnt, S.logtab);
          item(off, o.D, o.P, x, cnt);
        }
        continue;
      }
      if (w < n_hcu + n_hru) {  // contexts with a large total: context terms only (no gradient: the base cancels)
        const uint32_t i = (w - n_hcu) * 64u + lane;
        if (i < hr) {
          const bear_dp o = srt_general_fast(u + eps5, reinterpret_cast<const double *>(B.blk + L.hn)[i], S.logtab);
          acc[0] -= o.D;
          acc[1] = __builtin_fma(u, o.P, acc[1]);
        }
        continue;
      }
      const uint32_t un = n_work - 1u - w;
      uint32_t cmin, cmax;
      const uint32_t ci[1] = {pln_unit_counts(E, n_light, un, lane, &cmin, &cmax)};
      const uint32_t off = items[un * 64u + lane];
      const double x[1] = {__builtin_fma(S.pri[off], u, eps)};
      bear_dp o[1] = {{0.0, 0.0}};
      if (!AR) srt_light<1>(x, ci, cmin, cmax, S.logtab, o);
      item(off, o[0].D, o[0].P, x[0], (double)ci[0]);
    }
    srt_sync();
    // ---- C: contexts that own items: -f_b s into the gradient tables
    for (uint32_t row = tid; row < rows; row += PLN_THREADS) {
      const double s = S.srow[row];
      if (s != 0.0) {
        const unsigned long long code = B.codes[row];
        double fs[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) fs[b] = -S.pri[row * 5 + b] * s;
        int g = g_rot;
        for (int k = 0; k < ng; ++k) {
          double *gt = &S.GT[g * LIN_GSTRIDE + lin_combo(code, g) * 5u];
#pragma unroll
          for (int b = 0; b < 4; ++b) atomicAdd(&gt[b], fs[b]);
          g = g + 1 == ng ? 0 : g + 1;
        }
      }
    }
    cur = nxt;
    nxt = pln_load_tile(pv, t + 2 * G);
    slot ^= 1u;
  }
  srt_wait_dma();
  __syncthreads();
  // ---- items / contexts that overflowed to the plan's global lists (very dense tiles): self-contained
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    const uint64_t row = h.off / 5u;
    const uint32_t b = (uint32_t)(h.off - row * 5u);
    const unsigned long long code = kmer_code[row];
    double f[5];
    lin_row(S.T, S.exptab, code, ng, f);
    double q;
    if (AR) {
      const double pp = f[b] + eps;
      acc[0] = __builtin_fma((double)h.c, bear_log_tab(pp, S.logtab), acc[0]);
      q = (double)h.c * bear_rcp(pp);
    } else {
      const double x = __builtin_fma(f[b], u, eps);
      const bear_dp o = srt_general_fast(x, (double)h.c, S.logtab);
      acc[0] += o.D;
      acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
      q = u * o.P;
    }
    const double w = f[b] * q;
    for (int g = 0; g < ng; ++g) {
      double *gt = &S.GT[g * LIN_GSTRIDE + lin_combo(code, g) * 5u];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) atomicAdd(&gt[bb], (bb == (int)b ? w : 0.0) - f[bb] * w);
    }
  }
  for (uint64_t i = gtid; !AR && i < pv.n_heavy_row; i += gsz) {
    const bear_dp o = srt_general_fast(u + eps5, pv.heavy_row[i].n, S.logtab);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(u, o.P, acc[1]);
  }
  if (!AR && blockIdx.x == 0 && tid < SRT_CL) {  // context terms of the small totals: the plan's histogram
    const double m = (double)pv.hist[tid];
    acc[0] -= m * S.tabD[tid];
    acc[1] = __builtin_fma(u * m, S.tabP[tid], acc[1]);
  }
  __syncthreads();
  // ---- fold the pair tables into d/d mat[l][a][b]
  for (int k = tid; k < lag * 25; k += PLN_THREADS) {
    const int l = k / 25, r = k - l * 25, a = r / 5, b = r - a * 5, g = l >> 1;
    double s = 0.0;
    for (int p = 0; p < 6; ++p) {
      const int combo = (l & 1) ? p * 6 + a : a * 6 + p;
      const double *gt = &S.GT[g * LIN_GSTRIDE + combo * 5];
      s += b < 4 ? gt[b] : -((gt[0] + gt[1]) + (gt[2] + gt[3]));
    }
    grad_partials[(size_t)blockIdx.x * LIN_MAX_GRAD + k] = s;
  }
  block_store_partials<2>(acc, partials);
}

// fixed-order sum of the per-block d/d mat partials: one wave per entry
__global__ __launch_bounds__(256) void linear_finalize_kernel(const double *__restrict__ grad_partials, int n_blocks, int n_grad,
                                                              double *__restrict__ grad_mat) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_grad) return;
  double s = 0.0;
  for (int b = lane; b < n_blocks; b += 64) s += grad_partials[(size_t)b * LIN_MAX_GRAD + k];
  s = bear_wave_sum(s);
  if (lane == 0) grad_mat[k] = s;
}


// ---- the bear_net / linear optimizer step on the device (HIP-graph replay) ---------------------------------------
// theta = {h_signed, mat[lag,5,5]} contiguous.  net_params_kernel derives 1/h; adam_vec_kernel is tf.keras Adam on the whole
// vector with gradients grad[k] * scale (k = 0: d/dh from out[1], skipped in AR mode; k >= 1: d/d mat).
__global__ void net_params_kernel(const double *__restrict__ theta, double eps, bear_params *__restrict__ prm) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  bear_params p;
  p.inv_h = 1.0 / exp(theta[0]);
  p.eps = eps;
  p.E = p.tauE = p.tau = p.V = p.nw = 0.0;
  *prm = p;
}

__global__ __launch_bounds__(256) void adam_vec_kernel(double *__restrict__ theta, const double *__restrict__ out2,
                                                       const double *__restrict__ grad_rest, int n_rest, double *__restrict__ m,
                                                       double *__restrict__ v, const double *__restrict__ t_state, double lr,
                                                       double scale, int train_ar, double *__restrict__ loss_buf,
                                                       unsigned long long loss_cap) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k > n_rest) return;
  const double t = t_state[0] + 1.0;
  const double b1 = 0.9, b2 = 0.999, aeps = 1e-7;
  const double lr_t = lr * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t));
  if (!(train_ar && k == 0)) {
    const double g = scale * (k == 0 ? out2[1] : grad_rest[k - 1]);
    const double mk = b1 * m[k] + (1.0 - b1) * g, vk = b2 * v[k] + (1.0 - b2) * g * g;
    m[k] = mk;
    v[k] = vk;
    theta[k] -= lr_t * mk / (sqrt(vk) + aeps);
  }
  if (k == 0) {
    const unsigned long long step = (unsigned long long)t_state[0];
    if (loss_buf && step < loss_cap) loss_buf[step] = -scale * out2[0];
  }
}
__global__ void adam_tick_kernel(double *__restrict__ t_state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t_state[0] += 1.0;
}

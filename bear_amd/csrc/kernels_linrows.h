// kernels_linrows.h -- the linear AR function as prior ROWS, forward and backward (bear_linear_forward_f64 / bear_linear_backward_f64).
//
//   forward:   prior_i = softmax(sum_l mat[l, kmer_i[l], :])                         (ar_funcs.py:41-45)
//   backward:  d L / d mat[l, a, :] = sum over contexts with letter a at position l of  f (q - <f, q>),  q = d L / d prior_i
//
// bear_net.train never forms these rows (kernels_linear.h fuses the whole step).  Everybody else who calls the linear AR function
// on contexts does: evaluation / h_scan (bear_net.py:387-531: the rows feed the evaluation kernel), bear_ref.train with the linear
// net function (bear_ref.py:63-68: the rows are mixed with the reference prior before the DM step), get_var_probs.  As torch ops
// (an embedding-bag over [n, lag] int64 indices and its scatter-add backward) that was 121 ms per 1e7 contexts forward + backward.
// Both kernels share the letter-group tables of the fused step (pairs of letters, one triple; lin_build_tables, lin_row) and its
// gradient scatter (lin_scatter_grad: wave / row-of-16 / quad sums for the groups consecutive contexts share -- everything but
// the last letters when the rows are in k-mer order; correct in any order).
//   forward : 8 B read + 40 B written per context, one context per lane; a wave's 64 rows leave through LDS as 16-byte stores.
//   backward: 8 + 40 + 40 B read per context; block partials of d/d mat, summed in a fixed order by the last block to finish.
#pragma once
#include "kernels_linear.h"

#define LNR_THREADS 1024
#define LNR_WAVES (LNR_THREADS / 64)

struct lnr_lds_fwd {
  __attribute__((aligned(32))) double T[LIN_TAB_DOUBLES];
  double exptab[BEAR_EXPTAB_N];
  __attribute__((aligned(16))) double rows[LNR_WAVES][64 * 5];   // a wave's rows on their way out
  unsigned long long t_max;
};
static_assert(sizeof(lnr_lds_fwd) <= 64 * 1024, "linear rows forward: static LDS");

__global__ __launch_bounds__(LNR_THREADS) void linear_rows_forward_kernel(const unsigned long long *__restrict__ kmer_code, uint64_t n,
                                                                          const double *__restrict__ mat, int lag,
                                                                          double *__restrict__ prior) {
  __shared__ lnr_lds_fwd S;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = srt_uniform(tid >> 6);
  const lin_geom G = lin_make_geom(lag);
  const int ng = G.ng;
  if (tid < BEAR_EXPTAB_N) S.exptab[tid] = exp2((double)tid * (1.0 / BEAR_EXPTAB_N));
  const bool exp_tables = lin_build_tables(S.T, &S.t_max, mat, G, (int)tid, LNR_THREADS);
  double *R = S.rows[wave];
  const uint64_t n_chunks = (n + 63u) >> 6;
  auto run = [&](auto exp_tag) {
    constexpr bool EXP = decltype(exp_tag)::value;
    for (uint64_t c = (uint64_t)blockIdx.x * LNR_WAVES + wave; c < n_chunks; c += (uint64_t)gridDim.x * LNR_WAVES) {
      const uint64_t i0 = c << 6;
      const uint32_t valid = n - i0 < 64u ? (uint32_t)(n - i0) : 64u;                // rows of this chunk (wave-uniform)
      const unsigned long long cv = lin_index_word(kmer_code[i0 + (lane < valid ? lane : valid - 1u)], G);
      double f[5];
      LIN_FOR_NG(ng, (lin_row<NG, EXP>(S.T, S.exptab, cv, f)))
      bear_wave_store_rows5(R, f, prior, i0, valid, lane);
    }
  };
  if (exp_tables) run(std::true_type{});
  else run(std::false_type{});
}

struct lnr_lds_bwd {
  double GT[LIN_TAB_DOUBLES];            // gradient tables, letter-major (lin_scatter_grad); the last block's scratch afterwards
};
static_assert(LIN_TAB_DOUBLES >= 3 * LIN_MAX_GRAD, "lin_sum_block_partials scratch inside the gradient tables");

__global__ __launch_bounds__(LNR_THREADS) void linear_rows_backward_kernel(const unsigned long long *__restrict__ kmer_code, uint64_t n,
                                                                           int lag, const double *__restrict__ prior,
                                                                           const double *__restrict__ grad_prior,
                                                                           double *__restrict__ grad_partials, const bear_arrival arrive,
                                                                           double *__restrict__ grad_mat) {
  __shared__ lnr_lds_bwd S;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = srt_uniform(tid >> 6);
  const lin_geom G = lin_make_geom(lag);
  const int ng = G.ng;
  for (int k = tid; k < LIN_TAB_DOUBLES; k += LNR_THREADS) S.GT[k] = 0.0;
  __syncthreads();
  double unused[2] = {0.0, 0.0};
  const uint64_t n_chunks = (n + 63u) >> 6;
  for (uint64_t c = (uint64_t)blockIdx.x * LNR_WAVES + wave; c < n_chunks; c += (uint64_t)gridDim.x * LNR_WAVES) {
    const uint64_t i0 = c << 6;
    const uint32_t valid = n - i0 < 64u ? (uint32_t)(n - i0) : 64u;
    const bool in = lane < valid;
    // lanes beyond the end repeat the last context (they add nothing and never break a run)
    const uint64_t i = i0 + (in ? lane : valid - 1u);
    const unsigned long long cv = lin_index_word(kmer_code[i], G);
    double f[5], q[5], s = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      f[b] = prior[i * 5u + b];
      q[b] = grad_prior[i * 5u + b];
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) s = __builtin_fma(f[b], q[b], s);
    double g[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) g[b] = in ? f[b] * (q[b] - s) : 0.0;      // softmax backward: d L / d logit_b
    const bool nz = (g[0] != 0.0) | (g[1] != 0.0) | (g[2] != 0.0) | (g[3] != 0.0);
    if (__builtin_amdgcn_ballot_w64(nz) == 0ull) continue;
    LIN_FOR_NG(ng, (lin_scatter_grad<NG>(S.GT, cv, g, nz, lane, unused)))
  }
  __syncthreads();
  lin_fold_tables(S.GT, G, (int)tid, LNR_THREADS, grad_partials + (size_t)blockIdx.x * LIN_MAX_GRAD);
  if (!bear_arrive_last(arrive)) return;
  lin_sum_block_partials(grad_partials, lag * 25, S.GT, (int)tid, LNR_THREADS, grad_mat);
  if (tid == 0) bear_arrive_reset(arrive);
}

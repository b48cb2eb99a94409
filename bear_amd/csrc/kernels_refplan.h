// kernels_refplan.h -- bear_ref with the stop net function on a plan that also knows the REFERENCE column
// (bear_model/bear_ref.py:207-259; prior from bear_ref.py:30-33, 63-68, 332-337).
//
// The streaming mode-R kernel (kernels_plan.h: dm_ref_plan_kernel) reads the reference rows every step because its plan was
// built from the training counts alone.  But the reference column of a resident count table is as constant as the training
// column, and the concentration of a column item depends on the reference row only through (r_b, R = sum_b r_b):
//     alpha_b = (1/4 + e^-tau ((r_b + eps) / (R + 4 eps) - 1/4)) V / h + eps .
// A context WITHOUT reference counts (92 % of the k=13 synthetic table; most k-mers of a read set that a reference genome does
// not contain) has the same alpha_0 in all four letters, so all of its items collapse into a histogram over the count c --
// 24 numbers for the whole table, built once.  What is left per step is the list of items whose context does have reference
// counts: 16-byte records {c, r_b, R}, sorted by c so that 64 consecutive records are a wave-uniform unit.  The step kernel is a
// plain grid-stride stream over that list (no LDS tiles, no tickets): ~1.6 B per context on the synthetic table instead of 24.
// The context terms and the stop column come from the plan's histograms as in dm_ref_plan_kernel.
#pragma once
#include "kernels_plan.h"

#define RPL_NKEY 32          // buckets of the item sort: min(c, 25) - 1 (0 .. 24); [25..31] unused
#define RPL_HEAVY_KEY 24

struct rpl_item {
  uint32_t c, rb;   // training count of the item (letter b < 4) and the reference count of the same letter
  double R;         // sum of the context's four reference counts (exact: < 2^34)
};
static_assert(sizeof(rpl_item) == 16, "one 16-byte lane load per record");

struct rpl_view {
  const rpl_item *items;               // sorted by min(c, 25)
  uint64_t n_items;
  const unsigned long long *hist0;     // [c - 1] = number of items with count c <= 24 in contexts without reference counts
  const uint32_t *heavy0;              // counts c > 24 of such items
  uint64_t n_heavy0;
  const double *sum0;                  // [1] sum of c over all such items (multinomial mode)
};

// Pass 1 (fill == 0): bucket sizes of the reference items, histogram / count / sum of the others.  Pass 2 (fill != 0): scatter.
__global__ __launch_bounds__(256) void rpl_build_kernel(const uint32_t *__restrict__ train, const uint32_t *__restrict__ ref, uint64_t n_rows,
                                                        int fill, unsigned long long *__restrict__ bucket,   // [RPL_NKEY] sizes / cursors
                                                        unsigned long long *__restrict__ hist0, unsigned long long *__restrict__ n_heavy0,
                                                        double *__restrict__ sum0, rpl_item *__restrict__ items,
                                                        uint32_t *__restrict__ heavy0) {
  constexpr int RPT = 4;   // rows per thread and chunk
  __shared__ uint32_t s_bucket[RPL_NKEY], s_hist0[RPL_NKEY];
  __shared__ unsigned long long s_at[RPL_NKEY];
  __shared__ double s_sum;
  const uint32_t tid = threadIdx.x;
  const uint64_t per_block = (n_rows + gridDim.x - 1) / gridDim.x;
  const uint64_t r_begin = (uint64_t)blockIdx.x * per_block, r_end = r_begin + per_block < n_rows ? r_begin + per_block : n_rows;
  for (uint64_t base = r_begin; base < r_end; base += 256u * RPT) {
    if (tid < RPL_NKEY) s_bucket[tid] = s_hist0[tid] = 0u;
    if (tid == 0) s_sum = 0.0;
    __syncthreads();
    uint32_t c[RPT][4], rb[RPT][4], rank[RPT][4];
    double R[RPT];
    double my_sum = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const uint64_t row = base + tid + 256u * k;
      R[k] = -1.0;
#pragma unroll
      for (int b = 0; b < 4; ++b) c[k][b] = rb[k][b] = rank[k][b] = 0u;
      if (row < r_end) {
        uint64_t rs = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          c[k][b] = train[row * 5 + b];
          rb[k][b] = ref[row * 5 + b];
          rs += rb[k][b];
        }
        R[k] = (double)rs;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (c[k][b] == 0u) continue;
          if (rs == 0) {
            my_sum += (double)c[k][b];
            if (c[k][b] <= SRT_CL) {
              if (!fill) atomicAdd(&s_hist0[c[k][b] - 1], 1u);
            } else if (fill) {
              heavy0[atomicAdd(n_heavy0, 1ull)] = c[k][b];
            } else {
              atomicAdd(n_heavy0, 1ull);
            }
          } else {
            const uint32_t key = (c[k][b] > SRT_CL ? SRT_CL + 1u : c[k][b]) - 1u;
            rank[k][b] = atomicAdd(&s_bucket[key], 1u);
          }
        }
      }
    }
    if (!fill && my_sum != 0.0) atomicAdd(&s_sum, my_sum);
    __syncthreads();
    if (tid < RPL_NKEY && !fill) {
      if (s_bucket[tid]) atomicAdd(&bucket[tid], (unsigned long long)s_bucket[tid]);
      if (s_hist0[tid]) atomicAdd(&hist0[tid], (unsigned long long)s_hist0[tid]);
    }
    if (!fill && tid == 0 && s_sum != 0.0) atomicAdd(sum0, s_sum);
    if (fill) {
      // this chunk's slice of every bucket: one global cursor bump per bucket and chunk, then ranks inside the slice
      if (tid < RPL_NKEY) s_at[tid] = s_bucket[tid] ? atomicAdd(&bucket[tid], (unsigned long long)s_bucket[tid]) : 0ull;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        if (!(R[k] > 0.0)) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (c[k][b] == 0u) continue;
          const uint32_t key = (c[k][b] > SRT_CL ? SRT_CL + 1u : c[k][b]) - 1u;
          rpl_item it;
          it.c = c[k][b];
          it.rb = rb[k][b];
          it.R = R[k];
          items[s_at[key] + rank[k][b]] = it;
        }
      }
    }
    __syncthreads();
  }
}

// AR: multinomial mode of bear_ref (train_ar): sum LL = sum c log(f + eps); gradients w.r.t. tau_s, nu_s only.
template <bool AR>
__global__ __launch_bounds__(256) void dm_ref_items_kernel(bear_params prm_arg, rpl_view rv, pln_view pv, const double2 *__restrict__ logtab_g,
                                                           double *__restrict__ partials, const bear_step_io io, const bear_apply_io apply) {
  __shared__ double2 s_log[BEAR_LOGTAB_N];
  const bear_params prm = bear_params_of(prm_arg, io);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const double u = prm.inv_h, eps = prm.eps;
  const double A = u + 5.0 * eps;              // sum_b alpha_b
  const double x4 = prm.nw * prm.V * u + eps;  // alpha of the stop column
  const double VU = prm.V * u;
  const double tau = prm.tau;
  const double w2c = tau * (eps + 0.25 * VU);  // d alpha/d tau_s = -tau x + w2c
  const double nwV = prm.nw * prm.V;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (tid < BEAR_LOGTAB_N) s_log[tid] = logtab_g[tid];
  __syncthreads();
  // bear_ref.py:30-33 (Jukes-Cantor on the L1-normalised reference row), :63-68 (mix), bear_ref.py:106 -- as dm_ref_plan_kernel
  auto alpha_from = [&](double rb, double R) {
    const double dev = __builtin_fma(rb + eps, bear_rcp(R), -0.25);
    return __builtin_fma(__builtin_fma(prm.E, dev, 0.25), VU, eps);
  };
  auto accumulate = [&](double x, double m, const bear_dp &o) {   // m: multiplicity
    const double w1 = eps - x, P = m * o.P;
    acc[0] = __builtin_fma(m, o.D, acc[0]);
    acc[1] = __builtin_fma(w1, P, acc[1]);
    acc[2] = __builtin_fma(__builtin_fma(-tau, x, w2c), P, acc[2]);
    acc[3] = __builtin_fma(nwV * w1, P, acc[3]);
  };
  auto accumulate_ar = [&](double rb, double R, double c) {
    const double dev = __builtin_fma(rb + eps, bear_rcp(R), -0.25);
    const double f = __builtin_fma(prm.E, dev, 0.25) * prm.V;
    const double p = f + eps;
    const double dLdf = c * bear_rcp(p);
    acc[0] = __builtin_fma(c, bear_log_tab(p, s_log), acc[0]);
    acc[2] = __builtin_fma(dLdf, -prm.tauE * dev * prm.V, acc[2]);  // d f / d tau_s
    acc[3] = __builtin_fma(dLdf, -nwV * f, acc[3]);                 // d f / d nu_s (net function is 0 here)
  };
  // ---- items of contexts with reference counts: units of 64 sorted records
  const uint64_t n_units = (rv.n_items + 63u) >> 6;
  const uint64_t wave_g = ((uint64_t)blockIdx.x * 256u + tid) >> 6, n_waves = ((uint64_t)gridDim.x * 256u) >> 6;
  for (uint64_t un = wave_g; un < n_units; un += n_waves) {
    const uint64_t i = un * 64u + lane;
    rpl_item it;
    it.c = 0u;
    it.rb = 0u;
    it.R = 1.0;
    if (i < rv.n_items) it = rv.items[i];
    const double R = it.R + 4.0 * eps;
    if (AR) {
      if (it.c) accumulate_ar((double)it.rb, R, (double)it.c);
      continue;
    }
    const bool heavy = it.c > SRT_CL;
    const uint32_t cc = heavy ? SRT_CL + 1u : it.c;
    const uint32_t occ = (un + 1u) * 64u <= rv.n_items ? 64u : (uint32_t)(rv.n_items - un * 64u);
    const uint32_t cmin = occ == 64u ? (uint32_t)__builtin_amdgcn_readlane((int)cc, 0) : 0u;
    const uint32_t cmax = (uint32_t)__builtin_amdgcn_readlane((int)cc, (int)(occ - 1u));
    const double x[1] = {alpha_from((double)it.rb, R)};
    const uint32_t ci[1] = {heavy ? 0u : it.c};
    bear_dp o[1];
    srt_light<1>(x, ci, cmin > SRT_CL ? SRT_CL : cmin, cmax > SRT_CL ? SRT_CL : cmax, s_log, o);
    if (__builtin_amdgcn_ballot_w64(heavy)) {
      if (heavy) o[0] = srt_general_fast(x[0], (double)it.c, s_log);
    }
    if (it.c) accumulate(x[0], 1.0, o[0]);
  }
  // ---- contexts without reference counts: one concentration, a histogram over the count
  const uint64_t gtid = (uint64_t)blockIdx.x * 256u + tid, gsz = (uint64_t)gridDim.x * 256u;
  const double R0 = 4.0 * eps;
  if (AR) {
    if (gtid == 0) accumulate_ar(0.0, R0, rv.sum0[0]);
    // stop column: f_4 = nw V for every context, so its term is (sum of all stop counts) log(f_4 + eps)
    const double f4 = nwV, p4 = f4 + eps;
    double c4sum = 0.0;
    if (blockIdx.x == 0 && tid < SRT_CL) c4sum = (double)(tid + 1) * (double)pv.hist[SRT_NKEY + tid];
    for (uint64_t i = gtid; i < pv.n_heavy_stop; i += gsz) c4sum += (double)pv.heavy_stop[i];
    acc[0] = __builtin_fma(c4sum, bear_log_tab(p4, s_log), acc[0]);
    acc[3] = __builtin_fma(c4sum * bear_rcp(p4), nwV * (1.0 - f4), acc[3]);  // d f_4 / d nu_s = nw V (1 - f_4)
    __syncthreads();
    block_finish<4>(acc, partials, io, apply);
    return;
  }
  const double x0 = alpha_from(0.0, R0);
  if (blockIdx.x == 0 && tid < SRT_CL) {
    const double m = (double)rv.hist0[tid];
    if (m != 0.0) accumulate(x0, m, srt_general_fast(x0, (double)(tid + 1), s_log));
  }
  for (uint64_t i = gtid; i < rv.n_heavy0; i += gsz) accumulate(x0, 1.0, srt_general_fast(x0, (double)rv.heavy0[i], s_log));
  // ---- context terms and the stop column: shared concentrations, the plan's histograms (as dm_ref_plan_kernel)
  for (uint64_t i = gtid; i < pv.n_heavy_row; i += gsz) {
    const double n = pv.heavy_row[i].n;
    if (pln_in_big_hist(pv, n)) continue;      // (totals up to PLN_NBIG: the plan's histogram, next line)
    const bear_dp o = srt_general_fast(A, n, s_log);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(u, o.P, acc[1]);
  }
  pln_big_totals(pv, A, u, gtid, gsz, s_log, acc[0], acc[1]);
  for (uint64_t i = gtid; i < pv.n_heavy_stop; i += gsz) {
    const bear_dp o = srt_general_fast(x4, (double)pv.heavy_stop[i], s_log);
    acc[0] += o.D;
    acc[1] = __builtin_fma(eps - x4, o.P, acc[1]);
    acc[3] = __builtin_fma(VU * nwV, o.P, acc[3]);
  }
  if (blockIdx.x == 1 % gridDim.x && tid < SRT_CL) {
    const double mn = (double)pv.hist[tid], m4 = (double)pv.hist[SRT_NKEY + tid];
    const bear_dp on = srt_general_fast(A, (double)(tid + 1), s_log), o4 = srt_general_fast(x4, (double)(tid + 1), s_log);
    acc[0] -= mn * on.D;                                          // context terms: -D(A, n)
    acc[1] = __builtin_fma(u * mn, on.P, acc[1]);
    const double P4 = m4 * o4.P;                                  // stop column: +D(x4, c)
    acc[0] += m4 * o4.D;
    acc[1] = __builtin_fma(eps - x4, P4, acc[1]);
    acc[3] = __builtin_fma(VU * nwV, P4, acc[3]);                 // d alpha_4/d nu_s = u nw V^2
  }
  __syncthreads();
  block_finish<4>(acc, partials, io, apply);
}

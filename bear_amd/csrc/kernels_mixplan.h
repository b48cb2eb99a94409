// kernels_mixplan.h -- bear_ref's training step for a net function WITH parameters, on the plan: the reference mixing inside the DM
// step (bear_dm_refmix_plan_grad_f64).
//
//   f_i = (nw g_i + jc_i) / (nw + 1)                   g_i = the net function's row (linear, cnn, any plugin; normalised),
//   jc_ib = shape_b / 4 + exp(-tau) d_ib               d_ib = r_ib / sum_c |r_ic| - shape_b / 4,   shape = (1,1,1,1,0)
//   sum LL, d/dh_s                                      as dm_prior_plan_grad_kernel on the rows f        (bear_ref.py:207-259)
//   d sum LL / d g_ib = q_ib nw / (nw + 1)              q_ib = d sum LL / d f_ib = u P(x_ib, c_ib) - u P(A, n_i)
//   d/d tau_signed        = -tau exp(-tau) / (nw + 1)  sum_ib q_ib d_ib
//   d/d net_weight_signed = nw / (nw + 1)^2            sum_ib q_ib (g_ib - jc_ib)  =  1 / (nw + 1)  sum_ib q_ib (f_ib - jc_ib)
//
// With the stop net function all of this lives in the mode-R kernels.  With a net function that has parameters the rows g exist,
// and the step was three launches around them -- bear_ref_mix_forward (120 B per context), the gradient-row kernel (87 B) and
// bear_ref_mix_backward (160 B): 6.6 ms per 1e8 contexts.  Here: g and the reference rows in (80 B), the plan (7 B), d/dg out
// (40 B), one launch.  What makes it one pass:
//   * sum_b d_ib = 0 and sum_b (f_ib - jc_ib) = 0 (g, jc and f are all normalised), so the context term -u P(A, n), common to the
//     five cells of a row, drops out of both parameter gradients: only ITEM cells enter them, and an item has f_ib in hand;
//   * a row pass turns (g, r) into (f, d) in place; an item reads f and d of its own cell, adds its terms, leaves u P in the
//     d cell and marks the f cell with the sign bit (f >= 0; a cell belongs to at most one item); a second row pass forms
//     d/dg = (marked ? u P : 0) - u P(A, n) per cell, scaled by nw / (nw + 1), and the rows leave as one coalesced stream.
// Single-buffered like dm_prior_plan_grad_kernel (two row buffers fill the LDS).
// Rows g must be normalised (every reference net function ends in a softmax) and the alphabet has four letters.  Items and
// contexts in the plan's global overflow lists (very dense tiles only) are handled by the epilogue and dm_refmix_fixup_kernel.
// AR (train_ar, the multinomial of core.py:138-139): sum LL = sum c log(f + eps), q = c / (f + eps) on item cells only, no context
// terms and no h gradient -- the same passes with less in them.
#pragma once
#include "kernels_plan.h"
#include "kernels_refmix.h"

struct mxp_consts {
  double u, eps, nw, V, E, nwV;
};
// constants from the three signed parameters, read on the device (the optimizer's tensors)
__device__ __forceinline__ mxp_consts mxp_load(const double *h_s, const double *tau_s, const double *nw_s, double eps, double *tau_out) {
  mxp_consts C;
  const double tau = exp(tau_s[0]), nw = exp(nw_s[0]);
  C.u = bear_uniform_f64(1.0 / exp(h_s[0]));
  C.eps = eps;
  C.nw = bear_uniform_f64(nw);
  C.V = bear_uniform_f64(1.0 / (nw + 1.0));
  C.E = bear_uniform_f64(exp(-tau));
  C.nwV = C.nw * C.V;
  *tau_out = bear_uniform_f64(tau);
  return C;
}
// one cell of a row in global memory (overflow lists): f_b, d_b, jc_b
__device__ __forceinline__ void mxp_cell(const double *__restrict__ g_row, const double *__restrict__ r_row, uint32_t b, const mxp_consts &C,
                                         double *f, double *d, double *jc) {
  double r[5], dd[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) r[k] = r_row[k];
  rmx_dev(r, dd);
  double db = dd[0];
#pragma unroll
  for (int k = 1; k < 5; ++k) db = b == (uint32_t)k ? dd[k] : db;
  *d = db;
  *jc = (b < 4u ? 0.25 : 0.0) + C.E * db;
  *f = (C.nw * g_row[b] + *jc) * C.V;
}

template <bool AR>
__global__ __launch_bounds__(PLN_THREADS, 4) void dm_refmix_plan_grad_kernel(
    const double *__restrict__ net_rows, const double *__restrict__ ref_rows, const double *__restrict__ h_s,
    const double *__restrict__ tau_s, const double *__restrict__ nw_s, double eps_arg, pln_view pv, const double2 *__restrict__ logtab_g,
    double *__restrict__ grad_out, double *__restrict__ partials, const bear_step_io io) {
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_g &S = *reinterpret_cast<pln_lds_g *>(srt_smem);   // pri: g -> f (item cells: sign-marked); grad: r -> d -> u P -> d/dg
  double tau;
  const mxp_consts C = mxp_load(h_s, tau_s, nw_s, eps_arg, &tau);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
  const double u = C.u, eps = C.eps, eps5 = 5.0 * C.eps;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};   // sum LL | sum (eps - x) P | sum q d | sum q (f - jc)
  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid < SRT_NKEY) {
    const bear_dp o = srt_general_fast(u + eps5, (double)(tid + 1), logtab_g);
    S.tabD[tid] = o.D;
    S.tabP[tid] = o.P;
  }
  if (tid == 0) {
    S.pri[PLN_SENTINEL] = 1.0;    // what the unoccupied lanes of an item unit read (their count is zero)
    S.grad[PLN_SENTINEL] = 0.0;
    S.ticket = PLN_TICKET_START(PLN_WAVES);
  }
  pln_tile nxt = pln_load_tile(pv, blockIdx.x);      // descriptors one tile ahead (see dm_prior_plan_grad_kernel)
  for (uint64_t t = blockIdx.x; t < pv.n_tiles; t += gridDim.x) {
    const pln_tile cur = nxt;
    const uint32_t rows = cur.rows_items >> 16, n_light = cur.rows_items & 0xffffu;
    const uint32_t hc = cur.hc_hr >> 16, hr = cur.hc_hr & 0xffffu;
    const pln_layout L = pln_block_layout(rows, n_light, hc, hr);
    __syncthreads();  // previous tile written out
    {
      const uint32_t pbytes = rows * 40u, pieces = (pbytes + 1023u) >> 10;
      pln_dma(S.pri, net_rows + cur.row0 * 5, pbytes & ~15u, wave, lane, 0);
      pln_dma(S.grad, ref_rows + cur.row0 * 5, pbytes & ~15u, wave, lane, pieces);
      if (pbytes & 15u) {   // odd row count (last tile): the trailing doubles through the scalar path (see dm_prior_plan_kernel)
        const __attribute__((address_space(4))) double *tg =
            (const __attribute__((address_space(4))) double *)(uintptr_t)(net_rows + (cur.row0 + rows) * 5 - 1);
        const __attribute__((address_space(4))) double *tr =
            (const __attribute__((address_space(4))) double *)(uintptr_t)(ref_rows + (cur.row0 + rows) * 5 - 1);
        const double vg = *tg, vr = *tr;
        if (tid == 0) {
          S.pri[rows * 5 - 1] = vg;
          S.grad[rows * 5 - 1] = vr;
        }
      }
      pln_dma(S.blk, pv.stream + (size_t)cur.off16 * 16, cur.blk16 * 16u, wave, lane, 2u * pieces);
    }
    if (tid == 0) S.ticket = PLN_TICKET_START(PLN_WAVES);
    nxt = pln_load_tile(pv, t + gridDim.x);
    srt_wait_dma();
    srt_sync();
    const uint16_t *E = reinterpret_cast<const uint16_t *>(S.blk);
    const uint8_t *nrow = S.blk + L.nrow;
    const uint16_t *items = reinterpret_cast<const uint16_t *>(S.blk + L.items);
    // ---- 1: one thread per row: (g, r) -> (f, d) in place; context terms of the small totals (A = u + 5 eps for every context)
    for (uint32_t row = tid; row < rows; row += PLN_THREADS) {
      double g[5], r[5], d[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        g[b] = S.pri[row * 5 + b];
        r[b] = S.grad[row * 5 + b];
      }
      {   // d_b = r_b / sum |r| - shape_b / 4 with ONE reciprocal (rmx_dev's five divisions would cost as much as the tile's items)
        const double inv = bear_rcp(((__builtin_fabs(r[0]) + __builtin_fabs(r[1])) + (__builtin_fabs(r[2]) + __builtin_fabs(r[3]))) + __builtin_fabs(r[4]));
#pragma unroll
        for (int b = 0; b < 5; ++b) d[b] = __builtin_fma(r[b], inv, b < 4 ? -0.25 : 0.0);
      }
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        S.pri[row * 5 + b] = (C.nw * g[b] + ((b < 4 ? 0.25 : 0.0) + C.E * d[b])) * C.V;
        S.grad[row * 5 + b] = d[b];
      }
      const uint32_t n = nrow[row];
      if (!AR && n != 0 && n != 255u) {
        acc[0] -= S.tabD[n - 1];
        acc[1] = __builtin_fma(u, S.tabP[n - 1], acc[1]);
      }
    }
    srt_sync();
    // (No L2 prefetch of the next tile here, unlike dm_prior_plan_grad_kernel: two row arrays per tile are more than the L2 holds next
    // to the tiles in flight -- touching both bought nothing, 2.62 ms either way, and 60 of 146 B per context fetched twice; the net
    // rows alone 2.52 ms at 111 B.  Without it FETCH_SIZE is the algorithmic 87 B.)
    // ---- 2: item units (tickets, dearest first): sum LL, d/dh, the two parameter sums; u P into the d cell, the f cell marked
    // (o.D = the item's log-likelihood term, o.P = q / u: BEAR mode D, P of the DM item; multinomial mode c log(f + eps), c / (f + eps) / u)
    auto item = [&](uint32_t off, double x, double fb, const bear_dp &o, bool on) {
      const double uP = u * o.P, db = S.grad[off];
      const uint32_t b = off - 5u * (uint32_t)(((unsigned long long)off * 52429ull) >> 18);   // off % 5 for off < 2^16
      const double jc = (b < 4u ? 0.25 : 0.0) + C.E * db;
      if (on) {
        acc[0] += o.D;
        if (!AR) acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
        acc[2] = __builtin_fma(uP, db, acc[2]);
        acc[3] = __builtin_fma(uP, fb - jc, acc[3]);
        S.grad[off] = uP;
        S.pri[off] = -fb;
      }
    };
    auto multinomial = [&](double fb, double cnt) {   // c log(f + eps) and c / (f + eps), the latter over u (item() multiplies it back)
      const double pp = fb + eps;
      return bear_dp{cnt * bear_log_tab(pp, S.logtab), cnt * bear_rcp(pp) * bear_rcp(u)};
    };
    const uint32_t n_hcu = (hc + 63u) >> 6, n_units = (n_light + 63u) >> 6;
    PLN_FOR_UNITS_F(w, &S.ticket, n_hcu + n_units, wave, PLN_WAVES) {
      if (w < n_hcu) {
        const uint32_t i = w * 64u + lane;
        if (i < hc) {
          const uint32_t off = reinterpret_cast<const uint16_t *>(S.blk + L.hoff)[i];
          const double cnt = (double)reinterpret_cast<const uint32_t *>(S.blk + L.hcnt)[i];
          const double fb = S.pri[off], x = __builtin_fma(fb, u, eps);
          const bear_dp o = AR ? multinomial(fb, cnt) : srt_general_fast(x, cnt, S.logtab);
          item(off, x, fb, o, true);
        }
        continue;
      }
      const uint32_t un = n_hcu + n_units - 1u - w;
      uint32_t cmin, cmax;
      const uint32_t ci[1] = {pln_unit_counts(E, n_light, un, lane, &cmin, &cmax)};
      const uint32_t off = items[un * 64u + lane];
      const double fb = S.pri[off];
      const double x[1] = {__builtin_fma(fb, u, eps)};
      bear_dp o[1];
      if (AR) o[0] = multinomial(fb, (double)ci[0]);
      else srt_light<1>(x, ci, cmin, cmax, S.logtab, o);
      item(off, x[0], fb, o[0], ci[0] != 0);
    }
    srt_sync();
    // ---- 3: one thread per row: d/dg = ((marked ? u P : 0) - u P(A, n)) nw / (nw + 1); rows without counts get exact zeros
    for (uint32_t row = tid; row < rows; row += PLN_THREADS) {
      const uint32_t n = nrow[row];
      const double base = (!AR && n != 0 && n != 255u) ? -u * S.tabP[n - 1] : 0.0;   // large totals: step 4 / fix-up kernel
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const bool marked = n != 0 && __builtin_signbit(S.pri[row * 5 + b]);
        S.grad[row * 5 + b] = ((marked ? S.grad[row * 5 + b] : 0.0) + base) * C.nwV;
      }
    }
    srt_sync();
    // ---- 4: contexts of this tile with a large total: their context term, and its base on their five cells
    for (uint32_t i = tid; !AR && i < hr; i += PLN_THREADS) {
      const uint32_t row = reinterpret_cast<const uint16_t *>(S.blk + L.hrow)[i];
      const bear_dp o = srt_general_fast(u + eps5, reinterpret_cast<const double *>(S.blk + L.hn)[i], S.logtab);
      acc[0] -= o.D;
      acc[1] = __builtin_fma(u, o.P, acc[1]);
#pragma unroll
      for (int b = 0; b < 5; ++b) S.grad[row * 5 + b] -= u * o.P * C.nwV;
    }
    __syncthreads();
    // ---- 5: the tile's d/dg rows leave as one coalesced stream
    {
      const uint32_t n_dw = rows * 10u;  // dwords
      const uint4 *src = reinterpret_cast<const uint4 *>(S.grad);
      uint4 *dst = reinterpret_cast<uint4 *>(grad_out + cur.row0 * 5);
      for (uint32_t i = tid; i < (n_dw >> 2); i += PLN_THREADS) dst[i] = src[i];
      if ((n_dw & 3u) && tid == 0) grad_out[(cur.row0 + rows) * 5 - 1] = S.grad[rows * 5 - 1];  // odd row count
    }
  }
  // ---- the scalar terms of the items and contexts in the global overflow lists (their gradient cells: dm_refmix_fixup_kernel)
  __syncthreads();
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    const uint64_t row = h.off / 5u;
    double fb, db, jc;
    mxp_cell(net_rows + row * 5, ref_rows + row * 5, (uint32_t)(h.off - row * 5u), C, &fb, &db, &jc);
    const double x = __builtin_fma(fb, u, eps);
    const bear_dp o = AR ? bear_dp{(double)h.c * bear_log_tab(fb + eps, S.logtab), (double)h.c * bear_rcp(fb + eps) * bear_rcp(u)}
                         : srt_general_fast(x, (double)h.c, S.logtab);
    acc[0] += o.D;
    if (!AR) acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
    acc[2] = __builtin_fma(u * o.P, db, acc[2]);
    acc[3] = __builtin_fma(u * o.P, fb - jc, acc[3]);
  }
  for (uint64_t i = gtid; !AR && i < pv.n_heavy_row; i += gsz) {
    const bear_dp o = srt_general_fast(u + eps5, pv.heavy_row[i].n, S.logtab);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(u, o.P, acc[1]);
  }
  acc[2] *= -tau * C.E * C.V;   // d/d tau_signed
  acc[3] *= C.V;                // d/d net_weight_signed
  __syncthreads();
  block_finish<4>(acc, partials, io);
}

// Gradient cells of the items / contexts in the plan's global overflow lists, scaled like the rest (rare path: fp64 atomics).
template <bool AR>
__global__ __launch_bounds__(256) void dm_refmix_fixup_kernel(const double *__restrict__ net_rows, const double *__restrict__ ref_rows,
                                                              const double *__restrict__ h_s, const double *__restrict__ tau_s,
                                                              const double *__restrict__ nw_s, double eps_arg, pln_view pv,
                                                              const double2 *__restrict__ logtab_g, double *__restrict__ grad_out) {
  double tau;
  const mxp_consts C = mxp_load(h_s, tau_s, nw_s, eps_arg, &tau);
  __shared__ double2 logtab[BEAR_LOGTAB_N];
  if (threadIdx.x < BEAR_LOGTAB_N) logtab[threadIdx.x] = logtab_g[threadIdx.x];
  __syncthreads();
  const double u = C.u, eps = C.eps, eps5 = 5.0 * C.eps;
  const uint64_t gtid = (uint64_t)blockIdx.x * 256 + threadIdx.x, gsz = (uint64_t)gridDim.x * 256;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    const uint64_t row = h.off / 5u;
    double fb, db, jc;
    mxp_cell(net_rows + row * 5, ref_rows + row * 5, (uint32_t)(h.off - row * 5u), C, &fb, &db, &jc);
    const double q = AR ? (double)h.c * bear_rcp(fb + eps) : u * srt_general_fast(__builtin_fma(fb, u, eps), (double)h.c, logtab).P;
    atomicAdd(&grad_out[h.off], q * C.nwV);
  }
  for (uint64_t i = gtid; !AR && i < pv.n_heavy_row; i += gsz) {
    const pln_heavy_row h = pv.heavy_row[i];
    const bear_dp o = srt_general_fast(u + eps5, h.n, logtab);
    for (int b = 0; b < 5; ++b) atomicAdd(&grad_out[h.row * 5 + b], -u * o.P * C.nwV);
  }
}

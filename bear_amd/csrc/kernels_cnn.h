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
// the first tensordot).  Backward recomputes the cheap per-position quantities; sums over contexts go through LDS
// staging into fp64 MFMA products / column sums (see the backward section), one partial vector per block, fixed-order
// finalize.
//
// Packed parameter vector (doubles), the reference's parameter order (ar_funcs.py:98-99):
//   filters [fw][5][nf] | intercept0 [P][nf] | weights1 [P][nf][l1] | intercept1 [l1] | weights2 [l1][5] |
//   intercept2 [5] | scale0 [P][nf] | scale1 [l1]
#pragma once
#include "bear_common.h"
#include "kernels_plan.h"   // pln_tile, PLN_LIVE_STRIDE: the training step runs over the plan's lists of contexts that hold counts

#define CNN_NF 30
#define CNN_L1 16
#define CNN_THREADS 256
#define CNN_MAX_LAG 21
#define CNN_LN_EPS 1e-5
#ifndef CNN_SHARED_RUNS
#define CNN_SHARED_RUNS 6            // backward: a position is done per distinct window when a tile holds at most this many (cnn_backward_shared_window)
#endif
#ifndef CNN_FWD_RUNS
#define CNN_FWD_RUNS 4               // forward: a position is done per distinct window when a wave holds at most this many
#endif
#define CNN_FWD_KEEP 3            // forward: the layer-1 contributions of the last fully shared window of positions < this are kept per wave
#define CNN_FWD_SCRATCH (96 + 16 * CNN_FWD_KEEP)   // doubles of LDS per wave of the forward kernel: shared-window scratch [96] | kept contributions

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

// ---- prefix levels (bear_plan_attach_cnn_levels).  In a k-mer-sorted batch the window of position p = letters [p, p + fw) is
// shared by every context with the same first p + fw letters, and such contexts are neighbours: position p is evaluated once per
// DISTINCT prefix of p + fw letters instead of once per context.  Level 0 are the contexts themselves and take the last position;
// level k >= 1 are the distinct prefixes of lag - k letters (as packed contexts whose trailing letters are "unknown"), each with a
// row of 16 layer-1 sums: its own position P - 1 - k plus the row of its parent at level k + 1 (the last level takes all the
// positions that are left).  Forward runs the levels from the shortest prefixes down to the contexts; backward runs them the
// other way, a level's dT1 rows being the sums of its children's (everything a position does with dT1 is linear in it).  Both
// are the kernels below with a position range and a row source: a dense sorted table does ~1.3 positions per context, not 6.
#define CNN_MAX_WIN 6             // window tables a launch may read (the positions the contexts themselves would evaluate)
struct cnn_level_io {
  int p_lo, p_hi;                 // the positions this launch evaluates
  int head;                       // 1: the rows are contexts -- layer 1 onwards (forward: prior rows; backward: dT1 from the head)
  int accumulate;                 // backward: add this launch's block partials to what the buffer holds (a later level of a step)
  const double *t1_parent;        // forward: [n_parent][16] sums of the earlier positions, or NULL (the last level)
  const uint32_t *parent;         // forward: [n_rows] row of t1_parent
  double *dT1;                    // backward: head -> the contexts' dT1 rows are also written here [n_rows][16] (NULL: not wanted; may be the
                                  // t1 buffer itself); no head -> the rows' dT1 are READ from here
  // forward: window tables (bear_window_dev) of positions this launch does NOT evaluate itself: a row adds its window's row of each
  int n_win;
  const double *win_rows[CNN_MAX_WIN];        // [n_windows][16]
  const uint32_t *win_row_of[CNN_MAX_WIN];    // [n_rows] the window row of each row of this launch
};
static inline cnn_level_io cnn_all_positions(const cnn_dims &D) {
  cnn_level_io io;
  io.p_lo = 0;
  io.p_hi = D.P;
  io.head = 1;
  io.accumulate = 0;
  io.t1_parent = nullptr;
  io.parent = nullptr;
  io.dT1 = nullptr;
  io.n_win = 0;
  for (int q = 0; q < CNN_MAX_WIN; ++q) {
    io.win_rows[q] = nullptr;
    io.win_row_of[q] = nullptr;
  }
  return io;
}

// LDS image of the filter bank: 6 letter rows per tap (row 5 = zeros: characters outside the alphabet, core.py:173)
__device__ __forceinline__ void cnn_stage_filters(double *Fs, const double *__restrict__ params, const cnn_dims &D) {
  for (int k = threadIdx.x; k < D.fw * 6 * CNN_NF; k += CNN_THREADS) {
    const int w = k / (6 * CNN_NF), r = k - w * 6 * CNN_NF, a = r / CNN_NF, f = r - a * CNN_NF;
    Fs[k] = a < 5 ? params[D.oF + (w * 5 + a) * CNN_NF + f] : 0.0;
  }
}

// elu(y) and its derivative: y > 0 ? (y, 1) : (exp(y) - 1, exp(y))     (tf.nn.elu, alpha = 1)
__device__ __forceinline__ double cnn_exp_neg(double z, const double *__restrict__ tab) {   // bear_exp_tab without its branch
  z = z > -700.0 ? z : -700.0;
  const double kf = __builtin_rint(z * 184.66496523378731);
  double r = __builtin_fma(kf, -0x1.62e42fee00000p-8, z);
  r = __builtin_fma(kf, -0x1.a39ef35793c76p-40, r);
  const int ki = (int)kf;
  const double t = tab[ki & (BEAR_EXPTAB_N - 1)];
  double p = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = __builtin_fma(r, p, 1.0 / 6.0);
  p = __builtin_fma(r, p, 0.5);
  p = __builtin_fma(r, p, 1.0);
  const double v = __builtin_fma(t, r * p, t);
  return __longlong_as_double(__double_as_longlong(v) + ((long long)(ki >> 7) << 52));
}

__device__ __forceinline__ double cnn_elu(double y, const double *exptab, double &deriv) {
  const double ex = cnn_exp_neg(y < 0.0 ? y : 0.0, exptab);
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
  // sums in six independent chains: one wave per SIMD has no other wave to hide a 30-long dependent chain behind
  double m6[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
#pragma unroll
  for (int f = 6; f < CNN_NF; ++f) m6[f % 6] += x[f];
  const double mu = (((m6[0] + m6[1]) + (m6[2] + m6[3])) + (m6[4] + m6[5])) * (1.0 / CNN_NF);
  double v6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int f = 0; f < CNN_NF; ++f) {
    x[f] -= mu;
    v6[f % 6] = __builtin_fma(x[f], x[f], v6[f % 6]);
  }
  const double var = ((v6[0] + v6[1]) + (v6[2] + v6[3])) + (v6[4] + v6[5]);
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

// One window shared by all 64 contexts of a wave (or by a run of them): its conv row, layer norm and elu once (lane = filter),
// then its contribution to the 16 layer-1 sums (lane = unit).  Out of line: the forward kernel keeps its registers for the
// per-context path.  TWO (position, window) items go through at once, one per half of the wave (lane & 31 = filter, then = unit): the work is a
// chain of dependent steps, so the second item costs nothing.  Us: e0 of the items [0, 32) | [32, 64), their contributions
// [64, 80) | [80, 96).
__device__ __forceinline__ double cnn_fwd_half_sum(double v) {   // every lane gets the sum over its half of the wave (32 lanes)
  auto dpp = [](double x, auto ctrl_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::value;
    const long long q = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)q, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(q >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (uint32_t)lo);
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x124>{});   // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{});   // row_ror:8
  const long long q = __double_as_longlong(v);
  const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)c[0] << 32) | a[0]) + __longlong_as_double(((long long)c[1] << 32) | a[1]);
}
__device__ __noinline__ void cnn_forward_shared_window2(const double *Fs, const double *exptab, double *Us, const double *__restrict__ s0_all,
                                                        const double *__restrict__ b0_all, const double *__restrict__ W1_all, int fw,
                                                        int p_a, unsigned long long win_a, int p_b, unsigned long long win_b, uint32_t lane) {
  const bool second = lane >= 32u;
  const uint32_t l32 = lane & 31u, f = l32 < CNN_NF ? l32 : CNN_NF - 1;
  const int p = second ? p_b : p_a;
  double xf = 0.0;
  unsigned long long c = second ? win_b : win_a;
  for (int w = 0; w < fw; w += 4) {        // four taps' reads in flight (taps beyond the filter width read a zero row)
    double t4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int a = (int)((c >> (3 * k)) & 7ull);
      t4[k] = Fs[(w + k < fw ? (w + k) * 6 + (a < 5 ? a : 5) : 5) * CNN_NF + f];
    }
    xf += (t4[0] + t4[1]) + (t4[2] + t4[3]);
    c >>= 12;
  }
  const bool in = l32 < CNN_NF;
  const double mu = cnn_fwd_half_sum(in ? xf : 0.0) * (1.0 / CNN_NF);
  const double d = xf - mu;
  const double r = cnn_rsqrt(cnn_fwd_half_sum(in ? d * d : 0.0) * (1.0 / CNN_NF) + CNN_LN_EPS);
  double dv;
  const double e = cnn_elu(__builtin_fma(s0_all[p * CNN_NF + (int)f], d * r, b0_all[p * CNN_NF + (int)f]), exptab, dv);
  double *Ue = Us + (second ? 32 : 0);
  if (in) Ue[l32] = e;
  if (l32 < CNN_L1) {
    const double *__restrict__ W1 = W1_all + p * CNN_NF * CNN_L1 + (int)l32;
    double u4[2] = {0.0, 0.0};
#pragma unroll 6
    for (int ff = 0; ff < CNN_NF; ++ff) u4[ff & 1] = __builtin_fma(Ue[ff], W1[ff * CNN_L1], u4[ff & 1]);
    Us[64 + (second ? 16 : 0) + l32] = u4[0] + u4[1];
  }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(CNN_THREADS, 4) void cnn_forward_kernel(const unsigned long long *__restrict__ codes, uint64_t n_rows,
                                                                   cnn_dims D, const double *__restrict__ params,
                                                                   double *__restrict__ prior, double *__restrict__ t1_save,
                                                                   const pln_tile *__restrict__ tiles, const uint16_t *__restrict__ live_lists,
                                                                   uint64_t n_groups, const cnn_level_io io) {
  extern __shared__ __attribute__((aligned(16))) double cnn_lds[];
  double *exptab = cnn_lds;                       // [128]
  double *Fs = cnn_lds + BEAR_EXPTAB_N;           // [fw][6][nf]
  for (int k = threadIdx.x; k < BEAR_EXPTAB_N; k += blockDim.x) exptab[k] = exp2((double)k * (1.0 / BEAR_EXPTAB_N));   // (a block may be one wave)
  cnn_stage_filters(Fs, params, D);
  __syncthreads();
  // A wave walks groups of contexts, 64 at a time: without lists, group g = rows [64 g, 64 g + 64); with the plan's lists
  // (the training step) group g = plan tile g and only its contexts that hold counts -- nothing reads the others' rows.
  const uint32_t lane = threadIdx.x & 63u, n_waves = CNN_THREADS / 64;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: group numbers, list pointers, row bases
  double *Us = Fs + D.fw * 6 * CNN_NF + wave * CNN_FWD_SCRATCH;     // per wave: e0 of a shared window [32] | its layer-1 sums [16]
  const unsigned long long wmask = 3 * D.fw >= 64 ? ~0ull : (1ull << (3 * D.fw)) - 1ull;
  // A wave takes a CONTIGUOUS range of groups: in a k-mer-sorted table consecutive chunks share the windows of their leading
  // positions, and the contribution of such a window (16 layer-1 sums) is kept in LDS until the window changes.
  const uint64_t wave_id = (uint64_t)blockIdx.x * n_waves + wave, wave_cnt = (uint64_t)gridDim.x * n_waves;
  unsigned long long kept_w[CNN_FWD_KEEP] = {};
  uint32_t kept = 0;                          // bit p: Us[48 + 16 p ..) holds the contribution of window kept_w[p] at position p
  for (uint64_t g = n_groups * wave_id / wave_cnt; g < n_groups * (wave_id + 1) / wave_cnt; ++g) {
    const uint16_t *lst = live_lists ? live_lists + g * PLN_LIVE_STRIDE : nullptr;
    const uint64_t base = lst ? tiles[g].row0 : g * 64;
    const uint32_t cnt = lst ? (uint32_t)lst[0] : (uint32_t)(n_rows - base < 64 ? n_rows - base : 64);
   uint32_t row_next = lst ? (uint32_t)lst[1 + (lane < cnt ? lane : 0u)] : (lane < cnt ? lane : 0u);   // a chunk's list entries are read one chunk ahead
   for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
    const bool live = c0 + lane < cnt;
    const uint64_t i = base + (uint64_t)row_next;                                                        // lanes past the end: the first context
    if (lst && c0 + 64 < cnt) row_next = (uint32_t)lst[1 + (c0 + 64 + lane < cnt ? c0 + 64 + lane : 0u)];
    unsigned long long code = codes[i];              // lanes past the end repeat the chunk's first context (nothing is stored for them)
    {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)code), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(code >> 32));
      if (!live) code = ((unsigned long long)hi << 32) | lo;
    }
    double t1[CNN_L1];
#pragma unroll
    for (int j = 0; j < CNN_L1; ++j) t1[j] = 0.0;
    if (io.t1_parent) {      // a level's rows start from their parent's sums (the earlier positions); neighbours share the parent
      const double2 *src = reinterpret_cast<const double2 *>(io.t1_parent + (size_t)io.parent[i] * CNN_L1);
#pragma unroll
      for (int j = 0; j < CNN_L1 / 2; ++j) {
        const double2 v = src[j];
        t1[2 * j] = v.x;
        t1[2 * j + 1] = v.y;
      }
    }
#pragma unroll
    for (int q = 0; q < CNN_MAX_WIN; ++q) {      // positions that come from window tables: one 128-byte gather each (the tables are a few MB)
      if (q < io.n_win) {
        const double2 *src = reinterpret_cast<const double2 *>(io.win_rows[q] + (size_t)io.win_row_of[q][i] * CNN_L1);
#pragma unroll
        for (int j = 0; j < CNN_L1 / 2; ++j) {
          const double2 v = src[j];
          t1[2 * j] += v.x;
          t1[2 * j + 1] += v.y;
        }
      }
    }
    // In a k-mer-sorted batch (bear_net.train sorts at upload) the 64 contexts of a wave share their leading letters: a window
    // [p, p + fw) that lies inside the shared prefix gives every context the SAME conv row, activations and layer-1
    // contribution.  Then 30 lanes compute the row once (lane = filter), 16 lanes the contribution (lane = unit), and every
    // context adds it: ~200 instructions instead of ~1500 for the position.  The same per DISTINCT window while the wave
    // holds at most CNN_FWD_RUNS of them (the first positions behind the shared prefix); two such items go through
    // cnn_forward_shared_window2 at once.  The first position with more windows and all later ones (in a sorted batch they
    // reach further into the varying letters) take the per-context path below.
    int p_ctx = io.p_hi;
    {
      bool pend = false, keep_a = false;
      int p_a = 0;
      unsigned long long w_a = 0ull;
      auto add_u = [&](int p, unsigned long long w, const double *u) {
        if (((code >> (3 * p)) & wmask) == w) {
#pragma unroll
          for (int j = 0; j < CNN_L1; ++j) t1[j] += u[j];
        }
      };
      auto finish = [&](int p_b, unsigned long long w_b, bool two, bool keep_b) {
        cnn_forward_shared_window2(Fs, exptab, Us, params + D.os0, params + D.ob0, params + D.oW1, D.fw, p_a, w_a, p_b, w_b, lane);
        if (keep_a && lane < CNN_L1) Us[96 + 16 * p_a + lane] = Us[64 + lane];
        if (two && keep_b && lane < CNN_L1) Us[96 + 16 * p_b + lane] = Us[80 + lane];
        add_u(p_a, w_a, Us + 64);
        if (two) add_u(p_b, w_b, Us + 80);
        pend = false;
      };
      for (int p = io.p_lo; p < io.p_hi; ++p) {
        const unsigned long long win = (code >> (3 * p)) & wmask;
        auto next_run = [&](unsigned long long rem, unsigned long long *w) {   // the lanes that share the window of rem's first lane
          const int leader = __builtin_ctzll(rem);
          *w = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(win >> 32), leader) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)win, leader);
          return __builtin_amdgcn_ballot_w64(win == *w) & rem;
        };
        unsigned long long rem = ~0ull, wv;
        for (int runs = 0; rem != 0ull && runs < CNN_FWD_RUNS; ++runs) rem &= ~next_run(rem, &wv);
        if (rem != 0ull) {
          p_ctx = p;
          break;
        }
        for (rem = ~0ull; rem != 0ull;) {
          const unsigned long long m = next_run(rem, &wv);
          rem &= ~m;
          const bool full = m == ~0ull && p < CNN_FWD_KEEP;   // the whole wave shares the window: kept from the previous chunk?
          if (full) {
            bool have = false;
#pragma unroll
            for (int q = 0; q < CNN_FWD_KEEP; ++q)
              if (q == p) {
                have = ((kept >> q) & 1u) && kept_w[q] == wv;
                kept_w[q] = wv;
                kept |= 1u << q;
              }
            if (have) {
              add_u(p, wv, Us + 96 + 16 * p);
              continue;
            }
          }
          if (!pend) {
            p_a = p;
            w_a = wv;
            keep_a = full;
            pend = true;
          } else {
            finish(p, wv, true, full);
          }
        }
      }
      if (pend) finish(p_a, w_a, false, false);
    }
    for (int p = p_ctx; p < io.p_hi; ++p) {
      double x[CNN_NF];
      cnn_conv_norm(Fs, code, p, D.fw, x);
      const double *__restrict__ s0 = params + D.os0 + p * CNN_NF, *__restrict__ b0 = params + D.ob0 + p * CNN_NF;
      const double *__restrict__ W1 = params + D.oW1 + p * CNN_NF * CNN_L1;
      // weights of filter f+1 are fetched (uniform scalar loads) while filter f is consumed; the scheduling barrier keeps
      // the compiler from hoisting all 480 loads of a position at once (that spilled ~1100 SGPRs)
      double wc[CNN_L1], wn[CNN_L1], sc = s0[0], bc = b0[0], sn = 0.0, bn = 0.0;
#pragma unroll
      for (int j = 0; j < CNN_L1; ++j) wc[j] = W1[j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < CNN_NF; ++f) {
        if (f + 1 < CNN_NF) {
          sn = s0[f + 1];
          bn = b0[f + 1];
#pragma unroll
          for (int j = 0; j < CNN_L1; ++j) wn[j] = W1[(f + 1) * CNN_L1 + j];
        }
        double dv;
        const double e = cnn_elu(__builtin_fma(sc, x[f], bc), exptab, dv);
#pragma unroll
        for (int j = 0; j < CNN_L1; ++j) t1[j] = __builtin_fma(e, wc[j], t1[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CNN_L1; ++j) wc[j] = wn[j];
        sc = sn;
        bc = bn;
      }
    }
    if (!io.head) {          // a level of prefixes: its rows of layer-1 sums are all there is
      if (live) {
        double2 *o = reinterpret_cast<double2 *>(t1_save + i * CNN_L1);
#pragma unroll
        for (int j = 0; j < CNN_L1 / 2; ++j) o[j] = make_double2(t1[2 * j], t1[2 * j + 1]);
      }
      continue;
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
}

// ------------------------------------------------------------------------------------------------ backward
// grad_prior [n,5] = d L / d prior rows (what the DM kernel returns).  A wave owns a tile of 64 contexts (lane = context)
// and everything summed over contexts leaves the lanes through LDS staging, never through same-address atomics
// (64 lanes adding to one LDS address serialise: the first version of this kernel ran 80x slower):
//   * d/d weights1[p] = E0_p^T dT1 and d/d filters = OneHot_p^T dConv_p are 16x16x4 fp64 MFMA products whose K
//     dimension is the tile's 64 contexts (operands staged [feature][context], row stride 68 doubles: conflict-free
//     lane=context writes, 2-way -- the minimum for 8-byte reads -- operand reads);
//   * the per-feature sums (scale / intercept gradients, weights2) are staged with row stride 65 and summed by
//     (column, half) lanes;
//   * the tile results are added to the block's gradient image in LDS with one fp64 LDS atomic per lane and value,
//     all lanes on distinct addresses;  d e0 = W1 dT1 stays on the VALU with SGPR weights (as the forward pass).
typedef double cnn_d4 __attribute__((ext_vector_type(4)));
#define CNN_ES 68            // operand staging: row stride in doubles
#define CNN_CS 65            // column-sum staging: row stride in doubles
#define CNN_E_DOUBLES (32 * CNN_ES)
#define CNN_T_DOUBLES (16 * CNN_ES)
#define CNN_WAVE_DOUBLES (CNN_E_DOUBLES + CNN_T_DOUBLES + 64)

__device__ __forceinline__ void cnn_lds_add(double *addr, double v) { atomicAdd(addr, v); }

// sums 32 staged columns (layout [column][context], stride CNN_CS) over the 64 contexts; lane (c = lane & 31, half) adds
// its half-sum to dst[col0 + c] when col0 + c < n_cols
__device__ __forceinline__ void cnn_colsum_add(const double *E, double *dst, int n_cols, uint32_t lane) {
  const double *src = E + (lane & 31u) * CNN_CS + (lane >> 5) * 32u;
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 32; ++r) s4[r & 3] += src[r];
  if ((int)(lane & 31u) < n_cols) cnn_lds_add(dst + (lane & 31u), (s4[0] + s4[1]) + (s4[2] + s4[3]));
}

#ifdef CNN_STAMPS   // developer build: clocks per phase of the backward tile loop, summed per wave (scripts/dev/cnn_stamps.py)
__device__ unsigned long long cnn_stamp_sums[8];
#define CNN_STAMP(k)                                              \
  {                                                               \
    const unsigned long long now = __builtin_amdgcn_s_memtime();  \
    tph[k] += now - t_prev;                                       \
    t_prev = now;                                                 \
  }
#else
#define CNN_STAMP(k)
#endif
__global__ __launch_bounds__(CNN_THREADS) void cnn_backward_kernel(const unsigned long long *__restrict__ codes, uint64_t n_rows,
                                                                    cnn_dims D, const double *__restrict__ params,
                                                                    const double *__restrict__ t1_save,
                                                                    const double *__restrict__ prior,
                                                                    const double *__restrict__ grad_prior,
                                                                    double *__restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) double cnn_lds[];
  double *exptab = cnn_lds;
  double *Fs = cnn_lds + BEAR_EXPTAB_N;
  double *G = Fs + D.fw * 6 * CNN_NF;             // [total] block gradient image, parameter layout
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  double *E = G + ((D.total + 1) & ~1) + wave * CNN_WAVE_DOUBLES;   // [32][CNN_ES] staging
  double *T = E + CNN_E_DOUBLES;                                    // [16][CNN_ES] dT1
  unsigned long long *Cw = reinterpret_cast<unsigned long long *>(T + CNN_T_DOUBLES);   // [64] packed contexts
  for (int k = threadIdx.x; k < BEAR_EXPTAB_N; k += blockDim.x) exptab[k] = exp2((double)k * (1.0 / BEAR_EXPTAB_N));   // (a block may be one wave)
  for (int k = threadIdx.x; k < D.fw * 6 * CNN_NF; k += blockDim.x) {
    const int w = k / (6 * CNN_NF), r = k - w * 6 * CNN_NF, a = r / CNN_NF, f = r - a * CNN_NF;
    Fs[k] = a < 5 ? params[D.oF + (w * 5 + a) * CNN_NF + f] : 0.0;
  }
  for (int k = threadIdx.x; k < D.total; k += blockDim.x) G[k] = 0.0;
  __syncthreads();
  const uint64_t n_tiles = (n_rows + 63) / 64;
#ifdef CNN_STAMPS
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#endif
  const uint32_t lq = lane >> 4, lr = lane & 15u;     // MFMA lane coordinates: k / row-group index, row / column index
  const int n_mt = (4 * D.fw + 15) / 16;              // M tiles of the one-hot operand: rows (tap w, letter a < 4)
  for (uint64_t tile = (uint64_t)blockIdx.x * n_waves + wave; tile < n_tiles; tile += (uint64_t)gridDim.x * n_waves) {
    const uint64_t i = tile * 64 + lane;
    const bool live = i < n_rows;
    // dead lanes: every letter "other" and a zero gradient row -> all their contributions are exact zeros
    unsigned long long code = 0;
#pragma unroll
    for (int l = 0; l < 21; ++l) code |= 5ull << (3 * l);
    if (live) code = codes[i];
    Cw[lane] = code;
    double t1[CNN_L1], n1[CNN_L1], e1[CNN_L1], d1[CNN_L1];
#pragma unroll
    for (int j = 0; j < CNN_L1; ++j) t1[j] = 0.0;
    double pr[5], gp[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      pr[b] = 0.2;
      gp[b] = 0.0;
    }
    if (live) {
      const double2 *src = reinterpret_cast<const double2 *>(t1_save + i * CNN_L1);
#pragma unroll
      for (int j = 0; j < CNN_L1 / 2; ++j) {
        const double2 v = src[j];
        t1[2 * j] = v.x;
        t1[2 * j + 1] = v.y;
      }
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        pr[b] = prior[i * 5 + b];
        gp[b] = grad_prior[i * 5 + b];
      }
    }
    const double r1 = cnn_layer1(t1, params, D, exptab, n1, e1, d1);
    // softmax backward
    double dz[5], sg = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) sg = __builtin_fma(pr[b], gp[b], sg);
#pragma unroll
    for (int b = 0; b < 5; ++b) dz[b] = pr[b] * (gp[b] - sg);
    // layer 2 / layer-1 norm backward; the 117 small gradient columns leave through four 32-column rounds:
    //   round 0: d weights2[j][b], j < 6 (30 columns) ; round 1: j in [6,12) ; round 2: j in [12,16) + d intercept2 (25)
    //   round 3: d scale1 (16) + d intercept1 (16)
    double dy1[CNN_L1];
    {
      double ma = 0.0, mb = 0.0;
#pragma unroll
      for (int j = 0; j < CNN_L1; ++j) {
        double de = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) de = __builtin_fma(params[D.oW2 + j * 5 + b], dz[b], de);
        dy1[j] = de * d1[j];
        const double dn = dy1[j] * params[D.os1 + j];
        t1[j] = dn;
        ma += dn;
        mb = __builtin_fma(dn, n1[j], mb);
      }
      ma *= 1.0 / CNN_L1;
      mb *= 1.0 / CNN_L1;
#pragma unroll
      for (int j = 0; j < CNN_L1; ++j) t1[j] = r1 * (t1[j] - ma - n1[j] * mb);     // dT1
    }
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
#pragma unroll
      for (int c = 0; c < 30; ++c) {
        const int j = rd * 6 + c / 5, b = c % 5;
        if (j < CNN_L1) E[c * CNN_CS + lane] = e1[j] * dz[b];
        else if (j == CNN_L1) E[c * CNN_CS + lane] = dz[b];       // round 2, columns 20..24: d intercept2
      }
      // weights2 [16][5] and intercept2 [5] are adjacent in the parameter vector: columns map to oW2 + 30 rd + c
      cnn_colsum_add(E, G + D.oW2 + 30 * rd, rd < 2 ? 30 : 25, lane);
    }
#pragma unroll
    for (int j = 0; j < CNN_L1; ++j) {
      E[j * CNN_CS + lane] = dy1[j] * n1[j];
      E[(16 + j) * CNN_CS + lane] = dy1[j];
    }
    {
      const double *src = E + (lane & 31u) * CNN_CS + (lane >> 5) * 32u;
      double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 32; ++r) s4[r & 3] += src[r];
      cnn_lds_add(G + ((lane & 31u) < 16u ? D.os1 + (int)(lane & 31u) : D.ob1 + (int)(lane & 31u) - 16),
                  (s4[0] + s4[1]) + (s4[2] + s4[3]));
    }
#pragma unroll
    for (int j = 0; j < CNN_L1; ++j) T[j * CNN_ES + lane] = t1[j];
    double tb[16], tb2[4][4];     // dT1 as the B operand of d weights1 (K = contexts) and of d e0 (K = j), the same for every position
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) tb[ks] = T[lr * CNN_ES + 4 * ks + lq];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) tb2[ks][nt] = T[(4 * ks + lq) * CNN_ES + nt * 16 + lr];
    CNN_STAMP(6)
    // positions
    for (int p = 0; p < D.P; ++p) {
      double x[CNN_NF], dy[CNN_NF], dn[CNN_NF];
      const double r0 = cnn_conv_norm(Fs, code, p, D.fw, x);
      const double *__restrict__ s0 = params + D.os0 + p * CNN_NF, *__restrict__ b0 = params + D.ob0 + p * CNN_NF;
      const double *__restrict__ W1 = params + D.oW1 + p * CNN_NF * CNN_L1;
      double a0[2] = {0.0, 0.0}, a1[2] = {0.0, 0.0};
      // pass A: activations e0 (staged as the A operand of d weights1[p]) and elu'; dy holds elu' until pass B
#pragma unroll
      for (int f = 0; f < CNN_NF; ++f) {
        double dv;
        E[f * CNN_ES + lane] = cnn_elu(__builtin_fma(s0[f], x[f], b0[f]), exptab, dv);
        dy[f] = dv;
      }
      CNN_STAMP(0)
      // d weights1[p][f][j] += sum_ctx e0[ctx][f] dT1[ctx][j]
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        cnn_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(E[(mt * 16 + lr) * CNN_ES + 4 * ks + lq], tb[ks], acc, 0, 0, 0);
        double *g = G + D.oW1 + p * CNN_NF * CNN_L1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = mt * 16 + (int)lq + 4 * r;
          if (f < CNN_NF) cnn_lds_add(g + f * CNN_L1 + lr, acc[r]);
        }
      }
      CNN_STAMP(1)
      // pass B: d e0[f][ctx] = sum_j weights1[p][f][j] dT1[ctx][j] as MFMA products (rows f, columns ctx, K = j), the
      // result handed back to the context lanes through the staging buffer
      {
        double wa[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int f = mt * 16 + (int)lr;
            wa[mt][ks] = f < CNN_NF ? W1[f * CNN_L1 + 4 * ks + (int)lq] : 0.0;
          }
        cnn_d4 c[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) c[mt][nt] = cnn_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) c[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[mt][ks], tb2[ks][nt], c[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = mt * 16 + (int)lq + 4 * r;
            if (f < CNN_NF) {
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) E[f * CNN_ES + nt * 16 + lr] = c[mt][nt][r];
            }
          }
      }
      CNN_STAMP(2)
#pragma unroll
      for (int f = 0; f < CNN_NF; ++f) {
        dy[f] *= E[f * CNN_ES + lane];
        dn[f] = dy[f] * s0[f];
        a0[f & 1] += dn[f];
        a1[f & 1] = __builtin_fma(dn[f], x[f], a1[f & 1]);
      }
      const double ma0 = (a0[0] + a0[1]) * (1.0 / CNN_NF), ma1 = (a1[0] + a1[1]) * (1.0 / CNN_NF);
      // d scale0[p], d intercept0[p]: column sums of dy * n0 and dy
#pragma unroll
      for (int f = 0; f < CNN_NF; ++f) E[f * CNN_CS + lane] = dy[f] * x[f];
      cnn_colsum_add(E, G + D.os0 + p * CNN_NF, CNN_NF, lane);
#pragma unroll
      for (int f = 0; f < CNN_NF; ++f) E[f * CNN_CS + lane] = dy[f];
      cnn_colsum_add(E, G + D.ob0 + p * CNN_NF, CNN_NF, lane);
      CNN_STAMP(3)
      // layer-norm backward -> d conv[p][f], staged as the B operand of d filters
      {
        unsigned long long c = code >> (3 * p);
        bool any_start = false;
        for (int w = 0; w < D.fw; ++w) any_start |= ((c >> (3 * w)) & 7ull) == 4ull;
#pragma unroll
        for (int f = 0; f < CNN_NF; ++f) {
          const double dc = r0 * (dn[f] - ma0 - x[f] * ma1);
          E[f * CNN_ES + lane] = dc;
          dy[f] = dc;
        }
        if (any_start) {   // the start symbol '[' (rare: only contexts at a sequence start) bypasses the MFMA rows
          for (int w = 0; w < D.fw; ++w)
            if (((c >> (3 * w)) & 7ull) == 4ull) {
              double *gF = G + D.oF + (w * 5 + 4) * CNN_NF;
#pragma unroll
              for (int f = 0; f < CNN_NF; ++f) cnn_lds_add(gF + f, dy[f]);
            }
        }
      }
      CNN_STAMP(4)
      // d filters[w][a][f] += sum_ctx [letter_{p+w}(ctx) == a] d conv[ctx][f],  rows (w, a < 4), two column tiles of f
      for (int mt = 0; mt < n_mt; mt += 2) {      // two row tiles per pass share the B operand reads
        cnn_d4 acc[2][2] = {{{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}};
        int sh[2];
        unsigned long long want[2];
        bool row_ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int row = (mt + u) * 16 + (int)lr, w = row >> 2;
          want[u] = (unsigned long long)(row & 3);
          row_ok[u] = w < D.fw;
          sh[u] = row_ok[u] ? 3 * (p + w) : 0;
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const unsigned long long cc = Cw[4 * ks + lq];
          const double b0v = E[lr * CNN_ES + 4 * ks + lq], b1v = E[(16 + lr) * CNN_ES + 4 * ks + lq];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const double a = (row_ok[u] && ((cc >> sh[u]) & 7ull) == want[u]) ? 1.0 : 0.0;
            acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0v, acc[u][0], 0, 0, 0);
            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1v, acc[u][1], 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int orow = (mt + u) * 16 + (int)lq + 4 * r, ow = orow >> 2, oa = orow & 3;
            if (ow < D.fw) {
              double *gF = G + D.oF + (ow * 5 + oa) * CNN_NF;
              cnn_lds_add(gF + lr, acc[u][0][r]);
              if (lr < CNN_NF - 16) cnn_lds_add(gF + 16 + lr, acc[u][1][r]);
            }
          }
      }
      CNN_STAMP(5)
    }
  }
#ifdef CNN_STAMPS
  if (lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&cnn_stamp_sums[k], tph[k]);
#endif
  __syncthreads();
  for (int k = threadIdx.x; k < D.total; k += blockDim.x) partials[(size_t)blockIdx.x * D.total + k] = G[k];
}

// ------------------------------------------------------------------------------------------------ backward, two waves per SIMD
// The kernel above keeps a context's 30 filter responses (three arrays of them) in one lane: 256+ registers and 26 KB of
// staging per wave allow ONE wave per SIMD, and every LDS round trip and MFMA operand fetch of its dependent chains is
// exposed (stamped: no phase dominates, the SIMDs are busy 57 % of the time).  Here a wave owns a tile of 32 contexts and a
// lane is (context = lane & 31, half = lane >> 5): the half owns filters [16 half, 16 half + 16) (the second half holds 14
// real ones) and layer-1 units [8 half, 8 half + 8).  Per-context instruction counts stay what they were (a wave
// instruction covers 32 contexts x 2 features instead of 64 x 1), the register arrays halve, the staging tile halves
// (K = 32 contexts per MFMA product), and eight waves fit a CU: two per SIMD, each hiding the other's stalls.  Sums over a
// context's filters / units cross the parts with v_permlane32_swap (/ v_permlane16_swap).  Parameters a lane reads at a
// part-dependent index come from an LDS image.  Used when the LDS fits (it does for every reference config); otherwise the
// kernel above.
// geometry of the form with Q lanes per context (Q = 2: 32-context tiles, 8 waves)
template <int Q>
struct cnnq {
  // Q = 4 (16-context tiles, twelve waves: three per SIMD) was written and measured: 10.98 ms per 1e7 contexts against 10.27 ms
  // for Q = 2 -- the third wave's latency hiding does not pay for the doubled per-tile overheads -- and was not validated.
  static_assert(Q == 2, "lanes per context: 2");
  static constexpr int TILE = 64 / Q;            // contexts per wave tile
  static constexpr int FH = 32 / Q;              // filter slots per lane (30 filters: the last part holds two dummies)
  static constexpr int JH = CNN_L1 / Q;          // layer-1 units per lane
  static constexpr int ES = TILE + 4;            // operand staging: row stride in doubles ([feature][context])
  static constexpr int CS = TILE + 1;            // column-sum staging: row stride in doubles ([column][context])
  static constexpr int KS = TILE / 4;            // MFMA k-steps over the contexts of a tile
  static constexpr int NT = TILE / 16;           // MFMA column tiles over the contexts of a tile
  static constexpr int CARRY_POS = 2;            // leading positions whose fully shared windows are carried across tiles (below)
  static constexpr int CARRY_DOUBLES = CARRY_POS * CNN_L1 + 4;   // their column sums, then their windows as raw words
  static constexpr int E_DOUBLES = CNN_NF * ES, T_DOUBLES = CNN_L1 * ES, WAVE_DOUBLES = E_DOUBLES + T_DOUBLES + TILE + CARRY_DOUBLES;
  static constexpr int WAVES = Q == 2 ? 8 : 12;  // per block = per CU: two / three per SIMD
  static constexpr int W2Q = CNN_L1 * 5 / Q;     // d weights2 columns a part owns
  static constexpr int COLQ = 32 / Q;            // staging columns of a part per round of 32
  static_assert(32 * CS <= E_DOUBLES, "column-sum staging lives in the operand staging area");
};
// doubles of LDS next to the per-wave staging: exp table, filter image, half-dependent parameter image, gradient image
static inline size_t cnnq_fixed_doubles(const cnn_dims &D) {
  return (size_t)BEAR_EXPTAB_N + (size_t)D.fw * 6 * CNN_NF + 2 * CNN_L1 + CNN_L1 * 5 + 2 * (size_t)D.P * CNN_NF + (size_t)((D.total + 1) & ~1);
}

// v summed over the Q lanes of a context (lanes ctx, ctx + TILE, ...): every one of them gets the sum
template <int Q>
__device__ __forceinline__ double cnnq_psum(double v) {
#pragma unroll
  for (int step = 0; step < (Q == 2 ? 1 : 2); ++step) {
    const long long q = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
    const bool s32 = Q == 2 || step == 1;
    const auto a = s32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto c = s32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __longlong_as_double(((long long)c[0] << 32) | a[0]) + __longlong_as_double(((long long)c[1] << 32) | a[1]);
  }
  return v;
}
// staged columns [column][context] (stride CS): lane (c = lane & 31, hh) returns the sum of column c over its half of the contexts
template <int Q>
__device__ __forceinline__ double cnnq_colsum(const double *E, uint32_t lane) {
  constexpr int HALF = cnnq<Q>::TILE / 2;
  const double *src = E + (lane & 31u) * cnnq<Q>::CS + (lane >> 5) * HALF;
  double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < HALF; ++r) s4[r & 3] += src[r];
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}


// ---- a window shared by every context of a tile (k-mer-sorted batches: the contexts of a tile differ in their last letters).
// Its conv row, layer norm and elu are the same for all of them, and everything the backward pass does with such a position is
// LINEAR in the contexts' dT1 rows: with S = sum over the tile's contexts of dT1,
//   d weights1[p] = e0 (x) S,   d e0 = W1[p] S,   d scale0 / d intercept0 / d conv from d e0 as for one context,
//   d filters[w][letter_w] += d conv for the window's letters
// -- the position costs one context's worth of work (lane = filter) instead of the tile's.  Sx: [0, 16) S, [16, 48) scratch.
template <int CTRL>
__device__ __forceinline__ double cnn_dpp(double v) {
  const long long q = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)q, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(q >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (uint32_t)lo);
}
// the same lane of the four rows of 16 summed (every lane gets the sum): v_permlane16_swap / v_permlane32_swap
__device__ __forceinline__ double cnn_rows_sum(double v) {
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const long long q = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
    const auto a = step == 0 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto c = step == 0 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __longlong_as_double(((long long)c[0] << 32) | a[0]) + __longlong_as_double(((long long)c[1] << 32) | a[1]);
  }
  return v;
}
__device__ __forceinline__ double cnn_half_sum_all(double v) {   // every lane gets the sum over its half of the wave (32 lanes)
  v += cnn_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += cnn_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += cnn_dpp<0x124>(v);   // row_ror:4
  v += cnn_dpp<0x128>(v);   // row_ror:8
  const long long q = __double_as_longlong(v);
  const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto c = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)c[0] << 32) | a[0]) + __longlong_as_double(((long long)c[1] << 32) | a[1]);
}
// TWO (position, window) items per call, one per half of the wave (lane & 31 = filter): the work is a chain of dependent steps
// (conv, two-pass norm, elu, norm backward), so a second item costs nothing.  Sx: per item 48 doubles, [0, 16) S, [16, 48) scratch.
__device__ __forceinline__ void cnn_backward_shared_window(const double *Fs, const double *exptab, double *Sx, const double *Ps0,
                                                           const double *Pb0, const double *__restrict__ W1all, double *G,
                                                           const cnn_dims &D, int p_a, unsigned long long win_a, int p_b,
                                                           unsigned long long win_b, bool two, uint32_t lane_in) {
  uint32_t lane = lane_in;
  asm volatile("" : "+v"(lane));   // nothing derived from the lane number is hoisted out of the tile loop (registers of the position loop)
  const bool second = lane >= 32u;
  const uint32_t l32 = lane & 31u;
  const bool in = l32 < CNN_NF && (two || !second);
  const uint32_t f = l32 < CNN_NF ? l32 : CNN_NF - 1;
  const int p = second ? p_b : p_a;
  const unsigned long long win0 = second ? win_b : win_a;
  double *Sme = Sx + (second ? 48 : 0);
  // the lane's row of weights1[p] comes from global memory: asked for first, used after the conv / norm / elu chain
  double2 wv[CNN_L1 / 2];
  {
    const double2 *wrow = reinterpret_cast<const double2 *>(W1all + (p * CNN_NF + (int)f) * CNN_L1);
#pragma unroll
    for (int j2 = 0; j2 < CNN_L1 / 2; ++j2) wv[j2] = wrow[j2];
  }
  double xf = 0.0;
  unsigned long long c = win0;
  for (int w = 0; w < D.fw; w += 4) {      // four taps' reads in flight (taps beyond the filter width read a zero row)
    double t4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int a = (int)((c >> (3 * k)) & 7ull);
      t4[k] = Fs[(w + k < D.fw ? (w + k) * 6 + (a < 5 ? a : 5) : 5) * CNN_NF + f];
    }
    xf += (t4[0] + t4[1]) + (t4[2] + t4[3]);
    c >>= 12;
  }
  const double mu = cnn_half_sum_all(l32 < CNN_NF ? xf : 0.0) * (1.0 / CNN_NF);
  const double d = xf - mu;
  const double r0 = cnn_rsqrt(cnn_half_sum_all(l32 < CNN_NF ? d * d : 0.0) * (1.0 / CNN_NF) + CNN_LN_EPS);
  const double xn = d * r0, sc = Ps0[p * CNN_NF + (int)f];
  double dv;
  const double e = cnn_elu(__builtin_fma(sc, xn, Pb0[p * CNN_NF + (int)f]), exptab, dv);
  double de[2] = {0.0, 0.0};
#pragma unroll
  for (int j2 = 0; j2 < CNN_L1 / 2; ++j2) {
    de[0] = __builtin_fma(wv[j2].x, Sme[2 * j2], de[0]);
    de[1] = __builtin_fma(wv[j2].y, Sme[2 * j2 + 1], de[1]);
  }
  const double dyv = l32 < CNN_NF ? dv * (de[0] + de[1]) : 0.0, dn = dyv * sc;
  const double ma0 = cnn_half_sum_all(dn) * (1.0 / CNN_NF), ma1 = cnn_half_sum_all(dn * xn) * (1.0 / CNN_NF);
  const double dc = r0 * (dn - ma0 - xn * ma1);
  if (in) {
    Sme[16 + l32] = e;
    cnn_lds_add(G + D.os0 + p * CNN_NF + (int)l32, dyv * xn);
    cnn_lds_add(G + D.ob0 + p * CNN_NF + (int)l32, dyv);
  }
  c = win0;
  for (int w = 0; w < D.fw; ++w) {
    const int a = (int)(c & 7ull);       // 5..7: a character outside the alphabet has no filter row
    if (a < 5 && in) cnn_lds_add(G + D.oF + (w * 5 + a) * CNN_NF + (int)l32, dc);
    c >>= 3;
  }
  // d weights1[p][f][j] += e0[f] S[j]: element l32 + 32 r of the item is (f = (l32 >> 4) + 2 r, j = l32 & 15)
  if (two || !second) {
    const double sj = Sme[l32 & 15u];
    double *g1 = G + D.oW1 + p * CNN_NF * CNN_L1 + (int)l32;
#pragma unroll
    for (int r = 0; r < CNN_NF * CNN_L1 / 32; ++r) cnn_lds_add(g1 + 32 * r, Sme[16 + (l32 >> 4) + 2u * (uint32_t)r] * sj);
  }
}

template <int Q>
__global__ __launch_bounds__(cnnq<Q>::WAVES * 64) void cnn_backward_parts_kernel(const unsigned long long *__restrict__ codes, uint64_t n_rows,
                                                                                  cnn_dims D, const double *__restrict__ params,
                                                                                  const double *__restrict__ t1_save,
                                                                                  const double *__restrict__ prior,
                                                                                  const double *__restrict__ grad_prior,
                                                                                  double *__restrict__ partials,
                                                                                  const pln_tile *__restrict__ tiles,
                                                                                  const uint16_t *__restrict__ live_lists, uint64_t n_groups,
                                                                                  const cnn_level_io io) {
  using C = cnnq<Q>;
  constexpr int TILE = C::TILE, FH = C::FH, JH = C::JH, ES = C::ES, CS = C::CS, KS = C::KS, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) double cnn_lds[];
  double *exptab = cnn_lds;
  double *Fs = cnn_lds + BEAR_EXPTAB_N;
  // the parameters a lane reads at a part-dependent index: scale1 [16] | intercept1 [16] | weights2 [16][5] | scale0 [P][30] | intercept0 [P][30]
  double *Ps1 = Fs + D.fw * 6 * CNN_NF, *Pb1 = Ps1 + CNN_L1, *PW2 = Pb1 + CNN_L1, *Ps0 = PW2 + CNN_L1 * 5, *Pb0 = Ps0 + D.P * CNN_NF;
  double *G = Pb0 + D.P * CNN_NF;                 // [total] block gradient image, parameter layout
  const uint32_t lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: group numbers, list pointers and row bases stay out of the vector registers
  double *E = G + ((D.total + 1) & ~1) + wave * C::WAVE_DOUBLES;   // [30][ES] staging
  double *T = E + C::E_DOUBLES;                                     // [16][ES] dT1
  unsigned long long *Cw = reinterpret_cast<unsigned long long *>(T + C::T_DOUBLES);   // [TILE] packed contexts
  double *Cy = T + C::T_DOUBLES + TILE;                                                // carried column sums and windows
  for (int k = threadIdx.x; k < BEAR_EXPTAB_N; k += blockDim.x) exptab[k] = exp2((double)k * (1.0 / BEAR_EXPTAB_N));   // (a block may be one wave)
  for (int k = threadIdx.x; k < D.fw * 6 * CNN_NF; k += blockDim.x) {
    const int w = k / (6 * CNN_NF), r = k - w * 6 * CNN_NF, a = r / CNN_NF, f = r - a * CNN_NF;
    Fs[k] = a < 5 ? params[D.oF + (w * 5 + a) * CNN_NF + f] : 0.0;
  }
  for (int k = threadIdx.x; k < CNN_L1; k += blockDim.x) {
    Ps1[k] = params[D.os1 + k];
    Pb1[k] = params[D.ob1 + k];
  }
  for (int k = threadIdx.x; k < CNN_L1 * 5; k += blockDim.x) PW2[k] = params[D.oW2 + k];
  for (int k = threadIdx.x; k < D.P * CNN_NF; k += blockDim.x) {
    Ps0[k] = params[D.os0 + k];
    Pb0[k] = params[D.ob0 + k];
  }
  for (int k = threadIdx.x; k < D.total; k += blockDim.x) G[k] = 0.0;
  __syncthreads();
#ifdef CNN_STAMPS
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#endif
  // A wave walks groups of contexts, TILE at a time: without lists, group g = rows [TILE g, TILE g + TILE); with the plan's
  // lists (the training step) group g = plan tile g and only its contexts that hold counts (the others' gradient rows are zero)
  // A wave takes a CONTIGUOUS range of groups: in a k-mer-sorted table consecutive tiles share the windows of their leading
  // positions, and a fully shared window of position 0 or 1 that the next tile shares too is not worked off per tile -- its
  // column sums are carried in Cy (the wave's own LDS words: plain read-add-write) until the window changes.
  const uint64_t wave_id = (uint64_t)blockIdx.x * n_waves + wave, wave_cnt = (uint64_t)gridDim.x * n_waves;
  uint32_t carry = 0;                         // bit p: position p has a carried window
  // (position, window) items queue up in the wave's staging area E (free between position loops): slot k = E[48 k ..): S [16],
  // e0 scratch [30], and the window / the position as raw words at [46], [47]; drain_items works them off in pairs
  uint32_t n_items = 0;
  constexpr uint32_t MAX_ITEMS = 20;          // 48 * 20 <= the 1080 doubles of E
  auto word = [&](const double *q) {          // a word every lane reads from the same LDS address, as a scalar
    const long long v = __double_as_longlong(*q);
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  };
  auto push_item = [&](int p, unsigned long long w, double val, uint32_t ln) {   // S[ln] = val from lanes 0..15
    double *slot = E + 48u * n_items;
    if (ln < CNN_L1) slot[ln] = val;
    if (ln == 0u) {
      slot[46] = __longlong_as_double((long long)w);
      slot[47] = __longlong_as_double((long long)p);
    }
    ++n_items;
  };
  auto drain_items = [&](uint32_t ln) {
    for (uint32_t k = 0; k < n_items; k += 2) {
      const bool two = k + 1 < n_items;
      const double *sa = E + 48u * k, *sb = E + 48u * (two ? k + 1 : k);
      const int pa = (int)word(sa + 47), pb = (int)word(sb + 47);
      const unsigned long long wa = word(sa + 46), wb = word(sb + 46);
      cnn_backward_shared_window(Fs, exptab, E + 48u * k, Ps0, Pb0, params + D.oW1, G, D, pa, wa, pb, wb, two, ln);
    }
    n_items = 0;
  };
  auto flush_carried = [&](int q, uint32_t ln) {
    push_item(q, word(Cy + C::CARRY_POS * CNN_L1 + q), ln < CNN_L1 ? Cy[q * CNN_L1 + (int)ln] : 0.0, ln);
    carry &= ~(1u << q);
  };
  for (uint64_t g = n_groups * wave_id / wave_cnt; g < n_groups * (wave_id + 1) / wave_cnt; ++g) {
    const uint16_t *lst = live_lists ? live_lists + g * PLN_LIVE_STRIDE : nullptr;
    const uint64_t base = lst ? tiles[g].row0 : g * TILE;
    const uint32_t cnt = lst ? (uint32_t)lst[0] : (uint32_t)(n_rows - base < (uint64_t)TILE ? n_rows - base : (uint64_t)TILE);
   uint32_t row_next;                          // the list entry of a tile is read one tile ahead: its rows' loads then start at once
   {
     const uint32_t c = lane & (TILE - 1);
     row_next = lst ? (uint32_t)lst[1 + (c < cnt ? c : 0u)] : c;
   }
   for (uint32_t c0 = 0; c0 < cnt; c0 += TILE) {
    // the lane's coordinates are derived HERE, from a laundered copy of the lane number: hoisted out of the group loop, the
    // addresses built on them took (and spilled) two dozen registers
    uint32_t lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    const uint32_t ctx = lane_t & (TILE - 1);
    const uint32_t h = lane_t / TILE;               // the lane's part: filters [FH h, FH h + FH), layer-1 units [JH h, JH h + JH)
    const uint32_t lq = lane_t >> 4, lr = lane_t & 15u;   // MFMA lane coordinates: k / row-group index, row / column index
    const bool live = c0 + ctx < cnt;
    const uint64_t i = base + (uint64_t)row_next;
    if (lst && c0 + TILE < cnt) row_next = (uint32_t)lst[1 + (c0 + TILE + ctx < cnt ? c0 + TILE + ctx : 0u)];
    // dead lanes: every letter "other" and a zero gradient row -> all their contributions are exact zeros
    unsigned long long code = 0;
#pragma unroll
    for (int l = 0; l < 21; ++l) code |= 5ull << (3 * l);
    if (live) code = codes[i];
    if (h == 0) Cw[ctx] = code;
    double t1[JH], n1[JH], e1[JH], d1[JH], dy1[JH];
#pragma unroll
    for (int j = 0; j < JH; ++j) t1[j] = 0.0;
    double pr[5], gp[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      pr[b] = 0.2;
      gp[b] = 0.0;
    }
    if (live) {
      if (io.head && !t1_save) {
        // the forward pass kept no layer-1 sums: every position of these contexts came from a parent row and window rows
        // (cnn_level_io), and the sums are put together again exactly as the forward kernel did -- two or three gathers that
        // mostly hit the caches instead of a 128-byte row written and read back per context
        if (io.t1_parent) {
          const double2 *src = reinterpret_cast<const double2 *>(io.t1_parent + (size_t)io.parent[i] * CNN_L1 + h * JH);
#pragma unroll
          for (int j = 0; j < JH / 2; ++j) {
            const double2 v = src[j];
            t1[2 * j] = v.x;
            t1[2 * j + 1] = v.y;
          }
        }
#pragma unroll
        for (int q = 0; q < CNN_MAX_WIN; ++q) {
          if (q < io.n_win) {
            const double2 *src = reinterpret_cast<const double2 *>(io.win_rows[q] + (size_t)io.win_row_of[q][i] * CNN_L1 + h * JH);
#pragma unroll
            for (int j = 0; j < JH / 2; ++j) {
              const double2 v = src[j];
              t1[2 * j] += v.x;
              t1[2 * j + 1] += v.y;
            }
          }
        }
      } else {
      const double2 *src = reinterpret_cast<const double2 *>((io.head ? t1_save : io.dT1) + i * CNN_L1 + h * JH);
#pragma unroll
      for (int j = 0; j < JH / 2; ++j) {
        const double2 v = src[j];
        t1[2 * j] = v.x;
        t1[2 * j + 1] = v.y;
      }
      }
      if (io.head) {
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          pr[b] = prior[i * 5 + b];
          gp[b] = grad_prior[i * 5 + b];
        }
      }
    }
    if (io.head) {      // (a level of prefixes: t1 holds the rows' dT1 already -- the sums of their children's)
    // layer 1 (as cnn_layer1, the 16 units over the Q parts)
    double r1;
    {
      double mu = 0.0;
#pragma unroll
      for (int j = 0; j < JH; ++j) mu += t1[j];
      mu = cnnq_psum<Q>(mu) * (1.0 / CNN_L1);
      double var = 0.0;
#pragma unroll
      for (int j = 0; j < JH; ++j) {
        n1[j] = t1[j] - mu;
        var = __builtin_fma(n1[j], n1[j], var);
      }
      r1 = cnn_rsqrt(cnnq_psum<Q>(var) * (1.0 / CNN_L1) + CNN_LN_EPS);
#pragma unroll
      for (int j = 0; j < JH; ++j) {
        n1[j] *= r1;
        e1[j] = cnn_elu(__builtin_fma(Ps1[h * JH + j], n1[j], Pb1[h * JH + j]), exptab, d1[j]);
      }
    }
    // softmax backward
    double dz[5], sg = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) sg = __builtin_fma(pr[b], gp[b], sg);
#pragma unroll
    for (int b = 0; b < 5; ++b) dz[b] = pr[b] * (gp[b] - sg);
    // layer 2 / layer-1 norm backward
    {
      double ma = 0.0, mb = 0.0;
#pragma unroll
      for (int j = 0; j < JH; ++j) {
        double de = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) de = __builtin_fma(PW2[(h * JH + j) * 5 + b], dz[b], de);
        dy1[j] = de * d1[j];
        const double dn = dy1[j] * Ps1[h * JH + j];
        t1[j] = dn;
        ma += dn;
        mb = __builtin_fma(dn, n1[j], mb);
      }
      ma = cnnq_psum<Q>(ma) * (1.0 / CNN_L1);
      mb = cnnq_psum<Q>(mb) * (1.0 / CNN_L1);
#pragma unroll
      for (int j = 0; j < JH; ++j) t1[j] = r1 * (t1[j] - ma - n1[j] * mb);     // dT1
    }
    // d weights2[j][b] = the sum over the tile's contexts of e1[ctx][j] dz[ctx][b]: ONE 16 x 16 accumulation on the matrix core over
    // K = contexts (rows = units, columns = letters: 5 of 16 used) -- e1 and dz staged once (13 LDS writes, 16 reads per lane)
    // instead of three rounds of 32 staged product columns and their column sums (96 LDS operations per lane and tile)
#pragma unroll
    for (int j = 0; j < JH; ++j) E[(h * JH + j) * ES + ctx] = e1[j];
    if (h == 0) {
#pragma unroll
      for (int b = 0; b < 5; ++b) E[(CNN_L1 + b) * ES + ctx] = dz[b];
    }
    {
      cnn_d4 acc = {0.0, 0.0, 0.0, 0.0};
      const bool col_in = lr < 5u;
      const double *arow = E + lr * ES + lq, *brow = E + (CNN_L1 + (col_in ? lr : 0u)) * ES + lq;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const double a = arow[4 * ks], bv = col_in ? brow[4 * ks] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
      }
      if (col_in) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cnn_lds_add(G + D.oW2 + ((int)lq + 4 * r) * 5 + (int)lr, acc[r]);
      }
    }
    {                                      // d intercept2[b]: the column sums of the staged dz rows (lane: letter lane & 31, half of the contexts)
      const uint32_t c = lane_t & 31u;
      if (c < 5u) {
        const double2 *src = reinterpret_cast<const double2 *>(E + (CNN_L1 + c) * ES + (lane_t >> 5) * (TILE / 2));
        double s2[2] = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < TILE / 4; ++k) {
          const double2 v = src[k];
          s2[0] += v.x;
          s2[1] += v.y;
        }
        cnn_lds_add(G + D.ob2 + (int)c, s2[0] + s2[1]);
      }
    }
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      E[(h * JH + j) * CS + ctx] = dy1[j] * n1[j];
      E[(16 + h * JH + j) * CS + ctx] = dy1[j];
    }
    {
      const double cs = cnnq_colsum<Q>(E, lane);
      const uint32_t c = lane & 31u;
      cnn_lds_add(G + (c < 16u ? D.os1 + (int)c : D.ob1 + (int)c - 16), cs);
    }
    if (io.dT1 && live) {     // the contexts' dT1 rows: what the next level's rows are the sums of
      double2 *o = reinterpret_cast<double2 *>(io.dT1 + i * CNN_L1 + h * JH);
#pragma unroll
      for (int j = 0; j < JH / 2; ++j) o[j] = make_double2(t1[2 * j], t1[2 * j + 1]);
    }
    }
    if (io.p_lo >= io.p_hi) continue;      // rows whose positions all come from window tables / the parent level: nothing else to do here
#pragma unroll
    for (int j = 0; j < JH; ++j) T[(h * JH + j) * ES + ctx] = t1[j];
    // positions whose window every context of the tile shares (see cnn_backward_shared_window): handled here, once per tile,
    // from the column sums of dT1; the position loop below skips them
    uint32_t shared = 0;
    int n_common = 0;
    unsigned long long first = 0ull;
    CNN_STAMP(6)
#ifndef CNN_NO_SHARED_BACKWARD
    {
      uint32_t ln = lane;
      asm volatile("" : "+v"(ln));   // as in cnn_backward_shared_window
      static_assert(TILE == 32, "column sums of dT1: four quarters of eight contexts; lanes 0..31 of a ballot are the contexts");
      const unsigned long long wm = (D.fw < 21 ? (1ull << (3 * D.fw)) : 0ull) - 1ull;
      const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
      {   // leading letters that all the tile's contexts share (taps inside them need no one-hot product below)
        first = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(code >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)code);   // lane 0: the tile's first context (always live)
        const unsigned long long diff = live ? code ^ first : 0ull;
        while (n_common < D.lag && __builtin_amdgcn_ballot_w64(((diff >> (3 * n_common)) & 7ull) != 0ull) == 0ull) ++n_common;
      }
      auto next_run = [&](unsigned long long wid, unsigned long long rem, unsigned long long *w) {   // the contexts that share the window of rem's first
        const int leader = __builtin_ctzll(rem);
        *w = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wid >> 32), leader) << 32) |
             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wid, leader);
        return __builtin_amdgcn_ballot_w64(wid == *w) & rem;
      };
      bool have_full = false;
      double s_full = 0.0;
      for (int p = io.p_lo; p < io.p_hi; ++p) {
        const unsigned long long wid = (code >> (3 * p)) & wm;
        unsigned long long rem = live_mask, w;
        for (int runs = 0; rem != 0ull && runs < CNN_SHARED_RUNS; ++runs) rem &= ~next_run(wid, rem, &w);
        if (rem != 0ull) break;               // more distinct windows than pay: the position loop below takes this position and the
                                              // later ones (in a sorted batch their windows reach further into the varying letters)
        shared |= 1u << p;
        for (rem = live_mask; rem != 0ull;) {
          const unsigned long long m = next_run(wid, rem, &w);
          rem &= ~m;
          // S = the run's column sums of dT1 (lane: unit ln & 15, contexts 8 (ln >> 4) .. + 7); a run of all the tile's contexts
          // (a fully shared window: most of them) reuses the tile's sums
          double sq = s_full;
          if (m != live_mask || !have_full) {
            const double *src = T + (ln & 15u) * ES + (ln >> 4) * (TILE / 4);
            const uint32_t mine = (uint32_t)m >> ((ln >> 4) * (TILE / 4));
            sq = 0.0;
#pragma unroll
            for (int k = 0; k < TILE / 4; ++k) sq += ((mine >> k) & 1u) ? src[k] : 0.0;
            sq = cnn_rows_sum(sq);
            if (m == live_mask) {
              s_full = sq;
              have_full = true;
            }
          }
          if (m == live_mask && p < C::CARRY_POS) {   // fully shared, leading position: carried while the next tiles share it too
            double *cs = Cy + p * CNN_L1;
            if (((carry >> p) & 1u) && word(Cy + C::CARRY_POS * CNN_L1 + p) != w) flush_carried(p, ln);
            if ((carry >> p) & 1u) {
              if (ln < CNN_L1) cs[ln] += sq;
            } else {
              if (ln < CNN_L1) cs[ln] = sq;
              if (ln == 0u) Cy[C::CARRY_POS * CNN_L1 + p] = __longlong_as_double((long long)w);
              carry |= 1u << p;
            }
          } else {
            push_item(p, w, sq, ln);
          }
          if (n_items + C::CARRY_POS >= MAX_ITEMS) drain_items(ln);
        }
      }
      // a carried window that this tile does not share (any more, or in several runs) stays carried: its sums are complete
      // whenever it is flushed.  The queue is worked off here: E is the position loop's staging area.
      if (n_items) drain_items(ln);
    }
#endif
    CNN_STAMP(7)
    double tb[KS], tb2[4][NT];     // dT1 as the B operand of d weights1 (K = contexts) and of d e0 (K = j), the same for every position
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) tb[ks] = T[lr * ES + 4 * ks + lq];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) tb2[ks][nt] = T[(4 * ks + lq) * ES + nt * 16 + lr];
    CNN_STAMP(6)
    // positions.  Their lane coordinates come from the lane number itself again (not the laundered copy of the head above):
    // what the loop builds on them is loop-invariant, stays in registers across the tiles, and pairs of stores keep merging
    {
    const uint32_t ctx = lane & (TILE - 1), h = lane / TILE, lq = lane >> 4, lr = lane & 15u;
    const bool last = h == Q - 1;                   // the part whose last two filter slots are dummies (30 filters)
    for (int p = io.p_lo; p < io.p_hi; ++p) {
#ifndef CNN_NO_SKIP
      if ((shared >> p) & 1u) continue;
#endif
      const int w0 = n_common - p < 0 ? 0 : (n_common - p > D.fw ? D.fw : n_common - p);   // leading taps inside the tile's common prefix
      double x[FH], dy[FH], dn[FH];
      // conv row of the lane's filters, layer norm over all 30 (the two dummy slots of the last part stay exact zeros)
      double r0;
      {
#pragma unroll
        for (int f = 0; f < FH; ++f) x[f] = 0.0;
        unsigned long long c = code >> (3 * p);
        for (int w = 0; w < D.fw; ++w) {
          const int a = (int)(c & 7ull);
          const double2 *row = reinterpret_cast<const double2 *>(Fs + (w * 6 + (a < 5 ? a : 5)) * CNN_NF + h * FH);
          c >>= 3;
          double2 v[FH / 2];                // slots 30, 31 of the last part: the next row's first words, discarded below
#pragma unroll
          for (int f2 = 0; f2 < FH / 2; ++f2) v[f2] = row[f2];
#pragma unroll
          for (int f2 = 0; f2 < FH / 2; ++f2) {
            x[2 * f2] += v[f2].x;
            x[2 * f2 + 1] += v[f2].y;
          }
          // all the reads of a tap in flight, then its adds (left to itself the scheduler sometimes waits for every read in turn:
          // 5 % of the kernel)
          __builtin_amdgcn_sched_group_barrier(0x100, FH / 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, FH, 0);
        }
        if (last) {
          x[FH - 2] = 0.0;
          x[FH - 1] = 0.0;
        }
        double m4[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
        for (int f = 4; f < FH; ++f) m4[f & 3] += x[f];
        const double mu = cnnq_psum<Q>((m4[0] + m4[1]) + (m4[2] + m4[3])) * (1.0 / CNN_NF);
        double v4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int f = 0; f < FH; ++f) {
          x[f] -= mu;
          if (f >= FH - 2 && last) x[f] = 0.0;
          v4[f & 3] = __builtin_fma(x[f], x[f], v4[f & 3]);
        }
        r0 = cnn_rsqrt(cnnq_psum<Q>((v4[0] + v4[1]) + (v4[2] + v4[3])) * (1.0 / CNN_NF) + CNN_LN_EPS);
#pragma unroll
        for (int f = 0; f < FH; ++f) x[f] *= r0;
      }
      // the lane's scales / intercepts of this position (the dummies of the last part read two words past the row: unused)
      const double *s0 = Ps0 + p * CNN_NF + h * FH, *b0 = Pb0 + p * CNN_NF + h * FH;
      const double *__restrict__ W1 = params + D.oW1 + p * CNN_NF * CNN_L1;
      // pass A: activations e0 (staged as the A operand of d weights1[p]) and elu'; dy holds elu' until pass B
#pragma unroll
      for (int f = 0; f < FH; ++f) {
        double dv;
        const double e = cnn_elu(__builtin_fma(s0[f], x[f], b0[f]), exptab, dv);
        if (f < FH - 2 || !last) E[(h * FH + f) * ES + ctx] = e;
        dy[f] = (f >= FH - 2 && last) ? 0.0 : dv;
      }
      CNN_STAMP(0)
      // d weights1[p][f][j] += sum_ctx e0[ctx][f] dT1[ctx][j]
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        cnn_d4 acc = {0.0, 0.0, 0.0, 0.0};
        const bool row_in = mt * 16 + (int)lr < CNN_NF;
        const double *arow = E + (row_in ? mt * 16 + (int)lr : 0) * ES + lq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const double a = row_in ? arow[4 * ks] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tb[ks], acc, 0, 0, 0);
        }
        double *g = G + D.oW1 + p * CNN_NF * CNN_L1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = mt * 16 + (int)lq + 4 * r;
          if (f < CNN_NF) cnn_lds_add(g + f * CNN_L1 + lr, acc[r]);
        }
      }
      CNN_STAMP(1)
      // pass B: d e0[f][ctx] = sum_j weights1[p][f][j] dT1[ctx][j] as MFMA products (rows f, columns ctx, K = j), the
      // result handed back to the (context, part) lanes through the staging buffer
      {
        double wa[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int f = mt * 16 + (int)lr;
            wa[mt][ks] = f < CNN_NF ? W1[f * CNN_L1 + 4 * ks + (int)lq] : 0.0;
          }
        cnn_d4 c[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) c[mt][nt] = cnn_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) c[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[mt][ks], tb2[ks][nt], c[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = mt * 16 + (int)lq + 4 * r;
            if (f < CNN_NF) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) E[f * ES + nt * 16 + lr] = c[mt][nt][r];
            }
          }
      }
      CNN_STAMP(2)
      double a0[2] = {0.0, 0.0}, a1[2] = {0.0, 0.0};
#pragma unroll
      for (int f = 0; f < FH; ++f) {
        const double de0 = (f < FH - 2 || !last) ? E[(h * FH + f) * ES + ctx] : 0.0;
        dy[f] *= de0;
        dn[f] = dy[f] * s0[f];
        a0[f & 1] += dn[f];
        a1[f & 1] = __builtin_fma(dn[f], x[f], a1[f & 1]);
      }
      const double ma0 = cnnq_psum<Q>(a0[0] + a0[1]) * (1.0 / CNN_NF), ma1 = cnnq_psum<Q>(a1[0] + a1[1]) * (1.0 / CNN_NF);
      // d scale0[p], d intercept0[p]: column sums of dy * n0 and dy (staging columns = filter slots; 30, 31 carry zeros)
#pragma unroll
      for (int f = 0; f < FH; ++f) E[(h * FH + f) * CS + ctx] = dy[f] * x[f];
      {
        const double cs = cnnq_colsum<Q>(E, lane);
        if ((lane & 31u) < (uint32_t)CNN_NF) cnn_lds_add(G + D.os0 + p * CNN_NF + (int)(lane & 31u), cs);
      }
#pragma unroll
      for (int f = 0; f < FH; ++f) E[(h * FH + f) * CS + ctx] = dy[f];
      {
        const double cs = cnnq_colsum<Q>(E, lane);
        if ((lane & 31u) < (uint32_t)CNN_NF) cnn_lds_add(G + D.ob0 + p * CNN_NF + (int)(lane & 31u), cs);
      }
      CNN_STAMP(3)
      // layer-norm backward -> d conv[p][f], staged as the B operand of d filters
      {
        unsigned long long c = code >> (3 * p);
        bool any_start = false;
        for (int w = w0; w < D.fw; ++w) any_start |= ((c >> (3 * w)) & 7ull) == 4ull;
#pragma unroll
        for (int f = 0; f < FH; ++f) {
          const double dc = (f >= FH - 2 && last) ? 0.0 : r0 * (dn[f] - ma0 - x[f] * ma1);
          if (f < FH - 2 || !last) E[(h * FH + f) * ES + ctx] = dc;
          dy[f] = dc;
        }
        if (w0 > 0) {      // taps inside the tile's common prefix: every context has the same letter there -- its filter row takes the
                           // column sums of d conv (lane: feature lane & 31, half of the contexts), no one-hot rows
          const uint32_t cf = lane & 31u;
          const double2 *src = reinterpret_cast<const double2 *>(E + (cf < CNN_NF ? cf : 0u) * ES + (lane >> 5) * (TILE / 2));
          double s2[2] = {0.0, 0.0};
#pragma unroll
          for (int k = 0; k < TILE / 4; ++k) {
            const double2 v = src[k];
            s2[0] += v.x;
            s2[1] += v.y;
          }
          const double tot = cnnq_psum<Q>(s2[0] + s2[1]);
          unsigned long long cf1 = first >> (3 * p);
          for (int w = 0; w < w0; ++w) {
            const int a = (int)(cf1 & 7ull);
            if (a < 5 && lane < CNN_NF) cnn_lds_add(G + D.oF + (w * 5 + a) * CNN_NF + (int)lane, tot);
            cf1 >>= 3;
          }
        }
        if (any_start) {   // the start symbol '[' (rare: only contexts at a sequence start) bypasses the MFMA rows
          for (int w = w0; w < D.fw; ++w)
            if (((c >> (3 * w)) & 7ull) == 4ull) {
              double *gF = G + D.oF + (w * 5 + 4) * CNN_NF + h * FH;
#pragma unroll
              for (int f = 0; f < FH; ++f)
                if (f < FH - 2 || !last) cnn_lds_add(gF + f, dy[f]);
            }
        }
      }
      CNN_STAMP(4)
      // d filters[w][a][f] += sum_ctx [letter_{p+w}(ctx) == a] d conv[ctx][f],  rows (w - w0, a < 4) over the taps w >= w0 that
      // vary inside the tile, two column tiles of f; NU row tiles per pass share the B operand reads
      auto d_filters = [&](auto nu_tag, int mt0) {
        constexpr int NU = decltype(nu_tag)::value;
        cnn_d4 acc[NU][2];
        int sh[NU];
        unsigned long long want[NU];
        bool row_ok[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          acc[u][0] = cnn_d4{0.0, 0.0, 0.0, 0.0};
          acc[u][1] = cnn_d4{0.0, 0.0, 0.0, 0.0};
          const int row = (mt0 + u) * 16 + (int)lr, w = w0 + (row >> 2);
          want[u] = (unsigned long long)(row & 3);
          row_ok[u] = w < D.fw;
          sh[u] = row_ok[u] ? 3 * (p + w) : 0;
        }
        const bool col1_in = 16 + (int)lr < CNN_NF;
        const double *b1row = E + (col1_in ? 16 + (int)lr : 0) * ES + lq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const unsigned long long cc = Cw[4 * ks + lq];
          const double b0v = E[lr * ES + 4 * ks + lq], b1v = col1_in ? b1row[4 * ks] : 0.0;
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const double a = (row_ok[u] && ((cc >> sh[u]) & 7ull) == want[u]) ? 1.0 : 0.0;
            acc[u][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0v, acc[u][0], 0, 0, 0);
            acc[u][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1v, acc[u][1], 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int orow = (mt0 + u) * 16 + (int)lq + 4 * r, ow = w0 + (orow >> 2), oa = orow & 3;
            if (ow < D.fw) {
              double *gF = G + D.oF + (ow * 5 + oa) * CNN_NF;
              cnn_lds_add(gF + lr, acc[u][0][r]);
              if (lr < CNN_NF - 16) cnn_lds_add(gF + 16 + lr, acc[u][1][r]);
            }
          }
      };
      {
        const int n_mt_var = (4 * (D.fw - w0) + 15) / 16;
        int mt = 0;
        for (; mt + 1 < n_mt_var; mt += 2) d_filters(std::integral_constant<int, 2>{}, mt);
        if (mt < n_mt_var) d_filters(std::integral_constant<int, 1>{}, mt);
      }
      CNN_STAMP(5)
    }
    }
   }
  }
#ifndef CNN_NO_SHARED_BACKWARD
  {
    uint32_t ln = lane;
    asm volatile("" : "+v"(ln));
    for (int q = 0; q < C::CARRY_POS; ++q)
      if ((carry >> q) & 1u) flush_carried(q, ln);
    if (n_items) drain_items(ln);
  }
#endif
#ifdef CNN_STAMPS
  if (lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&cnn_stamp_sums[k], tph[k]);
#endif
  __syncthreads();
  for (int k = threadIdx.x; k < D.total; k += blockDim.x) {
    double *dst = partials + (size_t)blockIdx.x * D.total + k;
    *dst = io.accumulate ? *dst + G[k] : G[k];       // (launches of one step are stream-ordered: block b owns row b of the buffer)
  }
}

// ------------------------------------------------------------------------------------------------ backward, head only
// The contexts of a step whose positions ALL come from their parent level and window tables (bear_plan_attach_cnn_levels on a
// table dense enough: every reference-sized batch) need nothing from cnn_backward_parts_kernel but its head: layer 1, softmax and
// layer-2 backward, their dT1 row, and the 117 gradients of layer 2 / layer 1's norm.  A kernel of its own for that: no position
// loop, hence a third of the registers (three waves per SIMD instead of two), the gradient sums in REGISTERS across a wave's
// tiles (d weights2 in the accumulator of one f64 MFMA per four contexts, the others per lane) instead of staging rounds and LDS
// atomics per tile, and one fixed-order sum over lanes and waves at the end -- no atomics at all: reproducible bit for bit.
// Same lane layout as the part kernel: a wave tile = 32 contexts x 2 halves, half h owns layer-1 units [8 h, 8 h + 8).
#define CNH_WAVES 12
#define CNH_NG (CNN_L1 * 5 + 5 + 2 * CNN_L1)      // d weights2 [16][5] | d intercept2 [5] | d scale1 [16] | d intercept1 [16]
#define CNH_ES 36
#define CNH_WAVE_DOUBLES ((CNN_L1 + 5) * CNH_ES)
static inline size_t cnh_lds_bytes() { return sizeof(double) * (BEAR_EXPTAB_N + 2 * CNN_L1 + CNN_L1 * 5 + (size_t)CNH_WAVES * (CNH_WAVE_DOUBLES + CNH_NG)); }

__global__ __launch_bounds__(CNH_WAVES * 64) void cnn_backward_head_kernel(uint64_t n_rows, cnn_dims D, const double *__restrict__ params,
                                                                           const double *__restrict__ t1_save, const double *__restrict__ prior,
                                                                           const double *__restrict__ grad_prior, double *__restrict__ partials,
                                                                           const cnn_level_io io) {
  constexpr int TILE = 32, JH = CNN_L1 / 2, ES = CNH_ES, KS = TILE / 4;
  extern __shared__ __attribute__((aligned(16))) double cnn_lds[];
  double *exptab = cnn_lds, *Ps1 = exptab + BEAR_EXPTAB_N, *Pb1 = Ps1 + CNN_L1, *PW2 = Pb1 + CNN_L1;
  const uint32_t lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double *E = PW2 + CNN_L1 * 5 + wave * CNH_WAVE_DOUBLES;                 // [16 + 5][ES]: e1 rows, dz rows
  double *GW = PW2 + CNN_L1 * 5 + n_waves * CNH_WAVE_DOUBLES + wave * CNH_NG;   // this wave's sums
  for (int k = threadIdx.x; k < BEAR_EXPTAB_N; k += blockDim.x) exptab[k] = exp2((double)k * (1.0 / BEAR_EXPTAB_N));
  for (int k = threadIdx.x; k < CNN_L1; k += blockDim.x) {
    Ps1[k] = params[D.os1 + k];
    Pb1[k] = params[D.ob1 + k];
  }
  for (int k = threadIdx.x; k < CNN_L1 * 5; k += blockDim.x) PW2[k] = params[D.oW2 + k];
  __syncthreads();
  const uint32_t ctx = lane & (TILE - 1), h = lane / TILE, lq = lane >> 4, lr = lane & 15u;
  cnn_d4 acc_w2 = {0.0, 0.0, 0.0, 0.0};
  double acc_s1[JH], acc_b1[JH], acc_b2[5];
#pragma unroll
  for (int j = 0; j < JH; ++j) acc_s1[j] = acc_b1[j] = 0.0;
#pragma unroll
  for (int b = 0; b < 5; ++b) acc_b2[b] = 0.0;
  const uint64_t n_tiles = (n_rows + TILE - 1) / TILE;
  const uint64_t wave_id = (uint64_t)blockIdx.x * n_waves + wave, wave_cnt = (uint64_t)gridDim.x * n_waves;
  for (uint64_t g = n_tiles * wave_id / wave_cnt; g < n_tiles * (wave_id + 1) / wave_cnt; ++g) {
    const uint64_t i = g * TILE + ctx;
    const bool live = i < n_rows;
    double t1[JH], n1[JH], e1[JH], dy1[JH];
#pragma unroll
    for (int j = 0; j < JH; ++j) t1[j] = 0.0;
    double pr[5], gp[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      pr[b] = 0.2;
      gp[b] = 0.0;
    }
    if (live) {
      if (!t1_save) {      // the layer-1 sums put together again from the rows the forward pass took them from (cnn_level_io)
        if (io.t1_parent) {
          const double2 *src = reinterpret_cast<const double2 *>(io.t1_parent + (size_t)io.parent[i] * CNN_L1 + h * JH);
#pragma unroll
          for (int j = 0; j < JH / 2; ++j) {
            const double2 v = src[j];
            t1[2 * j] = v.x;
            t1[2 * j + 1] = v.y;
          }
        }
#pragma unroll
        for (int q = 0; q < CNN_MAX_WIN; ++q) {
          if (q < io.n_win) {
            const double2 *src = reinterpret_cast<const double2 *>(io.win_rows[q] + (size_t)io.win_row_of[q][i] * CNN_L1 + h * JH);
#pragma unroll
            for (int j = 0; j < JH / 2; ++j) {
              const double2 v = src[j];
              t1[2 * j] += v.x;
              t1[2 * j + 1] += v.y;
            }
          }
        }
      } else {
        const double2 *src = reinterpret_cast<const double2 *>(t1_save + i * CNN_L1 + h * JH);
#pragma unroll
        for (int j = 0; j < JH / 2; ++j) {
          const double2 v = src[j];
          t1[2 * j] = v.x;
          t1[2 * j + 1] = v.y;
        }
      }
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        pr[b] = prior[i * 5 + b];
        gp[b] = grad_prior[i * 5 + b];
      }
    }
    // layer 1 forward again (norm over the 16 units of the two halves, elu), as the part kernel
    double r1;
    {
      double mu = 0.0;
#pragma unroll
      for (int j = 0; j < JH; ++j) mu += t1[j];
      mu = cnnq_psum<2>(mu) * (1.0 / CNN_L1);
      double var = 0.0;
#pragma unroll
      for (int j = 0; j < JH; ++j) {
        n1[j] = t1[j] - mu;
        var = __builtin_fma(n1[j], n1[j], var);
      }
      r1 = cnn_rsqrt(cnnq_psum<2>(var) * (1.0 / CNN_L1) + CNN_LN_EPS);
    }
    double dz[5], sg = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) sg = __builtin_fma(pr[b], gp[b], sg);
#pragma unroll
    for (int b = 0; b < 5; ++b) dz[b] = pr[b] * (gp[b] - sg);
    {
      double ma = 0.0, mb = 0.0;
#pragma unroll
      for (int j = 0; j < JH; ++j) {
        n1[j] *= r1;
        double d1;
        e1[j] = cnn_elu(__builtin_fma(Ps1[h * JH + j], n1[j], Pb1[h * JH + j]), exptab, d1);
        double de = 0.0;
#pragma unroll
        for (int b = 0; b < 5; ++b) de = __builtin_fma(PW2[(h * JH + j) * 5 + b], dz[b], de);
        dy1[j] = de * d1;
        const double dn = dy1[j] * Ps1[h * JH + j];
        t1[j] = dn;
        ma += dn;
        mb = __builtin_fma(dn, n1[j], mb);
        acc_s1[j] = __builtin_fma(dy1[j], n1[j], acc_s1[j]);       // d scale1, d intercept1: per lane, summed over lanes at the end
        acc_b1[j] += dy1[j];
      }
      ma = cnnq_psum<2>(ma) * (1.0 / CNN_L1);
      mb = cnnq_psum<2>(mb) * (1.0 / CNN_L1);
#pragma unroll
      for (int j = 0; j < JH; ++j) t1[j] = r1 * (t1[j] - ma - n1[j] * mb);     // dT1
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) acc_b2[b] += dz[b];                 // (both halves hold the context's dz: half 0 is counted at the end)
    // d weights2 += e1 (x) dz over the tile's contexts: the MFMA accumulator carries the sum across the wave's tiles
#pragma unroll
    for (int j = 0; j < JH; ++j) E[(h * JH + j) * ES + ctx] = e1[j];
    if (h == 0) {
#pragma unroll
      for (int b = 0; b < 5; ++b) E[(CNN_L1 + b) * ES + ctx] = dz[b];
    }
    {
      const bool col_in = lr < 5u;
      const double *arow = E + lr * ES + lq, *brow = E + (CNN_L1 + (col_in ? lr : 0u)) * ES + lq;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const double a = arow[4 * ks], bv = col_in ? brow[4 * ks] : 0.0;
        acc_w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc_w2, 0, 0, 0);
      }
    }
    if (io.dT1 && live) {
      double2 *o = reinterpret_cast<double2 *>(io.dT1 + i * CNN_L1 + h * JH);
#pragma unroll
      for (int j = 0; j < JH / 2; ++j) o[j] = make_double2(t1[2 * j], t1[2 * j + 1]);
    }
  }
  // this wave's sums into its LDS slot: sums over the 32 contexts of a half in a fixed butterfly, every entry written by one lane
  if (lr < 5u) {
#pragma unroll
    for (int r = 0; r < 4; ++r) GW[((int)lq + 4 * r) * 5 + (int)lr] = acc_w2[r];
  }
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    const double v = cnn_half_sum_all(acc_b2[b]);
    if (lane == 0u) GW[CNN_L1 * 5 + b] = v;
  }
#pragma unroll
  for (int j = 0; j < JH; ++j) {
    const double vs = cnn_half_sum_all(acc_s1[j]), vb = cnn_half_sum_all(acc_b1[j]);
    if (ctx == 0u) {
      GW[CNN_L1 * 5 + 5 + h * JH + j] = vs;
      GW[CNN_L1 * 5 + 5 + CNN_L1 + h * JH + j] = vb;
    }
  }
  __syncthreads();
  // the block's row of the partial buffer: the waves' sums in wave order; everything else of the row is zero (first launch of a
  // step) or untouched (a later one)
  const double *G0 = PW2 + CNN_L1 * 5 + n_waves * CNH_WAVE_DOUBLES;
  double *row = partials + (size_t)blockIdx.x * D.total;
  if (!io.accumulate)
    for (int k = threadIdx.x; k < D.total; k += blockDim.x) row[k] = 0.0;
  __syncthreads();
  for (int k = threadIdx.x; k < CNH_NG; k += blockDim.x) {
    double v = 0.0;
    for (uint32_t w = 0; w < n_waves; ++w) v += G0[w * CNH_NG + k];
    const int dst = k < CNN_L1 * 5 ? D.oW2 + k : k < CNN_L1 * 5 + 5 ? D.ob2 + (k - CNN_L1 * 5)
                  : k < CNN_L1 * 5 + 5 + CNN_L1 ? D.os1 + (k - CNN_L1 * 5 - 5) : D.ob1 + (k - CNN_L1 * 5 - 5 - CNN_L1);
    row[dst] = io.accumulate ? row[dst] + v : v;
  }
}

// ------------------------------------------------------------------------------------------------ forward, head only
// The forward counterpart of cnn_backward_head_kernel: contexts whose layer-1 sums come from their parent level and window tables
// alone -- two or three 128-byte gathers, layer 1 (norm, elu), layer 2, softmax, the prior row.  Two lanes per context as in the
// backward kernels (a lane gathers and normalises 8 of the 16 units; the letters' logits meet through v_permlane32_swap).
__global__ __launch_bounds__(1024) void cnn_forward_head_kernel(uint64_t n_rows, cnn_dims D, const double *__restrict__ params,
                                                                double *__restrict__ prior, double *__restrict__ t1_save, const cnn_level_io io) {
  constexpr int TILE = 32, JH = CNN_L1 / 2;
  __shared__ double exptab[BEAR_EXPTAB_N], Ps1[CNN_L1], Pb1[CNN_L1], PW2[CNN_L1 * 5], Pb2[5];
  for (int k = threadIdx.x; k < BEAR_EXPTAB_N; k += blockDim.x) exptab[k] = exp2((double)k * (1.0 / BEAR_EXPTAB_N));
  for (int k = threadIdx.x; k < CNN_L1; k += blockDim.x) {
    Ps1[k] = params[D.os1 + k];
    Pb1[k] = params[D.ob1 + k];
  }
  for (int k = threadIdx.x; k < CNN_L1 * 5; k += blockDim.x) PW2[k] = params[D.oW2 + k];
  if (threadIdx.x < 5) Pb2[threadIdx.x] = params[D.ob2 + threadIdx.x];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, n_waves = blockDim.x >> 6;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t ctx = lane & (TILE - 1), h = lane / TILE;
  const uint64_t n_tiles = (n_rows + TILE - 1) / TILE;
  const uint64_t wave_id = (uint64_t)blockIdx.x * n_waves + wave, wave_cnt = (uint64_t)gridDim.x * n_waves;
  for (uint64_t g = n_tiles * wave_id / wave_cnt; g < n_tiles * (wave_id + 1) / wave_cnt; ++g) {
    const uint64_t i = g * TILE + ctx;
    const bool live = i < n_rows;
    double t1[JH];
#pragma unroll
    for (int j = 0; j < JH; ++j) t1[j] = 0.0;
    if (live) {
      if (io.t1_parent) {
        const double2 *src = reinterpret_cast<const double2 *>(io.t1_parent + (size_t)io.parent[i] * CNN_L1 + h * JH);
#pragma unroll
        for (int j = 0; j < JH / 2; ++j) {
          const double2 v = src[j];
          t1[2 * j] = v.x;
          t1[2 * j + 1] = v.y;
        }
      }
#pragma unroll
      for (int q = 0; q < CNN_MAX_WIN; ++q) {
        if (q < io.n_win) {
          const double2 *src = reinterpret_cast<const double2 *>(io.win_rows[q] + (size_t)io.win_row_of[q][i] * CNN_L1 + h * JH);
#pragma unroll
          for (int j = 0; j < JH / 2; ++j) {
            const double2 v = src[j];
            t1[2 * j] += v.x;
            t1[2 * j + 1] += v.y;
          }
        }
      }
      if (t1_save) {
        double2 *o = reinterpret_cast<double2 *>(t1_save + i * CNN_L1 + h * JH);
#pragma unroll
        for (int j = 0; j < JH / 2; ++j) o[j] = make_double2(t1[2 * j], t1[2 * j + 1]);
      }
    }
    double mu = 0.0;
#pragma unroll
    for (int j = 0; j < JH; ++j) mu += t1[j];
    mu = cnnq_psum<2>(mu) * (1.0 / CNN_L1);
    double var = 0.0;
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      t1[j] -= mu;
      var = __builtin_fma(t1[j], t1[j], var);
    }
    const double r1 = cnn_rsqrt(cnnq_psum<2>(var) * (1.0 / CNN_L1) + CNN_LN_EPS);
    double z[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
    for (int j = 0; j < JH; ++j) {      // (two units at a time: all eight exponentials in flight took 190 registers)
      double dv;
      const double e = cnn_elu(__builtin_fma(Ps1[h * JH + j], t1[j] * r1, Pb1[h * JH + j]), exptab, dv);
#pragma unroll
      for (int b = 0; b < 5; ++b) z[b] = __builtin_fma(e, PW2[(h * JH + j) * 5 + b], z[b]);
    }
    double m = -INFINITY;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      z[b] = cnnq_psum<2>(z[b]) + Pb2[b];
      m = z[b] > m ? z[b] : m;
    }
    double tot = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      z[b] = bear_exp_tab(z[b] - m, exptab);
      tot += z[b];
    }
    const double rt = bear_rcp(tot);
    if (live && h == 0) {
#pragma unroll
      for (int b = 0; b < 5; ++b) prior[i * 5 + b] = z[b] * rt;
    }
  }
}

// a level's dT1 rows from its children's (the children of a row are neighbours): one thread per (row, PAIR of units) -- 16-byte
// loads: the kernel is a stream of 128-byte rows bound by the bytes its waves keep in flight (round 5: 8-byte loads ran at 4 TB/s)
__global__ __launch_bounds__(256) void cnn_level_sum_kernel(const double *__restrict__ child_rows, const uint32_t *__restrict__ child_start,
                                                            uint64_t n_rows, double *__restrict__ rows) {
  static_assert(CNN_L1 % 2 == 0, "rows of double2");
  constexpr uint32_t H = CNN_L1 / 2;
  const double2 *__restrict__ src = reinterpret_cast<const double2 *>(child_rows);
  double2 *__restrict__ dst = reinterpret_cast<double2 *>(rows);
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < n_rows * H; t += (uint64_t)gridDim.x * 256) {
    const uint64_t u = t / H;
    const uint32_t j = (uint32_t)(t - u * H);
    const uint32_t c0 = child_start[u], c1 = child_start[u + 1];
    double2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
    uint32_t c = c0;
    for (; c + 1 < c1; c += 2) {
      const double2 a = src[(size_t)c * H + j], b = src[(size_t)(c + 1) * H + j];
      s0.x += a.x;
      s0.y += a.y;
      s1.x += b.x;
      s1.y += b.y;
    }
    if (c < c1) {
      const double2 a = src[(size_t)c * H + j];
      s0.x += a.x;
      s0.y += a.y;
    }
    dst[t] = make_double2(s0.x + s1.x, s0.y + s1.y);
  }
}

// a window's dT1 row from its contexts' rows (bear_window_dev: the contexts of window w are perm[child_start[w] .. child_start[w + 1]),
// anywhere in the batch): one wave per window, eight contexts x 8 pairs of units per load instruction (a context's row is 128
// contiguous bytes, 16 per lane), four such loads in flight; fixed order, no atomics
__global__ __launch_bounds__(256) void cnn_window_sum_kernel(const double *__restrict__ child_rows, const uint32_t *__restrict__ perm,
                                                             const uint32_t *__restrict__ child_start, uint64_t n_windows,
                                                             double *__restrict__ rows) {
  constexpr uint32_t H = CNN_L1 / 2;
  static_assert(H == 8, "eight lanes per row");
  const double2 *__restrict__ src = reinterpret_cast<const double2 *>(child_rows);
  const uint32_t lane = threadIdx.x & 63u, j = lane & 7u, slot = lane >> 3;
  const uint64_t wave_id = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * 256) >> 6;
  for (uint64_t w = wave_id; w < n_windows; w += n_waves) {
    const uint32_t c0 = child_start[w], c1 = child_start[w + 1];
    double2 s4[4] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    uint32_t c = c0 + slot;
    for (; c + 24 < c1; c += 32) {
      uint32_t r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = perm[c + 8 * k];
      double2 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = src[(size_t)r[k] * H + j];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s4[k].x += v[k].x;
        s4[k].y += v[k].y;
      }
    }
    for (; c < c1; c += 8) {
      const double2 v = src[(size_t)perm[c] * H + j];
      s4[0].x += v.x;
      s4[0].y += v.y;
    }
    double vx = (s4[0].x + s4[1].x) + (s4[2].x + s4[3].x), vy = (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y);
    vx += cnn_dpp<0x128>(vx);                   // row_ror:8: the two slots of a row of 16 lanes
    vy += cnn_dpp<0x128>(vy);
    vx = cnn_rows_sum(vx);                      // ... and the four rows
    vy = cnn_rows_sum(vy);
    if (slot == 0) reinterpret_cast<double2 *>(rows)[w * H + j] = make_double2(vx, vy);
  }
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

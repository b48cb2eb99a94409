// bear_math.h -- fp64 device math for the Dirichlet-multinomial hot path (gfx950).
//
// gfx950 has no fp64 transcendental hardware beyond v_rcp_f64 / v_rsq_f64 seeds, so the
// lgamma / digamma differences of bear_model/core.py:73-74 (TFP lbeta -> tf.math.lgamma,
// autodiff -> digamma) are re-derived for integer counts c >= 1:
//
//   D(x, c) = lgamma(x + c) - lgamma(x) = sum_{j<c} log(x + j)      (log rising factorial)
//   P(x, c) = psi(x + c)    - psi(x)    = sum_{j<c} 1 / (x + j)
//
//   c <= BEAR_KPROD : p = prod_j (x + j), p' = dp/dx by the product rule (2 FMA-class ops
//                     per factor); D = log p, P = p'/p -- one log + one reciprocal per item.
//   c >  BEAR_KPROD : shift x up to y = x + m >= BEAR_TSTIR with the same product, then the
//                     Stirling series difference between y and y + (c - m), written with
//                     log1p(c'/y) so that no large terms cancel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BEAR_KPROD 16u   // longest run evaluated as a plain product
#define BEAR_TSTIR 8.0   // smallest argument handed to the Stirling series

struct bear_dp {
  double D;  // lgamma(x+c) - lgamma(x)
  double P;  // psi(x+c) - psi(x)
};

// 1/x to ~1 ulp: v_rcp_f64 seed + two Newton steps (no IEEE division fix-up sequence).
__device__ __forceinline__ double bear_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

__device__ __forceinline__ double bear_log(double x) { return log(x); }

// log1p for t >= 0 of any magnitude, accurate when t is tiny (u-correction form).
__device__ __forceinline__ double bear_log1p_pos(double t) {
  double u = 1.0 + t;
  double d = t - (u - 1.0);  // rounding remainder of 1 + t
  return bear_log(u) + d * bear_rcp(u);
}

// Stirling tail of lgamma:  lgamma(y) = (y - 1/2) log y - y + log(2 pi)/2 + bear_stir_lg(1/y)
__device__ __forceinline__ double bear_stir_lg(double r) {
  double r2 = r * r;
  double s = 1.0 / 156.0;
  s = __builtin_fma(s, r2, -691.0 / 360360.0);
  s = __builtin_fma(s, r2, 1.0 / 1188.0);
  s = __builtin_fma(s, r2, -1.0 / 1680.0);
  s = __builtin_fma(s, r2, 1.0 / 1260.0);
  s = __builtin_fma(s, r2, -1.0 / 360.0);
  s = __builtin_fma(s, r2, 1.0 / 12.0);
  return s * r;
}

// Stirling tail of digamma:  psi(y) = log y - 1/(2y) - bear_stir_psi(1/y)
__device__ __forceinline__ double bear_stir_psi(double r) {
  double r2 = r * r;
  double s = 1.0 / 12.0;
  s = __builtin_fma(s, r2, -691.0 / 32760.0);
  s = __builtin_fma(s, r2, 1.0 / 132.0);
  s = __builtin_fma(s, r2, -1.0 / 240.0);
  s = __builtin_fma(s, r2, 1.0 / 252.0);
  s = __builtin_fma(s, r2, -1.0 / 120.0);
  s = __builtin_fma(s, r2, 1.0 / 12.0);
  return s * r2;
}

// D, P between y and y + c for y >= BEAR_TSTIR, c >= 1 (c as double, exact integer).
__device__ __forceinline__ bear_dp bear_stirling_diff(double y, double c) {
  double y1 = y + c;
  double ry = bear_rcp(y), ry1 = bear_rcp(y1);
  double l1p = bear_log1p_pos(c * ry);  // log(y1 / y)
  double ly = bear_log(y);
  bear_dp o;
  // (y1-1/2) log y1 - (y-1/2) log y - c  ==  (y1-1/2) log(y1/y) + c (log y - 1)
  o.D = __builtin_fma(y1 - 0.5, l1p, c * (ly - 1.0)) + (bear_stir_lg(ry1) - bear_stir_lg(ry));
  o.P = l1p - 0.5 * (ry1 - ry) - (bear_stir_psi(ry1) - bear_stir_psi(ry));
  return o;
}

// General item (x > 0, c >= 1; c is an exact integer carried as a double so that row totals
// beyond the uint32 range -- five columns at KMC's counter limit -- stay exact).
__device__ __forceinline__ bear_dp bear_dm_item(double x, double c) {
  uint32_t m;  // factors taken by the product
  if (c > (double)BEAR_KPROD || x > 0x1p60) {
    // shift so that x + m >= TSTIR (m = 0 when x is already large)
    double need = BEAR_TSTIR - x;
    m = need > 0.0 ? (uint32_t)ceil(need) : 0u;  // <= 8 < c
  } else {
    m = (uint32_t)c;
  }
  double p = 1.0, dp = 0.0, t = x;
  for (uint32_t j = 0; j < m; ++j) {
    dp = __builtin_fma(dp, t, p);
    p *= t;
    t += 1.0;
  }
  bear_dp o;
  o.D = 0.0;
  o.P = 0.0;
  if (m > 0) {
    o.D = bear_log(p);
    o.P = dp * bear_rcp(p);
  }
  if ((double)m < c) {
    bear_dp s = bear_stirling_diff(t, c - (double)m);
    o.D += s.D;
    o.P += s.P;
  }
  return o;
}

// log1p-type polynomial shared with bear_log_tab: log(1 + t) for |t| <= 2^-8 (|err| < 3e-18).  The Horner steps are spelled as
// three-address v_fma_f64: where a coefficient lives in a register pair the compiler forms v_fmac behind a 64-bit COPY of it
// (two instructions a step; same rounding, -ffp-contract=off).
__device__ __forceinline__ double bear_fma3(double a, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// COEF_V: the leading coefficient in a vector register pair too (a kernel short of SCALAR registers: the fused linear step, 1 % by
// A/B); by default it stays a scalar pair -- two more vector registers moved dm_ref_items_kernel from 78 to 82 and from 6 waves per
// SIMD to 5 (configs[3]: 128 -> 164 us).
template <bool COEF_V = false>
__device__ __forceinline__ double bear_log1p_small(double t) {
  double q;
  if (COEF_V) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "v"(-1.0 / 6.0), "v"(t), "v"(0.2));
  else asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "s"(-1.0 / 6.0), "v"(t), "v"(0.2));
  q = __builtin_fma(q, t, -0.25);
  q = bear_fma3(q, t, 1.0 / 3.0);
  q = __builtin_fma(q, t, -0.5);
  return __builtin_fma(t * t, q, t);
}

// ---- table-driven log -------------------------------------------------------------
// log(p) for finite p > 0 (normal or subnormal).  p = m * 2^e with m in [0.5, 1); the top 7
// mantissa bits pick r_i ~ 1/m from a 128-entry table {r_i, -log r_i} (built on the host with
// libm, staged in LDS); t = m r_i - 1 is one exact-rounded FMA with |t| <= 2^-8 and
//   log p = e ln2 - log r_i + log1p(t),   log1p by a degree-6 Taylor polynomial (|err| < 3e-18).
// ~15 fp64-rate instructions + one 16-byte LDS read, versus ~60 for the library log.
#define BEAR_LOGTAB_N 128
template <bool COEF_V = false>
__device__ __forceinline__ double bear_log_tab(double p, const double2 *__restrict__ tab) {
  const double m = __builtin_amdgcn_frexp_mant(p);
  const int e = __builtin_amdgcn_frexp_exp(p);
  const uint32_t hi = (uint32_t)(__double_as_longlong(m) >> 32);
  const double2 rl = tab[(hi >> 13) & 127u];
  const double t = __builtin_fma(m, rl.x, -1.0);
  const double l1p = bear_log1p_small<COEF_V>(t);
  return __builtin_fma((double)e, 0.6931471805599453094, rl.y + l1p);
}


// General item on the table log (x > 0 finite, c >= 1 an exact integer in a double): the same
// shifted-Stirling evaluation as bear_dm_item at ~150 instead of ~500 instructions.
//   log(y1 / y) = log1p(r), r = c'/y: the polynomial when r < 2^-8, else the table log of u = 1 + r plus
//   the rounding remainder (r - (u - 1)) / u  (absolute error ~1e-16 against a value >= 2^-8).
__device__ __forceinline__ bear_dp bear_dm_item_fast(double x, double c, const double2 *__restrict__ tab) {
  uint32_t m;  // factors taken by the product
  const bool stir = c > (double)BEAR_KPROD || x > 0x1p60;
  if (stir) {
    const double need = BEAR_TSTIR - x;
    m = need > 0.0 ? (uint32_t)ceil(need) : 0u;  // <= 8 < c
  } else {
    m = (uint32_t)c;
  }
  double p = 1.0, dp = 0.0, t = x;
  for (uint32_t j = 0; j < m; ++j) {
    dp = __builtin_fma(dp, t, p);
    p *= t;
    t += 1.0;
  }
  bear_dp o;
  o.D = 0.0;
  o.P = 0.0;
  if (m > 0) {
    o.D = bear_log_tab(p, tab);
    o.P = dp * bear_rcp(p);
  }
  if (stir) {
    const double y = t, cc = c - (double)m, y1 = y + cc;
    const double ry = bear_rcp(y), ry1 = bear_rcp(y1);
    const double r = cc * ry;
    const double ly = bear_log_tab(y, tab);
    const double u1 = 1.0 + r;
    const double l1p = r < 0x1p-8 ? bear_log1p_small(r) : __builtin_fma(r - (u1 - 1.0), bear_rcp(u1), bear_log_tab(u1, tab));
    o.D += __builtin_fma(y1 - 0.5, l1p, cc * (ly - 1.0)) + (bear_stir_lg(ry1) - bear_stir_lg(ry));
    o.P += l1p - 0.5 * (ry1 - ry) - (bear_stir_psi(ry1) - bear_stir_psi(ry));
  }
  return o;
}

// ---- exp(z) for z <= 0 on a 128-entry table -----------------------------------------
// z = k ln2/128 + r, |r| <= ln2/256: exp(z) = 2^(k >> 7) * tab[k & 127] * (1 + r + ... + r^5/120); tab[j] = 2^(j/128).
// Two-term Cody-Waite reduction (fdlibm's ln2 split, scaled by 1/128).  ~1 ulp; flushes to 0 below -700.
#define BEAR_EXPTAB_N 128
__device__ __forceinline__ double bear_exp_tab(double z, const double *__restrict__ tab) {
  if (z < -700.0) return 0.0;
  const double kf = __builtin_rint(z * 184.66496523378731);  // 128 / ln 2
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

// ---- reductions -----------------------------------------------------------------
// Sum over the 64 lanes, in every lane: quads and rows of 16 on DPP moves, the rows and halves on v_permlane16/32_swap (gfx950) --
// ~25 instructions where six __shfl_down of a double were twelve ds_bpermute round trips (every planned kernel's epilogue, twice).
template <int CTRL>
__device__ __forceinline__ double bear_dpp_mov(double v) {
  const long long q = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)q, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(q >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ double bear_wave_sum(double v) {
  v += bear_dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += bear_dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += bear_dpp_mov<0x124>(v);   // row_ror:4
  v += bear_dpp_mov<0x128>(v);   // row_ror:8
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const long long q = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
    const auto a = step == 0 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto c = step == 0 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __longlong_as_double(((long long)c[0] << 32) | a[0]) + __longlong_as_double(((long long)c[1] << 32) | a[1]);
  }
  return v;  // in every lane
}
// ... and the largest of 64 non-negative values, the same way
__device__ __forceinline__ double bear_wave_max(double v) {
  v = __builtin_fmax(v, bear_dpp_mov<0xB1>(v));
  v = __builtin_fmax(v, bear_dpp_mov<0x4E>(v));
  v = __builtin_fmax(v, bear_dpp_mov<0x124>(v));
  v = __builtin_fmax(v, bear_dpp_mov<0x128>(v));
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const long long q = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
    const auto a = step == 0 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto c = step == 0 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __builtin_fmax(__longlong_as_double(((long long)c[0] << 32) | a[0]), __longlong_as_double(((long long)c[1] << 32) | a[1]));
  }
  return v;
}

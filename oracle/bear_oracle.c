/*
 * CPU oracle, C restatement (TEST INFRASTRUCTURE ONLY -- never linked into or called
 * by the product library; used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py).
 *
 * Restates, with libm lgamma_r and a series digamma, the arithmetic that the
 * reference runs through TensorFlow(-Probability) on CPU:
 *   - bear_model/core.py:73-74      DM ordered log-prob  lbeta(a+c) - lbeta(a)
 *   - bear_model/core.py:138-139    multinomial ordered log-prob  sum c log p
 *   - bear_model/bear_net.py:43,68  alpha = f/h + eps ; probs = f + eps
 *   - bear_model/bear_ref.py:30-33, 63-68, 332-337  Jukes-Cantor reference prior
 *   - bear_model/bear_net.py:190-193, bear_ref.py:252-255  gradients (autodiff of
 *     lgamma = digamma), stated analytically.
 * TF evaluates lgamma through Eigen -> libm lgamma_r, the same routine used here.
 *
 * Pinning: identical closed forms to oracle/bear_oracle.py (which carries the
 * pinning statement); tests/test_oracle.py checks this file against that one and
 * against the committed golden values.  BEAR-mode ELBO / gradients: parity
 * unpinned by reference-held vectors (see bear_oracle.py header).
 *
 * Threading: OpenMP over rows, static schedule, per-thread partials summed in
 * thread order (deterministic for a fixed thread count).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

extern double lgamma_r(double, int *);

static inline double lg(double x) {
  int s;
  return lgamma_r(x, &s);
}

/* digamma: upward recurrence to x >= 12, then the asymptotic series. |err| ~ 1e-15 */
static double digamma(double x) {
  double r = 0.0;
  while (x < 12.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  double f = 1.0 / (x * x);
  double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0 + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
  return r + log(x) - 0.5 / x + t;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* One DM row: returns LL_i, writes g[b] = dLL_i/dalpha_b (core.py:73-74 + autodiff). */
static inline double dm_row(const double *a, const double *c, int W, double *g) {
  double A = 0.0, n = 0.0, ll = 0.0;
  for (int b = 0; b < W; ++b) {
    A += a[b];
    n += c[b];
  }
  for (int b = 0; b < W; ++b) ll += lg(a[b] + c[b]) - lg(a[b]);
  ll -= lg(A + n) - lg(A);
  if (g) {
    double gn = digamma(A + n) - digamma(A);
    for (int b = 0; b < W; ++b) g[b] = digamma(a[b] + c[b]) - digamma(a[b]) - gn;
  }
  return ll;
}

/*
 * bear_net._train_step arithmetic for one batch, unscaled (bear_net.py:146-197).
 * counts [n,5] uint32, prior [n,5] f64 (= ar_func rows).  out[0] = sum LL,
 * out[1] = d sum LL / d h_signed.  grad_prior (nullable) [n,5] = dLL/dprior.
 */
void oracle_dm_prior_f64(const uint32_t *counts, const double *prior, uint64_t n_rows,
                         double h_signed, double eps, int train_ar, double *out,
                         double *grad_prior, int nthreads) {
  const int W = 5;
  double h = exp(h_signed);
  if (nthreads < 1) nthreads = 1;
  double *part = (double *)calloc((size_t)nthreads * 2, sizeof(double));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double ll = 0.0, dh = 0.0;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n_rows; ++i) {
      double a[5], c[5], g[5];
      const double *f = prior + (size_t)i * W;
      for (int b = 0; b < W; ++b) c[b] = (double)counts[(size_t)i * W + b];
      if (train_ar) {
        for (int b = 0; b < W; ++b) {
          double p = f[b] + eps;
          if (c[b] != 0.0) ll += c[b] * log(p);
          if (grad_prior) grad_prior[(size_t)i * W + b] = c[b] / p;
        }
      } else {
        for (int b = 0; b < W; ++b) a[b] = f[b] / h + eps;
        ll += dm_row(a, c, W, g);
        for (int b = 0; b < W; ++b) {
          dh += g[b] * (-f[b] / h);
          if (grad_prior) grad_prior[(size_t)i * W + b] = g[b] / h;
        }
      }
    }
    part[2 * tid] = ll;
    part[2 * tid + 1] = dh;
  }
  out[0] = out[1] = 0.0;
  for (int t = 0; t < nthreads; ++t) {
    out[0] += part[2 * t];
    out[1] += part[2 * t + 1];
  }
  free(part);
}

/*
 * bear_ref._train_step arithmetic with the stop net-function (ar_funcs.py:121-126),
 * unscaled (bear_ref.py:207-259; prior bear_ref.py:30-33, 63-68; reference-column
 * preprocessing bear_ref.py:332-337).  out = [sum LL, d/dh_signed, d/dtau_signed,
 * d/dnet_weight_signed].
 */
void oracle_dm_ref_f64(const uint32_t *train, const uint32_t *ref, uint64_t n_rows,
                       double h_signed, double tau_signed, double nu_signed, double eps,
                       int train_ar, double *out, int nthreads) {
  const int W = 5;
  double h = exp(h_signed), tau = exp(tau_signed), nw = exp(nu_signed);
  double E = exp(-tau);
  if (nthreads < 1) nthreads = 1;
  double *part = (double *)calloc((size_t)nthreads * 4, sizeof(double));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double acc[4] = {0, 0, 0, 0};
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n_rows; ++i) {
      double c[5], r[5], f[5], dft[5], dfn[5], a[5], g[5], dLdf[5];
      double R = 0.0;
      for (int b = 0; b < W; ++b) {
        c[b] = (double)train[(size_t)i * W + b];
        r[b] = (b < W - 1) ? ((double)ref[(size_t)i * W + b] + eps) : 0.0;
        R += r[b];
      }
      for (int b = 0; b < W; ++b) {
        double sh = (b < W - 1) ? 1.0 : 0.0;
        double gnet = (b < W - 1) ? 0.0 : 1.0;
        double norm = r[b] / R;
        double base = 0.25 * sh + E * (norm - 0.25 * sh);
        f[b] = (nw * gnet + base) / (nw + 1.0);
        dft[b] = (-tau * E) * (norm - 0.25 * sh) / (nw + 1.0);
        dfn[b] = nw * (gnet - f[b]) / (nw + 1.0);
      }
      if (train_ar) {
        for (int b = 0; b < W; ++b) {
          double p = f[b] + eps;
          if (c[b] != 0.0) acc[0] += c[b] * log(p);
          dLdf[b] = c[b] / p;
        }
      } else {
        for (int b = 0; b < W; ++b) a[b] = f[b] / h + eps;
        acc[0] += dm_row(a, c, W, g);
        for (int b = 0; b < W; ++b) {
          dLdf[b] = g[b] / h;
          acc[1] += g[b] * (-f[b] / h);
        }
      }
      for (int b = 0; b < W; ++b) {
        acc[2] += dLdf[b] * dft[b];
        acc[3] += dLdf[b] * dfn[b];
      }
    }
    for (int k = 0; k < 4; ++k) part[4 * tid + k] = acc[k];
  }
  for (int k = 0; k < 4; ++k) {
    out[k] = 0.0;
    for (int t = 0; t < nthreads; ++t) out[k] += part[4 * t + k];
  }
  free(part);
}

/*
 * L1 MASS of the gradients above: the sum of the absolute values of the terms each gradient is the sum of, at the finest
 * granularity this file has (per row and letter, the item part psi(a+c) - psi(a) and the context part psi(A+n) - psi(A)
 * counted separately: they cancel).  The parity tests bound a gradient's error by a multiple of ITS OWN mass -- the scale
 * rounding errors of a sum live on -- instead of borrowing the ELBO's magnitude.  mass[0] = d/dh_signed; the reference form
 * also mass[1] = d/dtau_signed, mass[2] = d/dnet_weight_signed.  BEAR mode only for d/dh (AR mode has no h gradient).
 */
void oracle_dm_prior_mass_f64(const uint32_t *counts, const double *prior, uint64_t n_rows, double h_signed, double eps,
                              double *mass, int nthreads) {
  const int W = 5;
  const double h = exp(h_signed);
  if (nthreads < 1) nthreads = 1;
  double m = 0.0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : m)
#endif
  for (int64_t i = 0; i < (int64_t)n_rows; ++i) {
    const double *f = prior + (size_t)i * W;
    double A = 0.0, n = 0.0;
    for (int b = 0; b < W; ++b) {
      A += f[b] / h + eps;
      n += (double)counts[(size_t)i * W + b];
    }
    const double gn = digamma(A + n) - digamma(A);
    for (int b = 0; b < W; ++b) {
      const double a = f[b] / h + eps, c = (double)counts[(size_t)i * W + b];
      m += (fabs(digamma(a + c) - digamma(a)) + fabs(gn)) * fabs(f[b] / h);
    }
  }
  mass[0] = m;
}

void oracle_dm_ref_mass_f64(const uint32_t *train, const uint32_t *ref, uint64_t n_rows, double h_signed, double tau_signed,
                            double nu_signed, double eps, int train_ar, double *mass, int nthreads) {
  const int W = 5;
  const double h = exp(h_signed), tau = exp(tau_signed), nw = exp(nu_signed), E = exp(-tau);
  if (nthreads < 1) nthreads = 1;
  double m0 = 0.0, m1 = 0.0, m2 = 0.0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : m0, m1, m2)
#endif
  for (int64_t i = 0; i < (int64_t)n_rows; ++i) {
    double c[5], r[5], f[5], dft[5], dfn[5];
    double R = 0.0, A = 0.0, n = 0.0;
    for (int b = 0; b < W; ++b) {
      c[b] = (double)train[(size_t)i * W + b];
      r[b] = (b < W - 1) ? ((double)ref[(size_t)i * W + b] + eps) : 0.0;
      R += r[b];
    }
    for (int b = 0; b < W; ++b) {
      const double sh = (b < W - 1) ? 1.0 : 0.0, gnet = (b < W - 1) ? 0.0 : 1.0, norm = r[b] / R;
      const double base = 0.25 * sh + E * (norm - 0.25 * sh);
      f[b] = (nw * gnet + base) / (nw + 1.0);
      /* magnitudes of the terms INSIDE df/dtau and df/dnu as well: norm - 1/4 and gnet - f cancel (exactly, for a context
         without reference counts), and what is left of them in floating point scales with the parts, not with the difference */
      dft[b] = (tau * E) * (fabs(norm) + 0.25 * sh) / (nw + 1.0);
      dfn[b] = nw * (fabs(gnet) + fabs(f[b])) / (nw + 1.0);
      A += f[b] / h + eps;
      n += c[b];
    }
    const double gn = train_ar ? 0.0 : digamma(A + n) - digamma(A);
    for (int b = 0; b < W; ++b) {
      double q;      /* |dLL/df_b|, item and context part apart */
      if (train_ar) {
        q = c[b] / (f[b] + eps);
      } else {
        const double a = f[b] / h + eps;
        q = (fabs(digamma(a + c[b]) - digamma(a)) + fabs(gn)) / h;
        m0 += q * fabs(f[b]);
      }
      m1 += q * fabs(dft[b]);
      m2 += q * fabs(dfn[b]);
    }
  }
  mass[0] = m0;
  mass[1] = m1;
  mass[2] = m2;
}

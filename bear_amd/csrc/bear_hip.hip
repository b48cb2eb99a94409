// bear_hip.hip -- kernels + C ABI (include/bear_hip.h) for the BEAR training hot path on gfx950.
//
// Kernels live in kernels_sorted.h (BEAR mode: sorted work items), kernels_rows.h (AR mode and the
// gradient-row variant) and kernels_synth.h (synthetic tables); math in bear_math.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "bear_common.h"
#include "bear_levels.h"
#include "kernels_rows.h"
#include "kernels_eval.h"
#include "kernels_sorted.h"
#include "kernels_plan.h"
#include "kernels_synth.h"
#include "kernels_linear.h"
#include "kernels_linrows.h"
#include "kernels_refmix.h"
#include "kernels_mixplan.h"
#include "kernels_sample.h"
#include "kernels_shuffle.h"
#include "kernels_cnn.h"
#include "kernels_evalplan.h"
#include "kernels_refplan.h"

#ifdef EVP_STAMPS
#define EVP_DBG_ARG , ws->dbg
#else
#define EVP_DBG_ARG
#endif
#ifdef PLN_STAMPS  // developer build: per-wave phase timers of dm_prior_plan_kernel land in ws->dbg
#define PLN_DBG_ARG , ws->dbg
#else
#define PLN_DBG_ARG
#endif

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
      if (e == hipSuccess) e = hipMalloc(&ws->logtab, sizeof(double) * 2 * BEAR_LOGTAB_N);
      if (e == hipSuccess) e = hipMalloc(&ws->dbg, sizeof(unsigned long long) * 48 * (size_t)ws->max_blocks);
      ws->eval_blocks = ws->num_cu * 8;
      if (e == hipSuccess) e = hipMalloc(&ws->eval_partials, sizeof(double) * EVL_MAX_OUT * (size_t)ws->eval_blocks);
      if (e == hipSuccess) e = hipMalloc(&ws->eval_out, sizeof(double) * EVL_MAX_OUT);
      if (e == hipSuccess) e = hipMalloc(&ws->lin_partials, sizeof(double) * LIN_MAX_GRAD * (size_t)ws->num_cu * PLN_BLOCKS_PER_CU);
      if (e == hipSuccess) e = hipMalloc(&ws->lin_accum, sizeof(double) * LIN_MAX_GRAD);
      if (e == hipSuccess) e = hipMemset(ws->lin_accum, 0, sizeof(double) * LIN_MAX_GRAD);
      if (e == hipSuccess) e = hipMalloc(&ws->arrive, sizeof(unsigned long long) * BEAR_ARRIVE_WORDS);
      if (e == hipSuccess) e = hipMemset(ws->arrive, 0, sizeof(unsigned long long) * BEAR_ARRIVE_WORDS);
      ws->epoch = 0;
#define LIN_ALL_NGK(AR, PAIRED, DET)                                                                                     \
  reinterpret_cast<const void *>(dm_linear_plan_kernel<AR, PAIRED, DET, 0>), reinterpret_cast<const void *>(dm_linear_plan_kernel<AR, PAIRED, DET, 2>), \
      reinterpret_cast<const void *>(dm_linear_plan_kernel<AR, PAIRED, DET, 6>), reinterpret_cast<const void *>(dm_linear_plan_kernel<AR, PAIRED, DET, 7>)
      for (const void *fn : {LIN_ALL_NGK(false, false, false), LIN_ALL_NGK(true, false, false), LIN_ALL_NGK(false, true, false),
                             LIN_ALL_NGK(true, true, false), LIN_ALL_NGK(false, false, true), LIN_ALL_NGK(true, false, true),
                             LIN_ALL_NGK(false, true, true), LIN_ALL_NGK(true, true, true)})
#undef LIN_ALL_NGK
        if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_lin));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(cnn_backward_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)cnh_lds_bytes());
      if (e == hipSuccess) {
        // {r_i, -log r_i}: r_i = 1 / midpoint of the i-th mantissa cell of [0.5, 1) (bear_log_tab)
        double tab[2 * BEAR_LOGTAB_N];
        for (int i = 0; i < BEAR_LOGTAB_N; ++i) {
          const double r = 1.0 / ((BEAR_LOGTAB_N + i + 0.5) / (2.0 * BEAR_LOGTAB_N));
          tab[2 * i] = r;
          tab[2 * i + 1] = -log(r);
        }
        e = hipMemcpy(ws->logtab, tab, sizeof(tab), hipMemcpyHostToDevice);
      }
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_sorted_kernel<0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(srt_lds_n));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_sorted_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(srt_lds_n));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_sorted_kernel<9>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(srt_lds_n));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_plan_kernel<false, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_n));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_plan_grad_kernel<false, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_g));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_plan_grad_inplace_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_gi));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_refmix_plan_grad_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_g));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_refmix_plan_grad_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_g));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_plan_kernel<true, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_n));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_ref_plan_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_r));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_ref_plan_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_r));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_plan_kernel<true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_n));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_prior_plan_grad_kernel<true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(pln_lds_g));
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(dm_ref_sorted_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(srt_lds_r));
      {
        const void *evp[] = {reinterpret_cast<const void *>(eval_plan_kernel<0, 4>), reinterpret_cast<const void *>(eval_plan_kernel<1, 0>),
                             reinterpret_cast<const void *>(eval_plan_kernel<1, 4>), reinterpret_cast<const void *>(eval_plan_kernel<4, 0>)};
        for (const void *fn : evp)
          if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(evp_lds));
      }
    }
  }
  (void)hipSetDevice(prev);
  if (!ws) return e == hipSuccess ? BEAR_ERR_NOMEM : (g_last_hip_error = (int)e, BEAR_ERR_HIP);
  if (e != hipSuccess) {
    (void)hipFree(ws->partials);
    (void)hipFree(ws->logtab);
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
  (void)hipFree(ws->logtab);
  (void)hipFree(ws->dbg);
  (void)hipFree(ws->eval_partials);
  (void)hipFree(ws->eval_out);
  (void)hipFree(ws->lin_partials);
  (void)hipFree(ws->lin_accum);
  if (ws->cnn_partials) (void)hipFree(ws->cnn_partials);
  if (ws->arrive) (void)hipFree(ws->arrive);
  (void)hipSetDevice(prev);
  delete ws;
  return BEAR_OK;
}

// a fresh stamp for the launch that is about to use ws->arrive (bear_arrival, bear_common.h); 0 is the word's idle value
static bear_arrival ws_arrival(bear_ws *ws) {
  if (++ws->epoch == 0u) ws->epoch = 1u;
  return bear_arrival{ws->arrive, ws->epoch};
}

// BEAR_AMD_DETERMINISTIC=1: parameter gradients that are bit-identical from run to run (kernels_linear.h: fixed-point gradient
// tables; cnn_backward_grid: one wave per block).  Read per call: a process may switch it between steps (tests).
static bool bear_deterministic() {
#ifdef BEAR_DET_BUILD      // libbear_hip_det.so: everything deterministic, always (kernels_plan.h, PLN_FOR_UNITS)
  return true;
#else
  const char *e = getenv("BEAR_AMD_DETERMINISTIC");
  return e && e[0] && e[0] != '0';
#endif
}

int bear_deterministic_build(void) {
#ifdef BEAR_DET_BUILD
  return 1;
#else
  return 0;
#endif
}

static bear_step_io ws_io(bear_ws *ws, const double *theta, int kind, double *out) {
  const bear_arrival a = ws_arrival(ws);
  bear_step_io io;
  io.theta = theta;
  io.kind = kind;
  io.epoch = a.epoch;
  io.out = out;
  io.arrive_word = a.word;
  return io;
}

static int check_ws(const bear_ws *ws) {
  if (!ws) return BEAR_ERR_INVALID_ARG;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return BEAR_ERR_NO_DEVICE;
  if (dev != ws->device) return BEAR_ERR_WRONG_DEVICE;
  return BEAR_OK;
}

static inline bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
static inline bool misaligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) != 0; }

static int grid_sorted(const bear_ws *ws, uint64_t n_rows) {
  uint64_t tiles = (n_rows + SRT_TILE - 1) / SRT_TILE;
  uint64_t g = (uint64_t)ws->num_cu * 2;  // two resident blocks per CU (LDS-limited)
  if (g > (uint64_t)ws->max_blocks) g = ws->max_blocks;
  if (tiles < g) g = tiles;
  return g < 1 ? 1 : (int)g;
}

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
  int grid = grid_for(ws, n_rows);
  if (!train_ar && !grad_prior) {
    grid = grid_sorted(ws, n_rows);
    const char *dbg = getenv("BEAR_DEBUG_STOP");  // developer switch: phase timing (results are then meaningless)
    const int stop = dbg ? atoi(dbg) : 0;
    const double2 *lt = reinterpret_cast<const double2 *>(ws->logtab);
    if (stop == 1)
      hipLaunchKernelGGL(dm_prior_sorted_kernel<1>, dim3(grid), dim3(SRT_THREADS), sizeof(srt_lds_n), s, counts, prior, n_rows, prm, lt, ws->partials, ws->dbg);
    else if (stop == 9)
      hipLaunchKernelGGL(dm_prior_sorted_kernel<9>, dim3(grid), dim3(SRT_THREADS), sizeof(srt_lds_n), s, counts, prior, n_rows, prm, lt, ws->partials, ws->dbg);
    else
      hipLaunchKernelGGL(dm_prior_sorted_kernel<0>, dim3(grid), dim3(SRT_THREADS), sizeof(srt_lds_n), s, counts, prior, n_rows, prm, lt, ws->partials, ws->dbg);
  } else if (train_ar) {
    if (grad_prior)
      hipLaunchKernelGGL((dm_prior_kernel<true, true>), dim3(grid), dim3(BEAR_THREADS), 0, s, counts, prior, n_rows, prm, grad_prior, reinterpret_cast<const double2 *>(ws->logtab), ws->partials);
    else
      hipLaunchKernelGGL((dm_prior_kernel<true, false>), dim3(grid), dim3(BEAR_THREADS), 0, s, counts, prior, n_rows, prm, grad_prior, reinterpret_cast<const double2 *>(ws->logtab), ws->partials);
  } else {
    if (grad_prior)
      hipLaunchKernelGGL((dm_prior_kernel<false, true>), dim3(grid), dim3(BEAR_THREADS), 0, s, counts, prior, n_rows, prm, grad_prior, reinterpret_cast<const double2 *>(ws->logtab), ws->partials);
    else
      hipLaunchKernelGGL((dm_prior_kernel<false, false>), dim3(grid), dim3(BEAR_THREADS), 0, s, counts, prior, n_rows, prm, grad_prior, reinterpret_cast<const double2 *>(ws->logtab), ws->partials);
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
  prm.tau = tau;
  prm.V = 1.0 / (nw + 1.0);
  prm.nw = nw;
  int grid = grid_for(ws, n_rows);
  if (train_ar) {
    hipLaunchKernelGGL((dm_ref_kernel<true>), dim3(grid), dim3(BEAR_THREADS), 0, s, train, ref, n_rows, prm, reinterpret_cast<const double2 *>(ws->logtab), ws->partials);
  } else if (getenv("BEAR_ROWS_KERNEL")) {  // developer switch: v1 row-per-thread kernel (A/B measurements)
    hipLaunchKernelGGL((dm_ref_kernel<false>), dim3(grid), dim3(BEAR_THREADS), 0, s, train, ref, n_rows, prm, reinterpret_cast<const double2 *>(ws->logtab), ws->partials);
  } else {
    grid = grid_sorted(ws, n_rows);
    hipLaunchKernelGGL(dm_ref_sorted_kernel, dim3(grid), dim3(SRT_THREADS), sizeof(srt_lds_r), s, train, ref, n_rows, prm,
                       reinterpret_cast<const double2 *>(ws->logtab), ws->partials);
  }
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, ws->partials, grid, 4, out);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

// ------------------------------------------------------------------ plans
struct bear_plan {
  int device;
  int ncol;
  uint64_t n_rows;
  const uint32_t *counts;  // the buffer the plan was built from (identity check only)
  pln_tile *tiles;
  unsigned char *stream;
  pln_heavy_col *heavy_col;
  pln_heavy_row *heavy_row;
  uint64_t *heavy_stop;
  unsigned long long *hist;  // [64]
  uint16_t *live;            // five-column plans: per-tile lists of the contexts that hold counts (plan_live_kernel)
  uint64_t n_tiles;
  uint64_t n_heavy[3];
  int rows_ref;              // bear_plan_create_ref: the DENSE form (a table of large counts: nothing kept per item, dm_ref_rows_kernel)
  uint64_t n_live_rows;      // five-column plans: contexts that hold any count (the kernels that walk `live` skip the lists when all do)
  double count_total[3];     // of the table (all five columns): sum of all counts, cells that hold one, largest count
  double count_bound[3];     // the same of everything that is added into one gradient (bear_plan_set_count_bound; default: count_total)
  uint64_t bytes;
  // bear_plan_pair_contexts: the paired form of `live` for the index words at pair_codes (kernels_linear.h), and the plan's
  // tiles sorted into those that took it (tiles_p) and those that keep their plain list (tiles_u), each followed by PLN_DESC_PAD
  // zeroed descriptors; spare word of a descriptor = tile number << 32 | entries of the paired list
  // bear_plan_attach_cnn_levels: prefix levels of the (k-mer-sorted) contexts at cnn_codes for the convolutional step
  // (kernels_cnn.h, cnn_level_io); levels[k - 1] = level k, k = 1 .. n_cnn_levels
  bear_level_dev cnn_levels[CNN_MAX_LAG];
  int n_cnn_levels, cnn_lag, cnn_fw;
  // ... and window tables (bear_window_dev) per level k = 0 (the contexts) .. n_cnn_levels: cnn_win[k][q], q < n_cnn_win[k], are the
  // tables of the LAST n_cnn_win[k] positions of the level's range, ascending
  bear_window_dev cnn_win[CNN_MAX_LAG + 1][CNN_MAX_WIN];
  int n_cnn_win[CNN_MAX_LAG + 1];
  int n_cnn_windows;         // all of them
  const uint64_t *cnn_codes;
  uint16_t *live2;
  pln_tile *tiles_p, *tiles_u;
  uint64_t n_tiles_p, n_tiles_u;
  const uint64_t *pair_codes;
  int pair_lag;
  // reference-aware extension (bear_plan_create_ref, kernels_refplan.h)
  const uint32_t *ref;
  rpl_item *ref_items;
  uint64_t n_ref_items, n_heavy0;
  unsigned long long *hist0;   // [RPL_NKEY], inside the allocation hist0_base
  unsigned long long *hist0_base;
  uint32_t *heavy0;
  double *sum0;
};

static void plan_free(bear_plan *p) {
  if (!p) return;
  (void)hipFree(p->tiles);
  (void)hipFree(p->stream);
  (void)hipFree(p->heavy_col);
  (void)hipFree(p->heavy_row);
  (void)hipFree(p->heavy_stop);
  (void)hipFree(p->hist);
  (void)hipFree(p->live);
  for (int k = 0; k < p->n_cnn_levels; ++k) bear_level_free(&p->cnn_levels[k]);
  for (int k = 0; k <= CNN_MAX_LAG; ++k)
    for (int q = 0; q < p->n_cnn_win[k]; ++q) bear_window_free(&p->cnn_win[k][q]);
  (void)hipFree(p->live2);
  (void)hipFree(p->tiles_p);
  (void)hipFree(p->tiles_u);
  (void)hipFree(p->ref_items);
  (void)hipFree(p->hist0_base);
  (void)hipFree(p->heavy0);
  (void)hipFree(p->sum0);
  delete p;
}

int bear_plan_create(bear_ws *ws, const uint32_t *counts, uint64_t n_rows, int ncol, bear_plan **out) {
  if (!out) return BEAR_ERR_INVALID_ARG;
  *out = nullptr;
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if ((ncol != 4 && ncol != 5) || (n_rows && !counts) || misaligned(counts)) return BEAR_ERR_INVALID_ARG;
  bear_plan *p = new (std::nothrow) bear_plan();
  if (!p) return BEAR_ERR_NOMEM;
  memset(p, 0, sizeof(*p));
  p->device = ws->device;
  p->ncol = ncol;
  p->n_rows = n_rows;
  p->counts = counts;
  hipError_t e = hipMalloc(&p->hist, sizeof(unsigned long long) * (2 * SRT_NKEY + PLN_NBIG + 1));     // (+ the large totals' histogram)
  if (e == hipSuccess) e = hipMemset(p->hist, 0, sizeof(unsigned long long) * (2 * SRT_NKEY + PLN_NBIG + 1));
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    plan_free(p);
    return BEAR_ERR_HIP;
  }
  if (n_rows == 0) {
    *out = p;
    return BEAR_OK;
  }
  // ---- pass A: product-path items per group of 4 contexts, heavy counts, histograms
  const uint64_t n_quads = (n_rows + PLN_QUAD - 1) / PLN_QUAD;
  uint8_t *d_quad = nullptr, *h_quad = nullptr;
  unsigned long long *d_cnt = nullptr;  // [0..2] heavy counts, [3..5] fill cursors, [6..8] sum of all counts, cells that hold one, largest count
  std::vector<pln_tile> tiles;
  e = hipMalloc(&d_quad, 3 * n_quads);
  if (e == hipSuccess) e = hipMalloc(&d_cnt, sizeof(unsigned long long) * 9);
  if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof(unsigned long long) * 9);
  if (e == hipSuccess) {
    uint64_t gb = (n_quads + 255) / 256;
    const int grid = (int)(gb < (uint64_t)ws->num_cu * 8 ? gb : (uint64_t)ws->num_cu * 8);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(grid), dim3(256), 0, 0, counts, n_rows, ncol, d_quad, d_cnt, p->hist);
    e = hipGetLastError();
  }
  unsigned long long h_cnt[3] = {0, 0, 0}, h_total[3] = {0, 0, 0};
  if (e == hipSuccess) e = hipMemcpy(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_total, d_cnt + 6, sizeof(h_total), hipMemcpyDeviceToHost);
  for (int k = 0; k < 3; ++k) p->count_total[k] = p->count_bound[k] = (double)h_total[k];
  // ---- tiles: greedy cut so that a tile holds <= PLN_NI items and <= PLN_RMAX contexts -- on the device (plan_cut_*_kernel);
  // BEAR_PLAN_CUT=host keeps the sequential host loop over the per-group counters (the definition; used by the tests to compare)
  uint64_t off16 = 0, n_tiles = 0;
  const char *cut_env = getenv("BEAR_PLAN_CUT");
  const bool host_cut = cut_env && cut_env[0] == 'h';
  if (e == hipSuccess && host_cut) {
    h_quad = (uint8_t *)malloc(3 * n_quads);
    if (!h_quad) e = hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemcpy(h_quad, d_quad, 3 * n_quads, hipMemcpyDeviceToHost);
  }
  if (e == hipSuccess && host_cut) {
    uint64_t q = 0;
    while (q < n_quads) {
      uint32_t items = 0, rows = 0, hcol = 0, hrow = 0;
      const uint64_t q0 = q;
      while (q < n_quads && rows + PLN_QUAD <= PLN_RMAX && items + h_quad[q] <= PLN_NI_CUT) {
        items += h_quad[q];
        hcol += h_quad[n_quads + q];
        hrow += h_quad[2 * n_quads + q];
        rows += PLN_QUAD;
        ++q;
      }
      pln_tile ti;
      memset(&ti, 0, sizeof(ti));
      ti.row0 = q0 * PLN_QUAD;
      if (ti.row0 + rows > n_rows) rows = (uint32_t)(n_rows - ti.row0);  // ragged end of the table
      // large-count items evaluated inside the tile (their rows are in LDS anyway); the surplus of very dense
      // tiles goes to the global lists.  Mode R needs no row data for large totals: they stay global.
      const uint32_t hc = hcol < PLN_HCAP ? hcol : PLN_HCAP;
      const uint32_t hr = ncol == 5 ? (hrow < PLN_HCAP ? hrow : PLN_HCAP) : 0u;
      ti.rows_items = (rows << 16) | items;
      ti.off16 = (uint32_t)off16;
      ti.hc_hr = (hc << 16) | hr;
      ti.blk16 = pln_block_layout(rows, items, hc, hr).end / 16;
      off16 += ti.blk16;
      if (off16 > 0xffffffffull) {
        e = hipErrorOutOfMemory;
        break;
      }
      tiles.push_back(ti);
    }
    n_tiles = tiles.size();
  }
  free(h_quad);
  if (e == hipSuccess && !host_cut) {
    const uint32_t chunk = plan_cut_chunk(n_quads);
    const uint64_t n_chunks = (n_quads + chunk - 1) / chunk;
    uint32_t *d_walk = nullptr;
    uint16_t *d_step = nullptr;            // [n_quads] length of the tile that starts at a group
    uint64_t *d_entry = nullptr;           // [n_chunks] entry | [n_chunks] base
    unsigned long long *d_meta = nullptr, h_meta[3] = {0, 0, 0};
    e = hipMalloc(&d_walk, sizeof(uint32_t) * n_chunks * PLN_CUT_SPAN);
    if (e == hipSuccess) e = hipMalloc(&d_step, sizeof(uint16_t) * n_quads);
    if (e == hipSuccess) e = hipMalloc(&d_entry, sizeof(uint64_t) * 2 * n_chunks);
    if (e == hipSuccess) e = hipMalloc(&d_meta, sizeof(h_meta));
    if (e == hipSuccess) e = hipMemset(d_meta, 0, sizeof(h_meta));
    if (e == hipSuccess) {
      const uint64_t wb = (n_chunks * PLN_CUT_SPAN + 255) / 256;
      hipLaunchKernelGGL(plan_cut_step_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, 0, d_quad, n_quads, n_chunks, chunk, d_step);
      hipLaunchKernelGGL(plan_cut_walk_kernel, dim3((unsigned)wb), dim3(256), 0, 0, d_step, n_quads, n_chunks, chunk, d_walk);
      hipLaunchKernelGGL(plan_cut_chain_kernel, dim3(1), dim3(1), 0, 0, d_walk, n_chunks, chunk, d_entry, d_entry + n_chunks, d_meta);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h_meta, d_meta, sizeof(unsigned long long), hipMemcpyDeviceToHost);
    n_tiles = h_meta[0];
    // + PLN_DESC_PAD zeroed descriptors: the kernels fetch descriptors 32 at a time (1 KiB LDS-DMA pieces)
    if (e == hipSuccess) e = hipMalloc(&p->tiles, sizeof(pln_tile) * (n_tiles + PLN_DESC_PAD));
    if (e == hipSuccess) e = hipMemset(p->tiles, 0, sizeof(pln_tile) * (n_tiles + PLN_DESC_PAD));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(plan_cut_write_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, 0, d_quad, n_quads, n_rows, ncol,
                         n_chunks, chunk, d_entry, d_entry + n_chunks, p->tiles);
      hipLaunchKernelGGL(plan_cut_offsets_kernel, dim3(1), dim3(1024), 0, 0, p->tiles, n_tiles, d_meta);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h_meta, d_meta, sizeof(h_meta), hipMemcpyDeviceToHost);
    if (e == hipSuccess && h_meta[2]) e = hipErrorOutOfMemory;     // plan stream beyond 2^32 16-byte units
    off16 = h_meta[1];
    (void)hipFree(d_walk);
    (void)hipFree(d_step);
    (void)hipFree(d_entry);
    (void)hipFree(d_meta);
  }
  (void)hipFree(d_quad);
  p->n_tiles = n_tiles;
  const uint64_t stream_bytes = off16 * 16 + 1024;  // slack: a DMA piece may be issued for a partial KiB
  for (int k = 0; k < 3; ++k) p->n_heavy[k] = h_cnt[k];
  if (host_cut) {   // + PLN_DESC_PAD zeroed descriptors: the kernels fetch descriptors 32 at a time (1 KiB LDS-DMA pieces)
    if (e == hipSuccess) e = hipMalloc(&p->tiles, sizeof(pln_tile) * (tiles.size() + PLN_DESC_PAD));
    if (e == hipSuccess) e = hipMemset(p->tiles, 0, sizeof(pln_tile) * (tiles.size() + PLN_DESC_PAD));
    if (e == hipSuccess) e = hipMemcpy(p->tiles, tiles.data(), sizeof(pln_tile) * tiles.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipMalloc(&p->stream, stream_bytes);
  if (e == hipSuccess) e = hipMemset(p->stream, 0, stream_bytes);
  if (e == hipSuccess && h_cnt[0]) e = hipMalloc(&p->heavy_col, sizeof(pln_heavy_col) * h_cnt[0]);
  if (e == hipSuccess && h_cnt[1]) e = hipMalloc(&p->heavy_row, sizeof(pln_heavy_row) * h_cnt[1]);
  if (e == hipSuccess && h_cnt[2]) e = hipMalloc(&p->heavy_stop, sizeof(uint64_t) * h_cnt[2]);
  if (e == hipSuccess) {
    const uint64_t nt = n_tiles;
    const int grid = (int)(nt < (uint64_t)ws->num_cu * 2 ? nt : (uint64_t)ws->num_cu * 2);
    hipLaunchKernelGGL(plan_fill_kernel, dim3(grid), dim3(1024), 0, 0, counts, n_rows, ncol, p->tiles, nt, p->stream,
                       p->heavy_col, p->heavy_row, p->heavy_stop, d_cnt + 3);
    e = hipGetLastError();
  }
  const uint64_t live_bytes = ncol == 5 ? sizeof(uint16_t) * PLN_LIVE_STRIDE * (n_tiles + 1) : 0;   // + 1: DMA pieces are whole KiB
  if (e == hipSuccess && live_bytes) e = hipMalloc(&p->live, live_bytes);
  if (e == hipSuccess && live_bytes) e = hipMemset(p->live, 0, live_bytes);
  if (e == hipSuccess && live_bytes) {
    const uint64_t nt = n_tiles;
    const int grid = (int)(nt < (uint64_t)ws->num_cu * 2 ? nt : (uint64_t)ws->num_cu * 2);
    hipLaunchKernelGGL(plan_live_kernel, dim3(grid), dim3(1024), 0, 0, p->tiles, nt, p->stream, p->live);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  unsigned long long h_used[3] = {0, 0, 0};  // entries that actually went to the global lists
  if (e == hipSuccess) e = hipMemcpy(h_used, d_cnt + 3, sizeof(h_used), hipMemcpyDeviceToHost);
  for (int k = 0; k < 3; ++k) p->n_heavy[k] = h_used[k];
  // The global lists were filled through atomic cursors, in whatever order the blocks got there: into row order (bear_levels.h).
  // Neighbouring threads of the kernels that walk them then read neighbouring prior cells and -- the gradient fix-up -- update
  // neighbouring gradient cells without atomics; and the lists are the same bits in every build of the plan.
  // (lists of a few thousand entries -- every sparse k-mer table -- are left as they are: nothing to coalesce, and the sorts' set-up
  // would triple the plan's build time; the deterministic build sorts them all)
#ifdef BEAR_DET_BUILD
  const uint64_t sort_from = 2;
#else
  const uint64_t sort_from = 1u << 16;
#endif
  if (e == hipSuccess) {
    int cst = h_used[0] >= sort_from ? bear_canonical_order(p->heavy_col, h_used[0], 16, 0) : BEAR_OK;
    if (cst == BEAR_OK && h_used[1] >= sort_from) cst = bear_canonical_order(p->heavy_row, h_used[1], 16, 0);
    if (cst == BEAR_OK && h_used[2] >= sort_from) cst = bear_canonical_order(p->heavy_stop, h_used[2], 8, 0);
    if (cst != BEAR_OK) e = cst == BEAR_ERR_NOMEM ? hipErrorOutOfMemory : hipErrorUnknown;
  }
  p->n_live_rows = n_rows;
  if (e == hipSuccess && ncol == 5) {        // rows with a total of 1..SRT_CL (histogram) + rows with a larger one
    unsigned long long h_hist[SRT_NKEY];
    e = hipMemcpy(h_hist, p->hist, sizeof(h_hist), hipMemcpyDeviceToHost);
    unsigned long long live_rows = h_cnt[1];
    for (int k = 0; k < SRT_NKEY; ++k) live_rows += h_hist[k];
    p->n_live_rows = live_rows;
  }
  (void)hipFree(d_cnt);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    plan_free(p);
    return e == hipErrorOutOfMemory ? BEAR_ERR_NOMEM : BEAR_ERR_HIP;
  }
  p->bytes = stream_bytes + live_bytes + sizeof(pln_tile) * n_tiles + sizeof(pln_heavy_col) * h_cnt[0] +
             sizeof(pln_heavy_row) * h_cnt[1] + sizeof(uint64_t) * h_cnt[2];
  *out = p;
  return BEAR_OK;
}

int bear_plan_destroy(bear_plan *plan) {
  if (!plan) return BEAR_OK;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(plan->device);
  plan_free(plan);
  (void)hipSetDevice(prev);
  return BEAR_OK;
}

uint64_t bear_plan_bytes(const bear_plan *plan) { return plan ? plan->bytes : 0; }

// ---- the dense form: a five-column plan that keeps nothing per item (kernels_rows.h, dm_prior_rows_kernel).  Internally ncol =
// PLAN_ROWS: every entry point that walks a plan's tiles and lists asks for ncol == 5 and so turns such a plan away; the mode-N
// entry points (bear_dm_prior_plan_f64 / _grad_f64 / _dev_f64) take both.
#define PLAN_ROWS 15
static bool plan_is_rows(const bear_plan *p) { return p->ncol == PLAN_ROWS; }
int bear_plan_create_auto(bear_ws *ws, const uint32_t *counts, uint64_t n_rows, int *rowwise, bear_plan **out) {
  if (rowwise) *rowwise = 0;
  int st = bear_plan_create(ws, counts, n_rows, 5, out);
  if (st != BEAR_OK) return st;
  bear_plan *p = *out;
  // cells that hold a count (count_total[1]) against those the sorted encoding could not keep in its tiles (the global list of
  // large-count items): a table of large counts is all list
  const double cells = p->count_total[1], listed = (double)p->n_heavy[0];
  if (!(cells > 0.0) || listed * 2.0 <= cells) return BEAR_OK;
  (void)hipFree(p->tiles);
  (void)hipFree(p->stream);
  (void)hipFree(p->heavy_col);
  (void)hipFree(p->heavy_row);
  (void)hipFree(p->heavy_stop);
  (void)hipFree(p->live);
  p->tiles = nullptr;
  p->stream = nullptr;
  p->heavy_col = nullptr;
  p->heavy_row = nullptr;
  p->heavy_stop = nullptr;
  p->live = nullptr;
  p->n_tiles = 0;
  p->n_heavy[0] = p->n_heavy[1] = p->n_heavy[2] = 0;
  p->ncol = PLAN_ROWS;
  p->bytes = sizeof(unsigned long long) * (2 * SRT_NKEY + PLN_NBIG + 1);      // (the histograms stay: nothing else does)
  if (rowwise) *rowwise = 1;
  return BEAR_OK;
}
// mode N on such a plan: one launch (the last block sums), parameters by value or from device memory
static int launch_prior_rows(bear_ws *ws, const bear_plan *plan, const double *prior, const bear_params &prm, const double *theta,
                             int train_ar, double *out, double *grad_prior, hipStream_t s) {
  const uint64_t tiles = (plan->n_rows + DPR_TILE_ROWS - 1) / DPR_TILE_ROWS;
  uint64_t g = (uint64_t)ws->num_cu * DPR_BLOCKS_PER_CU;
  if (g > (uint64_t)ws->max_blocks) g = ws->max_blocks;
  const int grid = (int)(tiles < g ? (tiles ? tiles : 1) : g);
  const double2 *lt = reinterpret_cast<const double2 *>(ws->logtab);
  const bear_step_io io = ws_io(ws, theta, BEAR_THETA_NET, out);
#define ROWS_LAUNCH(AR, GRAD) \
  hipLaunchKernelGGL((dm_prior_rows_kernel<AR, GRAD>), dim3(grid), dim3(BEAR_THREADS), 0, s, plan->counts, prior, plan->n_rows, prm, grad_prior, lt, \
                     ws->partials, io)
  if (train_ar) {
    if (grad_prior) ROWS_LAUNCH(true, true);
    else ROWS_LAUNCH(true, false);
  } else {
    if (grad_prior) ROWS_LAUNCH(false, true);
    else ROWS_LAUNCH(false, false);
  }
#undef ROWS_LAUNCH
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

uint64_t bear_plan_tile_count(const bear_plan *plan) { return plan ? plan->n_tiles : 0; }

int bear_plan_tile_info(const bear_plan *plan, uint64_t first, uint64_t count, uint64_t *row0, uint32_t *rows, uint32_t *items,
                        uint64_t *stream_offset) {
  if (!plan || first + count > plan->n_tiles || (count && (!row0 || !rows || !items || !stream_offset))) return BEAR_ERR_INVALID_ARG;
  if (!count) return BEAR_OK;
  std::vector<pln_tile> h(count);
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(plan->device);
  const hipError_t e = hipMemcpy(h.data(), plan->tiles + first, sizeof(pln_tile) * count, hipMemcpyDeviceToHost);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return BEAR_ERR_HIP;
  }
  for (uint64_t k = 0; k < count; ++k) {
    row0[k] = h[k].row0;
    rows[k] = h[k].rows_items >> 16;
    items[k] = h[k].rows_items & 0xffffu;
    stream_offset[k] = (uint64_t)h[k].off16 * 16;
  }
  return BEAR_OK;
}

int bear_plan_create_ref(bear_ws *ws, const uint32_t *train, const uint32_t *ref, uint64_t n_rows, bear_plan **out) {
  if (!out) return BEAR_ERR_INVALID_ARG;
  *out = nullptr;
  if ((n_rows && !ref) || misaligned(ref)) return BEAR_ERR_INVALID_ARG;
  bear_plan *p = nullptr;
  int st = bear_plan_create(ws, train, n_rows, 4, &p);   // tiles, histograms of totals / stop counts, heavy lists
  if (st != BEAR_OK) return st;
  p->ref = ref;
  // a table of large counts (more than half of its cells beyond the sorted encoding's tiles: bear_plan_create_auto's test) keeps
  // nothing per item: the mode-R step then streams the training and reference rows, a context per thread (dm_ref_rows_kernel)
  if (p->count_total[1] > 0.0 && (double)p->n_heavy[0] * 2.0 > p->count_total[1]) {
    (void)hipFree(p->tiles);
    (void)hipFree(p->stream);
    (void)hipFree(p->heavy_col);
    (void)hipFree(p->heavy_row);
    (void)hipFree(p->heavy_stop);
    p->tiles = nullptr;
    p->stream = nullptr;
    p->heavy_col = nullptr;
    p->heavy_row = nullptr;
    p->heavy_stop = nullptr;
    p->n_tiles = 0;
    p->n_heavy[0] = p->n_heavy[1] = p->n_heavy[2] = 0;
    p->rows_ref = 1;
    p->bytes = sizeof(unsigned long long) * (2 * SRT_NKEY + PLN_NBIG + 1);
    *out = p;
    return BEAR_OK;
  }
  unsigned long long *d_meta = nullptr;   // [0..31] bucket sizes / cursors, [32..63] hist0, [64] n_heavy0
  hipError_t e = hipMalloc(&d_meta, sizeof(unsigned long long) * 72);
  if (e == hipSuccess) e = hipMemset(d_meta, 0, sizeof(unsigned long long) * 72);
  if (e == hipSuccess) e = hipMalloc(&p->sum0, sizeof(double));
  if (e == hipSuccess) e = hipMemset(p->sum0, 0, sizeof(double));
  p->hist0_base = d_meta;
  unsigned long long h_meta[72];
  memset(h_meta, 0, sizeof(h_meta));
  const uint64_t chunks = (n_rows + 1023) / 1024;
  const int grid = (int)(chunks < (uint64_t)ws->num_cu * 8 ? (chunks ? chunks : 1) : (uint64_t)ws->num_cu * 8);
  if (e == hipSuccess && n_rows) {
    hipLaunchKernelGGL(rpl_build_kernel, dim3(grid), dim3(256), 0, 0, train, ref, n_rows, 0, d_meta, d_meta + 32, d_meta + 64, p->sum0,
                       static_cast<rpl_item *>(nullptr), static_cast<uint32_t *>(nullptr));
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h_meta, d_meta, sizeof(h_meta), hipMemcpyDeviceToHost);
  }
  if (e == hipSuccess) {
    unsigned long long run = 0, cur[RPL_NKEY];
    for (int k = 0; k < RPL_NKEY; ++k) {
      cur[k] = run;
      run += h_meta[k];
    }
    p->n_ref_items = run;
    p->n_heavy0 = h_meta[64];
    if (run) e = hipMalloc(&p->ref_items, sizeof(rpl_item) * run);
    if (e == hipSuccess && p->n_heavy0) e = hipMalloc(&p->heavy0, sizeof(uint32_t) * p->n_heavy0);
    if (e == hipSuccess) e = hipMemcpy(d_meta, cur, sizeof(cur), hipMemcpyHostToDevice);      // sizes -> cursors
    if (e == hipSuccess) e = hipMemset(d_meta + 64, 0, sizeof(unsigned long long));
    if (e == hipSuccess && n_rows && (run || p->n_heavy0)) {
      hipLaunchKernelGGL(rpl_build_kernel, dim3(grid), dim3(256), 0, 0, train, ref, n_rows, 1, d_meta, d_meta + 32, d_meta + 64, p->sum0,
                         p->ref_items, p->heavy0);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
#ifdef BEAR_DET_BUILD      // the records of a bucket stand in the order of the blocks' cursor bumps: canonical order per bucket
    for (int k = 0; k < RPL_NKEY && e == hipSuccess; ++k)
      if (bear_canonical_order(p->ref_items + cur[k], h_meta[k], 16, 0) != BEAR_OK) e = hipErrorUnknown;
    if (e == hipSuccess && bear_canonical_order(p->heavy0, p->n_heavy0, 4, 0) != BEAR_OK) e = hipErrorUnknown;
#endif
  }
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    plan_free(p);
    return e == hipErrorOutOfMemory ? BEAR_ERR_NOMEM : BEAR_ERR_HIP;
  }
  p->hist0 = d_meta + 32;
  p->bytes += sizeof(rpl_item) * p->n_ref_items + sizeof(uint32_t) * p->n_heavy0 + sizeof(unsigned long long) * 72;
  *out = p;
  return BEAR_OK;
}

static pln_view plan_view(const bear_plan *p) {
  pln_view v;
  v.tiles = p->tiles;
  v.stream = p->stream;
  v.heavy_col = p->heavy_col;
  v.heavy_row = p->heavy_row;
  v.heavy_stop = p->heavy_stop;
  v.hist = p->hist;
  v.hist_big = p->hist + 2 * SRT_NKEY;
  v.big_in_hist = 1;
  v.live = p->live;
  v.live2 = p->live2;
  v.subset = 0;
  v.n_tiles = p->n_tiles;
  v.n_heavy_col = p->n_heavy[0];
  v.n_heavy_row = p->n_heavy[1];
  v.n_heavy_stop = p->n_heavy[2];
  return v;
}

static int grid_plan(const bear_ws *ws, uint64_t n_tiles) {
  uint64_t g = (uint64_t)ws->num_cu * PLN_BLOCKS_PER_CU;  // one resident 1024-thread block per CU (LDS ring; the half-tile build: two of 512)
  if (g > (uint64_t)ws->max_blocks) g = ws->max_blocks;
  if (n_tiles < g) g = n_tiles;
  return g < 1 ? 1 : (int)g;
}

// theta != NULL: the kernels derive their constants from the device-resident parameters (kind: BEAR_THETA_NET / _REF; prm.eps is
// still read from `prm`).  Either way ONE launch: the last block to finish writes the fixed-order sums to `out`.
static int launch_prior_plan(bear_ws *ws, const bear_plan *plan, const double *prior, uint64_t n_rows, const bear_params &prm,
                             const double *theta, int train_ar, int prior_normalized, double *out, hipStream_t s) {
  const int grid = grid_plan(ws, plan->n_tiles);
  const double2 *lt = reinterpret_cast<const double2 *>(ws->logtab);
  const bear_step_io io = ws_io(ws, theta, BEAR_THETA_NET, out);
  if (train_ar)
    hipLaunchKernelGGL((dm_prior_plan_kernel<true, true>), dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_n), s, prior, n_rows, prm,
                       plan_view(plan), lt, ws->partials, io PLN_DBG_ARG);
  else if (prior_normalized)
    hipLaunchKernelGGL((dm_prior_plan_kernel<true, false>), dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_n), s, prior, n_rows, prm,
                       plan_view(plan), lt, ws->partials, io PLN_DBG_ARG);
  else
    hipLaunchKernelGGL((dm_prior_plan_kernel<false, false>), dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_n), s, prior, n_rows, prm,
                       plan_view(plan), lt, ws->partials, io PLN_DBG_ARG);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_dm_prior_plan_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const double *prior,
                           uint64_t n_rows, double h_signed, double eps, int train_ar, int prior_normalized,
                           double *out, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !out || (n_rows && (!counts || !prior))) return BEAR_ERR_INVALID_ARG;
  if ((plan->ncol != 5 && !plan_is_rows(plan)) || plan->n_rows != n_rows || plan->counts != counts || plan->device != ws->device)
    return BEAR_ERR_INVALID_ARG;
  if (misaligned(prior) || (reinterpret_cast<uintptr_t>(out) & 7u)) return BEAR_ERR_INVALID_ARG;
  bear_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.inv_h = 1.0 / exp(h_signed);
  prm.eps = eps;
  if (plan_is_rows(plan)) return launch_prior_rows(ws, plan, prior, prm, nullptr, train_ar, out, nullptr, static_cast<hipStream_t>(stream));
  return launch_prior_plan(ws, plan, prior, n_rows, prm, nullptr, train_ar, prior_normalized, out, static_cast<hipStream_t>(stream));
}

static int launch_prior_plan_grad(bear_ws *ws, const bear_plan *plan, const double *prior, const bear_params &prm,
                                  const double *theta, int train_ar, int prior_normalized, double *out, double *grad_prior,
                                  hipStream_t s) {
  const int grid = grid_plan(ws, plan->n_tiles);
  const double2 *lt = reinterpret_cast<const double2 *>(ws->logtab);
  const pln_view pv = plan_view(plan);
  const bear_step_io io = ws_io(ws, theta, BEAR_THETA_NET, out);
  if (train_ar)
    hipLaunchKernelGGL((dm_prior_plan_grad_kernel<true, true>), dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_g), s, prior, prm, pv, lt,
                       grad_prior, ws->partials, io);
  else if (prior_normalized)   // rows asserted normalised: the in-place, double-buffered form
    hipLaunchKernelGGL(dm_prior_plan_grad_inplace_kernel, dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_gi), s, prior, prm, pv, lt,
                       grad_prior, ws->partials, io);
  else
    hipLaunchKernelGGL((dm_prior_plan_grad_kernel<false, false>), dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_g), s, prior, prm, pv, lt,
                       grad_prior, ws->partials, io);
  HIP_TRY(hipGetLastError());
  // the plan's global overflow lists (dense tables): the contexts' base, then the column items (kernels_plan.h)
  auto fixup_grid = [&](uint64_t nh) { return (int)((nh + 255) / 256 < (uint64_t)ws->num_cu * 8 ? (nh + 255) / 256 : (uint64_t)ws->num_cu * 8); };
#define FIXUP(NORM, AR, ROWS, N) \
  hipLaunchKernelGGL((dm_prior_grad_fixup_kernel<NORM, AR, ROWS>), dim3(fixup_grid(N)), dim3(256), 0, s, prior, prm, pv, lt, grad_prior, io)
  if (pv.n_heavy_row && !train_ar) {
    if (prior_normalized) FIXUP(true, false, true, pv.n_heavy_row);
    else FIXUP(false, false, true, pv.n_heavy_row);
  }
  if (pv.n_heavy_col) {
    if (train_ar) FIXUP(true, true, false, pv.n_heavy_col);
    else if (prior_normalized) FIXUP(true, false, false, pv.n_heavy_col);
    else FIXUP(false, false, false, pv.n_heavy_col);
  }
#undef FIXUP
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_dm_prior_plan_grad_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const double *prior,
                                uint64_t n_rows, double h_signed, double eps, int train_ar, int prior_normalized,
                                double *out, double *grad_prior, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !out || !grad_prior || (n_rows && (!counts || !prior))) return BEAR_ERR_INVALID_ARG;
  if ((plan->ncol != 5 && !plan_is_rows(plan)) || plan->n_rows != n_rows || plan->counts != counts || plan->device != ws->device)
    return BEAR_ERR_INVALID_ARG;
  if (misaligned(prior) || misaligned(grad_prior) || (reinterpret_cast<uintptr_t>(out) & 7u)) return BEAR_ERR_INVALID_ARG;
  bear_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.inv_h = 1.0 / exp(h_signed);
  prm.eps = eps;
  if (plan_is_rows(plan)) return launch_prior_rows(ws, plan, prior, prm, nullptr, train_ar, out, grad_prior, static_cast<hipStream_t>(stream));
  return launch_prior_plan_grad(ws, plan, prior, prm, nullptr, train_ar, prior_normalized, out, grad_prior,
                                static_cast<hipStream_t>(stream));
}

// h_signed read from device memory (a parameter the optimizer updates on the device): the step is enqueued without the host
// reading the parameter back.  grad_prior may be NULL (parameter-free ar_func: nothing to feed back).
int bear_dm_prior_plan_dev_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const double *prior, uint64_t n_rows,
                               const double *h_signed_dev, double eps, int train_ar, int prior_normalized, double *out,
                               double *grad_prior, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !out || !h_signed_dev || (n_rows && (!counts || !prior))) return BEAR_ERR_INVALID_ARG;
  if ((plan->ncol != 5 && !plan_is_rows(plan)) || plan->n_rows != n_rows || plan->counts != counts || plan->device != ws->device)
    return BEAR_ERR_INVALID_ARG;
  if (misaligned(prior) || misaligned(grad_prior) || (reinterpret_cast<uintptr_t>(out) & 7u)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_params only_eps;
  memset(&only_eps, 0, sizeof(only_eps));
  only_eps.eps = eps;
  if (plan_is_rows(plan)) return launch_prior_rows(ws, plan, prior, only_eps, h_signed_dev, train_ar, out, grad_prior, s);
  if (grad_prior) return launch_prior_plan_grad(ws, plan, prior, only_eps, h_signed_dev, train_ar, prior_normalized, out, grad_prior, s);
  return launch_prior_plan(ws, plan, prior, n_rows, only_eps, h_signed_dev, train_ar, prior_normalized, out, s);
}

// bear_ref's step for a net function with parameters: the reference mixing inside the DM step (kernels_mixplan.h)
int bear_dm_refmix_plan_grad_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const double *net_rows,
                                 const double *ref_rows, uint64_t n_rows, const double *h_signed_dev, const double *tau_signed_dev,
                                 const double *net_weight_signed_dev, double eps, int train_ar, double *out, double *grad_net_rows,
                                 void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !out || !h_signed_dev || !tau_signed_dev || !net_weight_signed_dev) return BEAR_ERR_INVALID_ARG;
  if (n_rows && (!counts || !net_rows || !ref_rows || !grad_net_rows)) return BEAR_ERR_INVALID_ARG;
  if (plan->ncol != 5 || plan->n_rows != n_rows || plan->counts != counts || plan->device != ws->device) return BEAR_ERR_INVALID_ARG;
  if (misaligned(net_rows) || misaligned(ref_rows) || misaligned(grad_net_rows) || misaligned8(out) || misaligned8(h_signed_dev) ||
      misaligned8(tau_signed_dev) || misaligned8(net_weight_signed_dev) || !(eps >= 0.0))
    return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = grid_plan(ws, plan->n_tiles);
  const double2 *lt = reinterpret_cast<const double2 *>(ws->logtab);
  const pln_view pv = plan_view(plan);
  const bear_step_io io = ws_io(ws, nullptr, BEAR_THETA_REF, out);
  if (train_ar)
    hipLaunchKernelGGL(dm_refmix_plan_grad_kernel<true>, dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_g), s, net_rows, ref_rows,
                       h_signed_dev, tau_signed_dev, net_weight_signed_dev, eps, pv, lt, grad_net_rows, ws->partials, io);
  else
    hipLaunchKernelGGL(dm_refmix_plan_grad_kernel<false>, dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_g), s, net_rows, ref_rows,
                       h_signed_dev, tau_signed_dev, net_weight_signed_dev, eps, pv, lt, grad_net_rows, ws->partials, io);
  HIP_TRY(hipGetLastError());
  if (pv.n_heavy_col + pv.n_heavy_row) {
    const uint64_t nh = pv.n_heavy_col + pv.n_heavy_row;
    const int g2 = (int)((nh + 255) / 256 < (uint64_t)ws->num_cu * 4 ? (nh + 255) / 256 : (uint64_t)ws->num_cu * 4);
    if (train_ar)
      hipLaunchKernelGGL(dm_refmix_fixup_kernel<true>, dim3(g2), dim3(256), 0, s, net_rows, ref_rows, h_signed_dev, tau_signed_dev,
                         net_weight_signed_dev, eps, pv, lt, grad_net_rows);
    else
      hipLaunchKernelGGL(dm_refmix_fixup_kernel<false>, dim3(g2), dim3(256), 0, s, net_rows, ref_rows, h_signed_dev, tau_signed_dev,
                         net_weight_signed_dev, eps, pv, lt, grad_net_rows);
    HIP_TRY(hipGetLastError());
  }
  return BEAR_OK;
}

// The mode-R step on a plan: the reference-aware item stream when the plan was built with this reference column
// (bear_plan_create_ref), the streaming kernel otherwise.  theta != NULL: constants from the device-resident parameters.
static const bear_apply_io NO_APPLY = {};      // theta == NULL: the launch only reduces

static int launch_ref_plan(bear_ws *ws, const bear_plan *plan, const uint32_t *ref, uint64_t n_rows, const bear_params &prm,
                           const double *theta, int train_ar, double *out, hipStream_t s, const bear_apply_io &apply = NO_APPLY) {
  const double2 *lt = reinterpret_cast<const double2 *>(ws->logtab);
  const bear_step_io io = ws_io(ws, theta, BEAR_THETA_REF, out);
  int grid;
  if (plan->rows_ref) {       // the dense form: rows streamed (kernels_rows.h)
    if (plan->ref != ref) return BEAR_ERR_INVALID_ARG;
    const uint64_t tiles = (n_rows + DPR_TILE_ROWS - 1) / DPR_TILE_ROWS;
    uint64_t g = (uint64_t)ws->num_cu * DPR_BLOCKS_PER_CU;
    if (g > (uint64_t)ws->max_blocks) g = ws->max_blocks;
    grid = (int)(tiles < g ? (tiles ? tiles : 1) : g);
    if (train_ar)
      hipLaunchKernelGGL(dm_ref_rows_kernel<true>, dim3(grid), dim3(BEAR_THREADS), 0, s, plan->counts, ref, n_rows, prm, lt, ws->partials, io, apply);
    else
      hipLaunchKernelGGL(dm_ref_rows_kernel<false>, dim3(grid), dim3(BEAR_THREADS), 0, s, plan->counts, ref, n_rows, prm, lt, ws->partials, io, apply);
  } else if (plan->ref) {
    if (plan->ref != ref) return BEAR_ERR_INVALID_ARG;   // the plan is valid for the reference buffer it was built from
    rpl_view rv;
    rv.items = plan->ref_items;
    rv.n_items = plan->n_ref_items;
    rv.hist0 = plan->hist0;
    rv.heavy0 = plan->heavy0;
    rv.n_heavy0 = plan->n_heavy0;
    rv.sum0 = plan->sum0;
    // 4 waves per block; blocks per CU by measurement (scripts/dev/cfg1_grid.py: a block's fixed work -- the log table into LDS, its
    // partials, two arrival atomics, its share of the last block's sum -- against the balance of more blocks): configs[1] (1e7
    // contexts, ~3e4 units) 25.3 / 23.9 / 22.2 / 23.4 / 27.3 us per step on 8 / 6 / 4 / 3 / 2 blocks per CU, configs[3] (1.25e8
    // contexts, ~4e5 units) 160 / 154 / 174 / 196 / 256 us
    const uint64_t units = (plan->n_ref_items + 63) / 64, want = (units + 3) / 4 + 2, cap = (uint64_t)ws->num_cu * (units > 131072 ? 6 : 4);
    grid = (int)(want < cap ? want : cap);
    if (train_ar)
      hipLaunchKernelGGL(dm_ref_items_kernel<true>, dim3(grid), dim3(256), 0, s, prm, rv, plan_view(plan), lt, ws->partials, io, apply);
    else
      hipLaunchKernelGGL(dm_ref_items_kernel<false>, dim3(grid), dim3(256), 0, s, prm, rv, plan_view(plan), lt, ws->partials, io, apply);
  } else {
    grid = grid_plan(ws, plan->n_tiles);
    if (train_ar)
      hipLaunchKernelGGL(dm_ref_plan_kernel<true>, dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_r), s, ref, n_rows, prm, plan_view(plan), lt,
                         ws->partials, io, apply);
    else
      hipLaunchKernelGGL(dm_ref_plan_kernel<false>, dim3(grid), dim3(PLN_THREADS), sizeof(pln_lds_r), s, ref, n_rows, prm, plan_view(plan), lt,
                         ws->partials, io, apply);
  }
  HIP_TRY(hipGetLastError());
  (void)grid;
  return BEAR_OK;
}

int bear_dm_ref_plan_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *train, const uint32_t *ref,
                         uint64_t n_rows, double h_signed, double tau_signed, double nu_signed, double eps,
                         int train_ar, double *out, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !out || (n_rows && (!train || !ref))) return BEAR_ERR_INVALID_ARG;
  if (plan->ncol != 4 || plan->n_rows != n_rows || plan->counts != train || plan->device != ws->device)
    return BEAR_ERR_INVALID_ARG;
  if (misaligned(ref) || (reinterpret_cast<uintptr_t>(out) & 7u)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_params prm;
  const double tau = exp(tau_signed), nw = exp(nu_signed);
  prm.inv_h = 1.0 / exp(h_signed);
  prm.eps = eps;
  prm.E = exp(-tau);
  prm.tauE = tau * prm.E;
  prm.tau = tau;
  prm.V = 1.0 / (nw + 1.0);
  prm.nw = nw;
  return launch_ref_plan(ws, plan, ref, n_rows, prm, nullptr, train_ar, out, s);
}

// ---- optimizer step in two halves: the shard's reduce (constants from theta -> planned kernel -> finalize into `packed`) and the
// apply (tf.keras Adam on theta from packed).  One rank runs them back to back (bear_*_train_step_f64, graph-capturable);
// several ranks put ONE all-reduce of `packed` between them (bear_net.py:278-290) -- no host round trip either way.
static bear_apply_io make_apply(double *theta, int n_theta, double *adam_m, double *adam_v, double *adam_t, double learning_rate,
                                double scale, int train_ar, double *loss_buf, uint64_t loss_cap) {
  bear_apply_io A;
  A.theta = theta;
  A.m = adam_m;
  A.v = adam_v;
  A.t_state = adam_t;
  A.loss_buf = loss_buf;
  A.loss_cap = (unsigned long long)loss_cap;
  A.lr = learning_rate;
  A.scale = scale;
  A.n_theta = n_theta;
  A.train_ar = train_ar;
  return A;
}

// BEAR_AMD_TWO_LAUNCH_STEP=1: bear_*_train_step_f64 as reduce + bear_train_apply_f64 again (two launches; tests compare the two forms)
static bool two_launch_step() {
  const char *e = getenv("BEAR_AMD_TWO_LAUNCH_STEP");
  return e && e[0] && e[0] != '0';
}

static int launch_train_apply(double *theta, int n_theta, const double *packed, double *adam_m, double *adam_v, double *adam_t,
                              double learning_rate, double scale, int train_ar, double *loss_buf, uint64_t loss_cap, hipStream_t s) {
  hipLaunchKernelGGL(adam_vec_kernel, dim3(1), dim3(1024), 0, s,
                     make_apply(theta, n_theta, adam_m, adam_v, adam_t, learning_rate, scale, train_ar, loss_buf, loss_cap), packed);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_train_apply_f64(double *theta, int n_theta, const double *packed, double *adam_m, double *adam_v, double *adam_t,
                         double learning_rate, double scale, int train_ar, double *loss_buf, uint64_t loss_cap, void *stream) {
  if (!theta || !packed || !adam_m || !adam_v || !adam_t || n_theta < 1) return BEAR_ERR_INVALID_ARG;
  return launch_train_apply(theta, n_theta, packed, adam_m, adam_v, adam_t, learning_rate, scale, train_ar, loss_buf, loss_cap,
                            static_cast<hipStream_t>(stream));
}

int bear_ref_train_reduce_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *train, const uint32_t *ref, uint64_t n_rows,
                              const double *theta, double eps, int train_ar, double *packed, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !packed || !theta || !n_rows || !train || !ref) return BEAR_ERR_INVALID_ARG;
  if (plan->ncol != 4 || plan->n_rows != n_rows || plan->counts != train || plan->device != ws->device)
    return BEAR_ERR_INVALID_ARG;
  if (misaligned(ref) || (reinterpret_cast<uintptr_t>(packed) & 7u)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_params only_eps;
  memset(&only_eps, 0, sizeof(only_eps));
  only_eps.eps = eps;
  return launch_ref_plan(ws, plan, ref, n_rows, only_eps, theta, train_ar, packed, s);
}

int bear_ref_train_step_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *train, const uint32_t *ref, uint64_t n_rows,
                            double *theta, double *adam_m, double *adam_v, double *adam_t, double eps, int train_ar,
                            double learning_rate, double scale, double *out, double *loss_buf, uint64_t loss_cap, void *stream) {
  if (!adam_m || !adam_v || !adam_t) return BEAR_ERR_INVALID_ARG;
  if (two_launch_step()) {
    int st = bear_ref_train_reduce_f64(ws, plan, train, ref, n_rows, theta, eps, train_ar, out, stream);
    if (st != BEAR_OK) return st;
    return launch_train_apply(theta, 3, out, adam_m, adam_v, adam_t, learning_rate, scale, train_ar, loss_buf, loss_cap,
                              static_cast<hipStream_t>(stream));
  }
  // ONE launch: the last block of the reduce kernel runs the update behind its sums (bear_apply_in_block)
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !out || !theta || !n_rows || !train || !ref) return BEAR_ERR_INVALID_ARG;
  if (plan->ncol != 4 || plan->n_rows != n_rows || plan->counts != train || plan->device != ws->device) return BEAR_ERR_INVALID_ARG;
  if (misaligned(ref) || (reinterpret_cast<uintptr_t>(out) & 7u)) return BEAR_ERR_INVALID_ARG;
  bear_params only_eps;
  memset(&only_eps, 0, sizeof(only_eps));
  only_eps.eps = eps;
  return launch_ref_plan(ws, plan, ref, n_rows, only_eps, theta, train_ar, out, static_cast<hipStream_t>(stream),
                         make_apply(theta, 3, adam_m, adam_v, adam_t, learning_rate, scale, train_ar, loss_buf, loss_cap));
}

int bear_dm_items_f64(bear_ws *ws, const double *x, const uint32_t *c, uint64_t n, int path, double *D, double *P,
                      void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (n == 0) return BEAR_OK;
  if (!x || !c || !D || !P || path < 0 || path > 2) return BEAR_ERR_INVALID_ARG;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffffull) return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(dm_items_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, c, n, path,
                     reinterpret_cast<const double2 *>(ws->logtab), D, P);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

#ifdef CNN_STAMPS
extern "C" int bear_dbg_cnn_stamps(unsigned long long *host_out, int reset) {   // developer build only
  if (host_out) HIP_TRY(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cnn_stamp_sums), sizeof(unsigned long long) * 8));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(cnn_stamp_sums), z, sizeof(z)));
  }
  return BEAR_OK;
}
#endif

#ifdef LIN_STAMPS
extern "C" int bear_dbg_lin_stamps(unsigned long long *host_out, int reset) {   // developer build only
  if (host_out) HIP_TRY(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lin_stamp_sums), sizeof(unsigned long long) * 8));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lin_stamp_sums), z, sizeof(z)));
  }
  return BEAR_OK;
}
extern "C" int bear_dbg_lin_pe_stamps(unsigned long long *host_out) {   // prologue / epilogue sections of the last launch (12 words)
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lin_pe_stamps), sizeof(unsigned long long) * 12));
  return BEAR_OK;
}
#endif

// ---- fused linear AR head (kernels_linear.h) -----------------------------------------------------------
int bear_pack_kmers_u64(const int8_t *codes, uint64_t n_rows, int lag, uint64_t *packed, void *stream) {
  if (lag < 1 || lag > LIN_MAX_LAG) return BEAR_ERR_INVALID_ARG;
  if (n_rows == 0) return BEAR_OK;
  if (!codes || !packed) return BEAR_ERR_INVALID_ARG;
  const uint64_t blocks = (n_rows + 255) / 256;
  if (blocks > 0x7fffffffull) return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(pack_kmers_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), codes, n_rows, lag,
                     reinterpret_cast<unsigned long long *>(packed));
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_linear_index_u64(const uint64_t *kmer_code, uint64_t n_rows, int lag, uint64_t *kmer_index, void *stream) {
  if (lag < 1 || lag > LIN_MAX_LAG) return BEAR_ERR_INVALID_ARG;
  if (n_rows == 0) return BEAR_OK;
  if (!kmer_code || !kmer_index) return BEAR_ERR_INVALID_ARG;
  uint64_t blocks = (n_rows + 255) / 256;
  if (blocks > 1u << 20) blocks = 1u << 20;
  hipLaunchKernelGGL(linear_index_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const unsigned long long *>(kmer_code), n_rows, lag, reinterpret_cast<unsigned long long *>(kmer_index));
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_encode_kmers_i8(const uint8_t *ascii, uint64_t n_rows, int lag, int rna, int8_t *codes, void *stream) {
  if (lag < 1) return BEAR_ERR_INVALID_ARG;
  if (n_rows == 0) return BEAR_OK;
  if (!ascii || !codes) return BEAR_ERR_INVALID_ARG;
  const uint64_t n_bytes = n_rows * (uint64_t)lag;
  uint64_t blocks = (n_bytes + 255) / 256;
  if (blocks > 1u << 20) blocks = 1u << 20;
  hipLaunchKernelGGL(encode_kmers_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), ascii, n_bytes, rna,
                     codes);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

// The fused step's launch.  A plan paired for exactly these index words (bear_plan_pair_contexts): the PAIRED form of the kernel
// over the paired tiles and, when some tiles kept their plain list, a second launch of the plain form over those, which adds
// its sums to the first one's (same stream: the workspace is free again when it starts).
static void launch_linear(bear_ws *ws, const bear_plan *plan, const uint64_t *kmer_code, const double *mat, int lag, const bear_params &prm,
                          int train_ar, const bear_step_io &io, double *grad_mat, hipStream_t s, const bear_apply_io &apply = NO_APPLY) {
  const double2 *lt = reinterpret_cast<const double2 *>(ws->logtab);
  const unsigned long long *kc = reinterpret_cast<const unsigned long long *>(kmer_code);
  const bool paired = plan->live2 && plan->pair_codes == kmer_code && plan->pair_lag == lag && !getenv("BEAR_AMD_LINEAR_UNPAIRED");
  // BEAR_AMD_DETERMINISTIC: fixed-point gradient tables (kernels_linear.h, lin_fx); the kernel derives their scale from these bounds
  const bool det = bear_deterministic() && plan->count_bound[0] >= 1.0 && plan->count_bound[0] < 0x1p50;
  const lin_fx_bound gt_bound = {plan->count_bound[0], plan->count_bound[1], log(plan->count_bound[2] > 1.0 ? plan->count_bound[2] : 1.0)};
  // (the update, if any, goes with the step's LAST launch: the one that completes the sums)
#define LIN_LAUNCH_K(AR, PAIRED, DET, NGK, PV, NT, ACC)                                                                                     \
  hipLaunchKernelGGL((dm_linear_plan_kernel<AR, PAIRED, DET, NGK>), dim3(grid_plan(ws, NT)), dim3(PLN_THREADS), sizeof(pln_lds_lin), s, kc, \
                     mat, lag, prm, PV, lt, ws->partials, ws->lin_accum, (ACC) == 1 ? io2 : io, grad_mat, ACC, gt_bound,                    \
                     ((ACC) == 1 || !two_launches) ? apply : NO_APPLY)
  // the group count as a compile-time constant for the lags 12 / 13 (6 groups: BASELINE's k = 13), 14 / 15 (7) and 4 / 5 (2: the
  // bundled table); every other lag takes the kernel that finds it at run time (distinct13: 0.847 -> 0.833 ms, 128 -> 97 registers)
#define LIN_LAUNCH_D(AR, PAIRED, DET, PV, NT, ACC)                       \
  do {                                                                   \
    if (n_groups == 6) LIN_LAUNCH_K(AR, PAIRED, DET, 6, PV, NT, ACC);    \
    else if (n_groups == 7) LIN_LAUNCH_K(AR, PAIRED, DET, 7, PV, NT, ACC); \
    else if (n_groups == 2) LIN_LAUNCH_K(AR, PAIRED, DET, 2, PV, NT, ACC); \
    else LIN_LAUNCH_K(AR, PAIRED, DET, 0, PV, NT, ACC);                  \
  } while (0)
#define LIN_LAUNCH(AR, PAIRED, PV, NT, ACC)                    \
  do {                                                         \
    if (det) LIN_LAUNCH_D(AR, PAIRED, true, PV, NT, ACC);      \
    else LIN_LAUNCH_D(AR, PAIRED, false, PV, NT, ACC);         \
  } while (0)
  bear_step_io io2 = io;      // the second launch of a step: its own stamp on the arrival word
  io2.epoch = ws_arrival(ws).epoch;
  pln_view pv = plan_view(plan);
  const bool two_launches = paired && plan->n_tiles_u != 0;
  // (BEAR_AMD_LINEAR_GENERIC=1: always the kernel that takes the group count at run time; tests compare the two)
  const int n_groups = getenv("BEAR_AMD_LINEAR_GENERIC") ? 0 : lin_make_geom(lag).ng;
  if (!paired) {
    if (train_ar) LIN_LAUNCH(true, false, pv, plan->n_tiles, 0);
    else LIN_LAUNCH(false, false, pv, plan->n_tiles, 0);
    return;
  }
  pln_view pp = pv;           // the paired tiles; the plan's global lists and histogram go with this launch
  pp.tiles = plan->tiles_p;
  pp.n_tiles = plan->n_tiles_p;
  pp.subset = 1;
  if (two_launches) {
    if (train_ar) LIN_LAUNCH(true, true, pp, plan->n_tiles_p, 2);
    else LIN_LAUNCH(false, true, pp, plan->n_tiles_p, 2);
  } else if (train_ar) LIN_LAUNCH(true, true, pp, plan->n_tiles_p, 0);
  else LIN_LAUNCH(false, true, pp, plan->n_tiles_p, 0);
  if (plan->n_tiles_u == 0) return;
  pln_view pu = pv;           // the rest: tiles only
  pu.tiles = plan->tiles_u;
  pu.n_tiles = plan->n_tiles_u;
  pu.subset = 1;
  pu.n_heavy_col = pu.n_heavy_row = pu.n_heavy_stop = 0;
  pu.hist = nullptr;
  pu.hist_big = nullptr;
  if (train_ar) LIN_LAUNCH(true, false, pu, plan->n_tiles_u, 1);
  else LIN_LAUNCH(false, false, pu, plan->n_tiles_u, 1);
#undef LIN_LAUNCH
#undef LIN_LAUNCH_D
#undef LIN_LAUNCH_K
}

static void plan_unpair(bear_plan *plan) {
  if (!plan->live2) return;
  plan->bytes -= plan->n_tiles * LIN_LIVE2_STRIDE * sizeof(uint16_t) + (plan->n_tiles + 2 * PLN_DESC_PAD) * sizeof(pln_tile);
  (void)hipFree(plan->live2);
  (void)hipFree(plan->tiles_p);
  (void)hipFree(plan->tiles_u);
  plan->live2 = nullptr;
  plan->tiles_p = plan->tiles_u = nullptr;
  plan->n_tiles_p = plan->n_tiles_u = 0;
  plan->pair_codes = nullptr;
}

// Pairs the contexts of every tile's list for the fused linear step (kernels_linear.h, LIN_PAIR_CAP): kmer_index are the index
// words the step will be called with (bear_linear_index_u64 for `lag`), in the row order of the plan's count slab.
int bear_plan_pair_contexts(bear_plan *plan, const uint64_t *kmer_index, int lag, int *paired, void *stream) {
  if (paired) *paired = 0;
  if (!plan || plan->ncol != 5 || lag < 1 || lag > LIN_MAX_LAG) return BEAR_ERR_INVALID_ARG;
  if (plan->n_rows && (!kmer_index || misaligned(kmer_index))) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (plan->live2) {           // a plan holds one pairing: the new one replaces it
    HIP_TRY(hipStreamSynchronize(s));
    plan_unpair(plan);
  }
  const uint64_t nt = plan->n_tiles;
  if (nt == 0 || !plan->live) return BEAR_OK;
  uint16_t *live2 = nullptr, *n_ent_dev = nullptr;
  pln_tile *tp = nullptr, *tu = nullptr;
  std::vector<uint16_t> n_ent;
  std::vector<pln_tile> host, hp, hu;
  try {
    n_ent.resize(nt);
    host.resize(nt);
  } catch (const std::bad_alloc &) {
    return BEAR_ERR_NOMEM;
  }
  hipError_t e = hipMalloc(&live2, nt * LIN_LIVE2_STRIDE * sizeof(uint16_t));
  if (e == hipSuccess) e = hipMalloc(&n_ent_dev, nt * sizeof(uint16_t));
  if (e == hipSuccess) {
    uint64_t blocks = nt;                   // one wave per tile
    if (blocks > (1u << 18)) blocks = 1u << 18;
    hipLaunchKernelGGL(plan_pair_kernel, dim3((unsigned)blocks), dim3(64), 0, s, plan->tiles, nt, plan->live,
                       reinterpret_cast<const unsigned long long *>(kmer_index), lag, live2, n_ent_dev,
                       getenv("BEAR_AMD_PAIR_NO_EMPTY") ? 0 : 1);     // (developer switch: the dealt order without the extra empty slots)
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(n_ent.data(), n_ent_dev, nt * sizeof(uint16_t), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(host.data(), plan->tiles, nt * sizeof(pln_tile), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(n_ent_dev);
  uint64_t n_p = 0;
  bool keep = false;
  if (e == hipSuccess) {
    try {
      for (uint64_t t = 0; t < nt; ++t) {
        pln_tile d = host[t];
        const bool fits = n_ent[t] != 0xffffu;         // (a tile without live contexts fits with 0 entries)
        d.pad = (t << 32) | (fits ? n_ent[t] : 0u);
        (fits ? hp : hu).push_back(d);
      }
      n_p = hp.size();
      // fewer than half of the tiles paired (a sparse table: runs of one context): nothing to gain, the plan stays as it was
      keep = 2 * n_p >= nt;
      if (keep) {
        const pln_tile zero = {};
        hp.insert(hp.end(), PLN_DESC_PAD, zero);
        hu.insert(hu.end(), PLN_DESC_PAD, zero);
      }
    } catch (const std::bad_alloc &) {
      (void)hipFree(live2);
      return BEAR_ERR_NOMEM;
    }
  }
  if (e == hipSuccess && keep) e = hipMalloc(&tp, hp.size() * sizeof(pln_tile));
  if (e == hipSuccess && keep) e = hipMalloc(&tu, hu.size() * sizeof(pln_tile));
  if (e == hipSuccess && keep) e = hipMemcpy(tp, hp.data(), hp.size() * sizeof(pln_tile), hipMemcpyHostToDevice);
  if (e == hipSuccess && keep) e = hipMemcpy(tu, hu.data(), hu.size() * sizeof(pln_tile), hipMemcpyHostToDevice);
  if (e != hipSuccess || !keep) {
    (void)hipFree(live2);
    (void)hipFree(tp);
    (void)hipFree(tu);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      return e == hipErrorOutOfMemory ? BEAR_ERR_NOMEM : BEAR_ERR_HIP;
    }
    return BEAR_OK;
  }
  plan->live2 = live2;
  plan->tiles_p = tp;
  plan->tiles_u = tu;
  plan->n_tiles_p = n_p;
  plan->n_tiles_u = nt - n_p;
  plan->pair_codes = kmer_index;
  plan->pair_lag = lag;
  plan->bytes += nt * LIN_LIVE2_STRIDE * sizeof(uint16_t) + (nt + 2 * PLN_DESC_PAD) * sizeof(pln_tile);
  if (paired) *paired = 1;
  return BEAR_OK;
}

int bear_plan_count_total(const bear_plan *plan, double *total, double *bound) {
  if (!plan) return BEAR_ERR_INVALID_ARG;
  for (int k = 0; k < 3; ++k) {
    if (total) total[k] = plan->count_total[k];
    if (bound) bound[k] = plan->count_bound[k];
  }
  return BEAR_OK;
}

int bear_plan_set_count_bound(bear_plan *plan, const double *bound) {
  if (!plan || !bound) return BEAR_ERR_INVALID_ARG;
  for (int k = 0; k < 3; ++k)
    if (!(bound[k] >= plan->count_total[k]) || !(bound[k] < 0x1p50)) return BEAR_ERR_INVALID_ARG;
  for (int k = 0; k < 3; ++k) plan->count_bound[k] = bound[k];
  return BEAR_OK;
}

int bear_plan_pair_info(const bear_plan *plan, uint64_t *paired_tiles, uint64_t *plain_tiles) {
  if (!plan) return BEAR_ERR_INVALID_ARG;
  if (paired_tiles) *paired_tiles = plan->live2 ? plan->n_tiles_p : 0;
  if (plain_tiles) *plain_tiles = plan->live2 ? plan->n_tiles_u : plan->n_tiles;
  return plan->live2 ? 1 : 0;
}

int bear_dm_linear_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_code,
                       const double *mat, int lag, uint64_t n_rows, double h_signed, double eps, int train_ar,
                       double *out, double *grad_mat, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !out || !grad_mat || !mat || lag < 1 || lag > LIN_MAX_LAG) return BEAR_ERR_INVALID_ARG;
  if (plan->counts != counts || plan->n_rows != n_rows || plan->ncol != 5 || plan->device != ws->device) return BEAR_ERR_INVALID_ARG;
  if ((n_rows && !kmer_code) || misaligned(kmer_code) || (reinterpret_cast<uintptr_t>(out) & 7u)) return BEAR_ERR_INVALID_ARG;
  if (!(eps >= 0.0) || !isfinite(h_signed)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.inv_h = 1.0 / exp(h_signed);
  prm.eps = eps;
  const bear_step_io io = ws_io(ws, nullptr, BEAR_THETA_NET, out);   // one launch: the last block sums the partials
  launch_linear(ws, plan, kmer_code, mat, lag, prm, train_ar, io, grad_mat, s);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

// ---- the linear AR function as rows (kernels_linrows.h): evaluation, bear_ref with the linear net function ------------
static int linrows_grid(const bear_ws *ws, uint64_t n_rows) {
  uint64_t blocks = (n_rows + LNR_THREADS - 1) / LNR_THREADS;
  if (blocks > (uint64_t)ws->num_cu) blocks = (uint64_t)ws->num_cu;   // lin_partials holds num_cu blocks
  return blocks ? (int)blocks : 1;
}

int bear_linear_forward_f64(bear_ws *ws, const uint64_t *kmer_code, uint64_t n_rows, int lag, const double *mat, double *prior,
                            void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (lag < 1 || lag > LIN_MAX_LAG || !mat) return BEAR_ERR_INVALID_ARG;
  if (n_rows == 0) return BEAR_OK;
  if (!kmer_code || !prior || misaligned(prior) || (reinterpret_cast<uintptr_t>(kmer_code) & 7u) || (reinterpret_cast<uintptr_t>(mat) & 7u))
    return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(linear_rows_forward_kernel, dim3(linrows_grid(ws, n_rows)), dim3(LNR_THREADS), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const unsigned long long *>(kmer_code), n_rows, mat, lag, prior);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_linear_backward_f64(bear_ws *ws, const uint64_t *kmer_code, uint64_t n_rows, int lag, const double *prior,
                             const double *grad_prior, double *grad_mat, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (lag < 1 || lag > LIN_MAX_LAG || !grad_mat || (reinterpret_cast<uintptr_t>(grad_mat) & 7u)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_rows == 0) {
    HIP_TRY(hipMemsetAsync(grad_mat, 0, sizeof(double) * (size_t)lag * 25, s));
    return BEAR_OK;
  }
  if (!kmer_code || !prior || !grad_prior) return BEAR_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(kmer_code) | reinterpret_cast<uintptr_t>(prior) | reinterpret_cast<uintptr_t>(grad_prior)) & 7u)
    return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(linear_rows_backward_kernel, dim3(linrows_grid(ws, n_rows)), dim3(LNR_THREADS), 0, s,
                     reinterpret_cast<const unsigned long long *>(kmer_code), n_rows, lag, prior, grad_prior, ws->lin_partials,
                     ws_arrival(ws), grad_mat);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

// ---- bear_ref's prior rows for a parametrised net function (kernels_refmix.h) ---------------------------------------------
static int refmix_grid(const bear_ws *ws, uint64_t n_rows) {
  uint64_t blocks = (n_rows + RMX_THREADS - 1) / RMX_THREADS;
  uint64_t cap = (uint64_t)ws->num_cu * 2;
  if (cap > (uint64_t)ws->max_blocks) cap = (uint64_t)ws->max_blocks;
  if (blocks > cap) blocks = cap;
  return blocks ? (int)blocks : 1;
}

int bear_ref_mix_forward_f64(bear_ws *ws, const double *net_rows, const double *ref_rows, uint64_t n_rows, const double *tau_signed,
                             const double *net_weight_signed, double *prior, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!tau_signed || !net_weight_signed || misaligned8(tau_signed) || misaligned8(net_weight_signed)) return BEAR_ERR_INVALID_ARG;
  if (n_rows == 0) return BEAR_OK;
  if (!net_rows || !ref_rows || !prior || misaligned(net_rows) || misaligned(ref_rows) || misaligned(prior)) return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(ref_mix_forward_kernel, dim3(refmix_grid(ws, n_rows)), dim3(RMX_THREADS), 0, static_cast<hipStream_t>(stream),
                     net_rows, ref_rows, tau_signed, net_weight_signed, n_rows, prior);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_ref_mix_backward_f64(bear_ws *ws, const double *net_rows, const double *ref_rows, const double *grad_prior, uint64_t n_rows,
                              const double *tau_signed, const double *net_weight_signed, double *grad_net_rows, double *grad_scalars,
                              void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!tau_signed || !net_weight_signed || !grad_scalars || misaligned8(tau_signed) || misaligned8(net_weight_signed) ||
      misaligned8(grad_scalars))
    return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_rows == 0) {
    HIP_TRY(hipMemsetAsync(grad_scalars, 0, 2 * sizeof(double), s));
    return BEAR_OK;
  }
  if (!net_rows || !ref_rows || !grad_prior || !grad_net_rows || misaligned(net_rows) || misaligned(ref_rows) ||
      misaligned(grad_prior) || misaligned(grad_net_rows))
    return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(ref_mix_backward_kernel, dim3(refmix_grid(ws, n_rows)), dim3(RMX_THREADS), 0, s, net_rows, ref_rows, grad_prior,
                     tau_signed, net_weight_signed, n_rows, grad_net_rows, ws->partials, ws_arrival(ws), grad_scalars);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

// ---- held-out evaluation / BMM marginal (kernels_eval.h) ------------------------------------------------
int bear_net_linear_train_reduce_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_code, int lag,
                                     uint64_t n_rows, const double *theta, double eps, int train_ar, double *packed, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !packed || !theta || lag < 1 || lag > LIN_MAX_LAG || !n_rows) return BEAR_ERR_INVALID_ARG;
  if (plan->counts != counts || plan->n_rows != n_rows || plan->ncol != 5 || plan->device != ws->device) return BEAR_ERR_INVALID_ARG;
  if (!kmer_code || misaligned(kmer_code) || (reinterpret_cast<uintptr_t>(packed) & 7u)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_params dummy;
  memset(&dummy, 0, sizeof(dummy));
  dummy.eps = eps;
  const double *mat = theta + 1;
  const bear_step_io io = ws_io(ws, theta, BEAR_THETA_NET, packed);   // constants from theta, sums by the last block: one launch
  launch_linear(ws, plan, kmer_code, mat, lag, dummy, train_ar, io, packed + 2, s);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_net_linear_train_step_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_code, int lag,
                                   uint64_t n_rows, double *theta, double *adam_m, double *adam_v, double *adam_t, double *packed,
                                   double eps, int train_ar, double learning_rate, double scale, double *loss_buf,
                                   uint64_t loss_cap, void *stream) {
  if (!adam_m || !adam_v || !adam_t) return BEAR_ERR_INVALID_ARG;
  if (two_launch_step()) {
    int st = bear_net_linear_train_reduce_f64(ws, plan, counts, kmer_code, lag, n_rows, theta, eps, train_ar, packed, stream);
    if (st != BEAR_OK) return st;
    return launch_train_apply(theta, 1 + lag * 25, packed, adam_m, adam_v, adam_t, learning_rate, scale, train_ar, loss_buf, loss_cap,
                              static_cast<hipStream_t>(stream));
  }
  // ONE launch: the last block of the step's (last) kernel runs the update behind its sums
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || !packed || !theta || lag < 1 || lag > LIN_MAX_LAG || !n_rows) return BEAR_ERR_INVALID_ARG;
  if (plan->counts != counts || plan->n_rows != n_rows || plan->ncol != 5 || plan->device != ws->device) return BEAR_ERR_INVALID_ARG;
  if (!kmer_code || misaligned(kmer_code) || (reinterpret_cast<uintptr_t>(packed) & 7u)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_params dummy;
  memset(&dummy, 0, sizeof(dummy));
  dummy.eps = eps;
  const bear_step_io io = ws_io(ws, theta, BEAR_THETA_NET, packed);
  launch_linear(ws, plan, kmer_code, theta + 1, lag, dummy, train_ar, io, packed + 2, s,
                make_apply(theta, 1 + lag * 25, adam_m, adam_v, adam_t, learning_rate, scale, train_ar, loss_buf, loss_cap));
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

static int launch_eval(bear_ws *ws, const uint32_t *test, const uint32_t *train, const double *prior, uint64_t n_rows,
                       const evl_args &A, double *out, hipStream_t s) {
  // sorted formulation (kernels_eval.h): at most EVS_CHUNK DM models per launch; the first launch also carries the AR
  // model and the total length.  Output slots: ll_ear[n_h], ll_arm, ll_van[n_van], cor_ear[n_h], cor_arm, cor_van[n_van], total
  const int n_models = A.n_h + A.n_van;
  const uint64_t tiles = (n_rows + EVS_THREADS - 1) / EVS_THREADS;
  const int grid = (int)(tiles < (uint64_t)ws->eval_blocks ? (tiles ? tiles : 1) : (uint64_t)ws->eval_blocks);
  static_assert(EVS_NOUT <= EVL_MAX_OUT, "compact partials fit the evaluation partial buffer");
  for (int m0 = 0; m0 == 0 || m0 < n_models; m0 += EVS_CHUNK) {
    const int m_cnt = n_models - m0 < EVS_CHUNK ? n_models - m0 : EVS_CHUNK;
    const int common = m0 == 0;
    evs_slots S;
    for (int k = 0; k < EVS_NOUT; ++k) S.slot[k] = -1;
    for (int k = 0; k < m_cnt; ++k) {
      const int m = m0 + k;
      const int ll_slot = m < A.n_h ? m : m + 1;                 // ll_arm sits between the BEAR and vanilla blocks
      S.slot[k] = ll_slot;
      S.slot[EVS_CHUNK + k] = n_models + 1 + ll_slot;
    }
    if (common) {
      S.slot[2 * EVS_CHUNK] = A.n_h;
      S.slot[2 * EVS_CHUNK + 1] = n_models + 1 + A.n_h;
      S.slot[2 * EVS_CHUNK + 2] = 2 * n_models + 2;
    }
    hipLaunchKernelGGL(eval_sorted_kernel, dim3(grid), dim3(EVS_THREADS), 0, s, test, train, prior, n_rows, A, m0, m_cnt, common,
                       reinterpret_cast<const double2 *>(ws->logtab), ws->eval_partials);
    hipLaunchKernelGGL(eval_sorted_finalize_kernel, dim3((EVS_NOUT + 3) / 4), dim3(256), 0, s, ws->eval_partials, grid, S, out);
  }
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}
static int eval_make_args(const uint32_t *test, const uint32_t *train, const double *prior, uint64_t n_rows, const double *h, int n_h,
                          int with_ar, const double *van_reg, int n_van, double eps, uint64_t noise_seed, uint64_t row_base,
                          const double *out, evl_args *Aout) {
  if (!out || n_h < 0 || n_van < 0 || n_h + n_van > EVL_MAX_MODELS || (n_h && !h) || (n_van && !van_reg))
    return BEAR_ERR_INVALID_ARG;
  if ((n_h || with_ar) && !prior && n_rows) return BEAR_ERR_INVALID_ARG;
  if (n_rows && !test) return BEAR_ERR_INVALID_ARG;
  if (misaligned(test) || misaligned(train) || misaligned(prior)) return BEAR_ERR_INVALID_ARG;
  if (!(eps >= 0.0)) return BEAR_ERR_INVALID_ARG;
  evl_args &A = *Aout;
  memset(&A, 0, sizeof(A));
  A.n_h = n_h;
  A.n_van = n_van;
  A.arm = with_ar ? 1 : 0;
  A.has_train = train ? 1 : 0;
  A.has_prior = prior ? 1 : 0;
  A.eps = eps;
  A.seed = noise_seed;
  A.row_base = row_base;
  for (int j = 0; j < n_h; ++j) {
    if (!(h[j] > 0.0)) return BEAR_ERR_INVALID_ARG;
    A.inv_h[j] = 1.0 / h[j];
  }
  for (int k = 0; k < n_van; ++k) A.inv_h[n_h + k] = van_reg[k];
  return BEAR_OK;
}

int bear_eval_f64(bear_ws *ws, const uint32_t *test, const uint32_t *train, const double *prior, uint64_t n_rows,
                  const double *h, int n_h, int with_ar, const double *van_reg, int n_van, double eps,
                  uint64_t noise_seed, uint64_t row_base, double *out, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  evl_args A;
  st = eval_make_args(test, train, prior, n_rows, h, n_h, with_ar, van_reg, n_van, eps, noise_seed, row_base, out, &A);
  if (st != BEAR_OK) return st;
  return launch_eval(ws, test, train, prior, n_rows, A, out, static_cast<hipStream_t>(stream));
}

// ---- evaluation on a sorted plan of the test column (kernels_evalplan.h) ------------------------------------------------
struct bear_eval_plan {
  int device;
  uint64_t n_rows, n_tiles;
  const uint32_t *test, *train;  // the buffers the plan was built from (identity check only; train may be NULL)
  uint16_t *items;       // [n_tiles][EVP_ITEMS_CAP]
  uint2 *tile_info;      // [n_tiles]
  unsigned long long *consts;   // [EVP_NCONST]: what the vanilla models, the total length need of the table as a whole (kernels_evalplan.h, EVP_C_*)
  uint64_t bytes;
};

int bear_eval_plan_create(bear_ws *ws, const uint32_t *test, const uint32_t *train, uint64_t n_rows, bear_eval_plan **out, void *stream) {
  if (!out) return BEAR_ERR_INVALID_ARG;
  *out = nullptr;
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if ((n_rows && !test) || misaligned(test) || misaligned(train)) return BEAR_ERR_INVALID_ARG;
  bear_eval_plan *p = new (std::nothrow) bear_eval_plan();
  if (!p) return BEAR_ERR_NOMEM;
  memset(p, 0, sizeof(*p));
  p->device = ws->device;
  p->n_rows = n_rows;
  p->test = test;
  p->train = train;
  p->n_tiles = (n_rows + EVP_ROWS - 1) / EVP_ROWS;
  if (p->n_tiles) {
    // + 1 KiB: the last DMA piece of a tile's lists may be issued for a partial KiB
    const size_t ibytes = sizeof(uint16_t) * EVP_ITEMS_CAP * (size_t)p->n_tiles + 1024;
    hipError_t e = hipMalloc(&p->items, ibytes);
    if (e == hipSuccess) e = hipMalloc(&p->tile_info, sizeof(uint2) * ((size_t)p->n_tiles + 2));
    if (e == hipSuccess) e = hipMalloc(&p->consts, sizeof(unsigned long long) * EVP_NCONST);
    if (e == hipSuccess) e = hipMemsetAsync(p->tile_info, 0, sizeof(uint2) * ((size_t)p->n_tiles + 2), static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipMemsetAsync(p->consts, 0, sizeof(unsigned long long) * EVP_NCONST, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
      (void)hipFree(p->items);
      (void)hipFree(p->tile_info);
      (void)hipFree(p->consts);
      delete p;
      g_last_hip_error = (int)e;
      return e == hipErrorOutOfMemory ? BEAR_ERR_NOMEM : BEAR_ERR_HIP;
    }
    const uint64_t cap = (uint64_t)ws->num_cu * 16;
    const int grid = (int)(p->n_tiles < cap ? p->n_tiles : cap);
    hipLaunchKernelGGL(evp_build_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), test, train, n_rows, p->n_tiles,
                       p->items, p->tile_info, p->consts);
    e = hipGetLastError();
    if (e != hipSuccess) {
      (void)hipFree(p->items);
      (void)hipFree(p->tile_info);
      (void)hipFree(p->consts);
      delete p;
      g_last_hip_error = (int)e;
      return BEAR_ERR_HIP;
    }
    p->bytes = ibytes + sizeof(uint2) * ((size_t)p->n_tiles + 2) + sizeof(unsigned long long) * EVP_NCONST;
  }
  *out = p;
  return BEAR_OK;
}

int bear_eval_plan_destroy(bear_eval_plan *plan) {
  if (!plan) return BEAR_OK;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(plan->device);
  (void)hipFree(plan->items);
  (void)hipFree(plan->tile_info);
  (void)hipFree(plan->consts);
  (void)hipSetDevice(prev);
  delete plan;
  return BEAR_OK;
}

uint64_t bear_eval_plan_bytes(const bear_eval_plan *plan) { return plan ? plan->bytes : 0; }

int bear_eval_plan_f64(bear_ws *ws, const bear_eval_plan *plan, const uint32_t *test, const uint32_t *train, const double *prior,
                       uint64_t n_rows, const double *h, int n_h, int with_ar, const double *van_reg, int n_van, double eps,
                       uint64_t noise_seed, uint64_t row_base, const uint32_t *row_ids, double *out, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!plan || plan->test != test || plan->train != train || plan->n_rows != n_rows || plan->device != ws->device)
    return BEAR_ERR_INVALID_ARG;
  if (misaligned(row_ids)) return BEAR_ERR_INVALID_ARG;
  evl_args A;
  st = eval_make_args(test, train, prior, n_rows, h, n_h, with_ar, van_reg, n_van, eps, noise_seed, row_base, out, &A);
  if (st != BEAR_OK) return st;
  A.has_rid = row_ids ? 1 : 0;
  // The plan decides the vanilla models' arg-max on the INTEGER training counts (a letter a whole count below the top cannot win):
  // that is the arg-max of count + van_reg + eps + noise only while 17.5 sigma = 1750 eps stays below a count and the sum keeps
  // the counts apart (bear_eval_f64 takes any values).
  if (A.n_van && !(1750.0 * eps < 0.5)) return BEAR_ERR_INVALID_ARG;
  for (int k = 0; k < A.n_van; ++k)
    if (!(van_reg[k] >= 0.0 && van_reg[k] <= 0x1p30)) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n_models = A.n_h + A.n_van;
  const uint64_t nt = plan->n_tiles;
  const int grid = (int)(nt < (uint64_t)ws->num_cu ? (nt ? nt : 1) : (uint64_t)ws->num_cu);   // one resident 768-thread block per CU
  const double2 *lt = reinterpret_cast<const double2 *>(ws->logtab);
  // launches: BEAR models (product path) four at a time, vanilla models (lgamma tables) four at a time -- a BEAR group and a
  // vanilla group share a launch; the first launch also carries the AR model and the total length.
  int h0 = 0, v0 = 0;
  bool first = true;
  while (first || h0 < A.n_h || v0 < A.n_van) {
    const int nh = A.n_h - h0 < EVP_MAXH ? A.n_h - h0 : EVP_MAXH;
    int nv = A.n_van - v0 < EVP_MAXV ? A.n_van - v0 : EVP_MAXV;
    if (nh > 1) nv = 0;   // four BEAR models fill the register file (168 per lane at three waves per SIMD): the vanilla group follows
    int common = first ? 1 : 0;
    evs_slots S;
    for (int k = 0; k < EVS_NOUT; ++k) S.slot[k] = -1;
    for (int k = 0; k < nh; ++k) {            // output: ll_ear[n_h], ll_arm, ll_van[n_van], cor_ear[n_h], cor_arm, cor_van[n_van], total
      S.slot[k] = h0 + k;
      S.slot[EVS_CHUNK + k] = n_models + 1 + h0 + k;
    }
    for (int k = 0; k < nv; ++k) {
      S.slot[EVP_SLOT_VAN + k] = A.n_h + 1 + v0 + k;
      S.slot[EVS_CHUNK + EVP_SLOT_VAN + k] = n_models + 1 + A.n_h + 1 + v0 + k;
    }
    if (common) {
      S.slot[2 * EVS_CHUNK] = A.n_h;
      S.slot[2 * EVS_CHUNK + 1] = n_models + 1 + A.n_h;
      S.slot[2 * EVS_CHUNK + 2] = 2 * n_models + 2;
    }
#define EVP_LAUNCH(NH_, NV_)                                                                                                        \
  hipLaunchKernelGGL((eval_plan_kernel<NH_, NV_>), dim3(grid), dim3(EVP_THREADS), sizeof(evp_lds), s, test, train, prior, row_ids, n_rows, A, \
                     h0, nh, v0, nv, common, plan->items, plan->tile_info, plan->consts, nt, lt, ws->eval_partials EVP_DBG_ARG)
    if (nh == 0) EVP_LAUNCH(0, 4);
    else if (nh == 1 && nv == 0) EVP_LAUNCH(1, 0);
    else if (nh == 1) EVP_LAUNCH(1, 4);
    else EVP_LAUNCH(4, 0);
#undef EVP_LAUNCH
    hipLaunchKernelGGL(eval_sorted_finalize_kernel, dim3((EVS_NOUT + 3) / 4), dim3(256), 0, s, ws->eval_partials, grid, S, out);
    h0 += nh;
    v0 += nv;
    first = false;
  }
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_bmm_f64(bear_ws *ws, const uint32_t *counts, uint64_t n_rows, const double *alpha, int n_alpha, double *out,
                 void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!out || !alpha || n_alpha <= 0 || n_alpha > EVL_MAX_MODELS || (n_rows && !counts) || misaligned(counts))
    return BEAR_ERR_INVALID_ARG;
  evl_args A;
  memset(&A, 0, sizeof(A));
  A.n_van = n_alpha;
  for (int k = 0; k < n_alpha; ++k) {
    if (!(alpha[k] > 0.0)) return BEAR_ERR_INVALID_ARG;
    A.inv_h[k] = alpha[k];
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  st = launch_eval(ws, counts, nullptr, nullptr, n_rows, A, ws->eval_out, s);
  if (st != BEAR_OK) return st;
  // result vector layout: [ll_arm (unused), ll_van[n_alpha], ...]
  HIP_TRY(hipMemcpyAsync(out, ws->eval_out + 1, sizeof(double) * (size_t)n_alpha, hipMemcpyDeviceToDevice, s));
  return BEAR_OK;
}

// Developer probe (not part of include/bear_hip.h): resident blocks per CU the runtime reports for
// the two sorted kernels at their dynamic-LDS sizes.
// Developer probe: copies the phase-timing buffer written by BEAR_DEBUG_STOP=9 (n u64 words) to the host.
int bear_debug_read_timing(bear_ws *ws, unsigned long long *host, int n_words) {
  if (!ws || !host) return BEAR_ERR_INVALID_ARG;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host, ws->dbg, sizeof(unsigned long long) * (size_t)n_words, hipMemcpyDeviceToHost));
  return BEAR_OK;
}

// Developer probe: the paired lists (bear_plan_pair_contexts) of tiles [first, first + n) and their first rows, to the host
// (scripts/dev/pair_conflicts.py counts the bank-pair collisions of the triple adds from them).  A row is LIN_LIVE2_STRIDE uint16:
// [0] = entries m, [1] = 0, m entries, then lin_lev_len(m) level words; a call with lists == NULL returns that stride instead.
extern "C" int bear_debug_pair_lists(const bear_plan *plan, uint64_t first, uint64_t n, uint16_t *lists, uint64_t *row0) {
  if (!lists) return (int)LIN_LIVE2_STRIDE;
  if (!plan || !plan->live2 || !lists || !row0 || first + n > plan->n_tiles) return BEAR_ERR_INVALID_ARG;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(lists, plan->live2 + first * LIN_LIVE2_STRIDE, n * LIN_LIVE2_STRIDE * sizeof(uint16_t), hipMemcpyDeviceToHost));
  std::vector<pln_tile> t(n);
  HIP_TRY(hipMemcpy(t.data(), plan->tiles + first, n * sizeof(pln_tile), hipMemcpyDeviceToHost));
  for (uint64_t k = 0; k < n; ++k) row0[k] = t[k].row0;
  return BEAR_OK;
}

int bear_debug_occupancy(int which) {
  int nb = -1;
  hipError_t e = which == 0
      ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dm_prior_sorted_kernel<0>, SRT_THREADS, sizeof(srt_lds_n))
      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dm_ref_sorted_kernel, SRT_THREADS, sizeof(srt_lds_r));
  return e == hipSuccess ? nb : -1000 - (int)e;
}

int bear_log_gamma_f64(const double *conc, uint64_t n, uint64_t n_samples, uint64_t seed, double *out, void *stream) {
  if (n == 0 || n_samples == 0) return BEAR_OK;
  if (!conc || !out) return BEAR_ERR_INVALID_ARG;
  if (n_samples > (~0ull) / n) return BEAR_ERR_INVALID_ARG;
  const uint64_t total = n * n_samples;
  uint64_t blocks = (total + SMP_THREADS - 1) / SMP_THREADS;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(log_gamma_kernel, dim3((unsigned)blocks), dim3(SMP_THREADS), 0, static_cast<hipStream_t>(stream), conc, n,
                     n_samples, seed, out);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_logdir_sample_f64(const uint32_t *counts, const double *prior, uint64_t n_rows, const double *h, int n_h,
                           int with_ar, const double *van, int n_van, int mc_samples, int map, uint64_t seed,
                           uint64_t row_base, double *out, void *stream) {
  if (n_h < 0 || n_van < 0 || n_h + n_van > SMP_MAX_MODELS || (n_h && !h) || (n_van && !van)) return BEAR_ERR_INVALID_ARG;
  if (with_ar && !map) return BEAR_ERR_INVALID_ARG;   // the AR model enters only the MAP table (get_var_probs.py:150-153)
  if (map) mc_samples = 1;                            // get_var_probs.py:131-132
  if (mc_samples <= 0) return BEAR_ERR_INVALID_ARG;
  const int M = (with_ar ? 1 : 0) + n_h + n_van;
  if (n_rows == 0 || M == 0) return BEAR_OK;
  if (!out || ((n_h || with_ar) && !prior)) return BEAR_ERR_INVALID_ARG;
  smp_args A;
  memset(&A, 0, sizeof(A));
  A.n_h = n_h;
  A.n_van = n_van;
  A.arm = with_ar ? 1 : 0;
  A.has_counts = counts ? 1 : 0;
  A.has_prior = prior ? 1 : 0;
  A.map = map ? 1 : 0;
  A.mc = (uint32_t)mc_samples;
  A.seed = seed;
  A.row_base = row_base;
  for (int j = 0; j < n_h; ++j) {
    if (!(h[j] > 0.0)) return BEAR_ERR_INVALID_ARG;
    A.w[j] = 1.0 / h[j];
  }
  for (int k = 0; k < n_van; ++k) A.w[n_h + k] = van[k];
  const uint64_t per_row = (uint64_t)M * (uint64_t)mc_samples;
  if (per_row > 0xffffffffull || n_rows > (~0ull) / (5 * per_row)) return BEAR_ERR_INVALID_ARG;
  uint64_t blocks = (n_rows * per_row + SMP_THREADS - 1) / SMP_THREADS;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(logdir_sample_kernel, dim3((unsigned)blocks), dim3(SMP_THREADS), 0, static_cast<hipStream_t>(stream), counts,
                     prior, n_rows, A, out);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_shuffle_rows(const void *src, void *dst, uint64_t n_rows, uint32_t row_bytes, uint64_t seed, void *stream) {
  if (n_rows == 0 || row_bytes == 0) return BEAR_OK;
  if (!src || !dst || src == dst) return BEAR_ERR_INVALID_ARG;
  if (n_rows > (1ull << 62) / row_bytes) return BEAR_ERR_INVALID_ARG;
  const uint32_t hb = shf_half_bits(n_rows);
  const bool words = (row_bytes % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 4 == 0);
  const uint64_t total = words ? n_rows * (row_bytes / 4) : n_rows * (uint64_t)row_bytes;
  uint64_t blocks = (total + 255) / 256;
  if (blocks > 1u << 20) blocks = 1u << 20;
  if (words)
    hipLaunchKernelGGL(shuffle_words_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint32_t *>(src), static_cast<uint32_t *>(dst), n_rows, row_bytes / 4, hb, seed);
  else
    hipLaunchKernelGGL(shuffle_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), n_rows, row_bytes, hb, seed);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

uint64_t bear_shuffle_source_row(uint64_t i, uint64_t n_rows, uint64_t seed) {
  return (n_rows == 0 || i >= n_rows) ? i : shf_perm(i, n_rows, shf_half_bits(n_rows), seed);
}

static int cnn_check(const bear_ws *ws, int lag, int fw, int nf, int l1) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (lag < 1 || lag > CNN_MAX_LAG || fw < 1 || fw > lag || nf != CNN_NF || l1 != CNN_L1) return BEAR_ERR_INVALID_ARG;
  return BEAR_OK;
}

int bear_cnn_param_count(int lag, int filter_width, int num_filters, int layer1_width) {
  if (lag < 1 || lag > CNN_MAX_LAG || filter_width < 1 || filter_width > lag || num_filters != CNN_NF || layer1_width != CNN_L1)
    return BEAR_ERR_INVALID_ARG;
  return cnn_make_dims(lag, filter_width).total;
}

int bear_cnn_forward_f64(bear_ws *ws, const uint64_t *kmer_code, uint64_t n_rows, int lag, int filter_width, int num_filters,
                         int layer1_width, const double *params, double *prior, double *t1_save, void *stream) {
  int st = cnn_check(ws, lag, filter_width, num_filters, layer1_width);
  if (st != BEAR_OK) return st;
  if (n_rows == 0) return BEAR_OK;
  if (!kmer_code || !params || !prior || misaligned(t1_save) || (reinterpret_cast<uintptr_t>(prior) & 7u)) return BEAR_ERR_INVALID_ARG;
  const cnn_dims D = cnn_make_dims(lag, filter_width);
  const size_t lds = sizeof(double) * (BEAR_EXPTAB_N + (size_t)filter_width * 6 * CNN_NF + (CNN_THREADS / 64) * CNN_FWD_SCRATCH);
  uint64_t blocks = (n_rows + CNN_THREADS - 1) / CNN_THREADS;
  if (blocks > (uint64_t)ws->num_cu * 16) blocks = (uint64_t)ws->num_cu * 16;
  hipLaunchKernelGGL(cnn_forward_kernel, dim3((unsigned)blocks), dim3(CNN_THREADS), lds, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const unsigned long long *>(kmer_code), n_rows, D, params, prior, t1_save,
                     static_cast<const pln_tile *>(nullptr), static_cast<const uint16_t *>(nullptr), (n_rows + 63) / 64, cnn_all_positions(D));
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

// sizes the block-partial buffer of the CNN backward pass; returns the grid.  With may_alloc == 0 (inside a stream capture) a
// buffer that is too small is an error: call bear_cnn_reserve first.
static int cnn_backward_grid(bear_ws *ws, const cnn_dims &D, uint64_t n_rows, int filter_width, int *waves_out, int *parts_out,
                             size_t *lds_out, uint64_t *blocks_out, hipStream_t s, int may_alloc) {
  const size_t fixed = sizeof(double) * (BEAR_EXPTAB_N + (size_t)filter_width * 6 * CNN_NF + (size_t)((D.total + 1) & ~1));
  // cnn_backward_parts_kernel<2>: two lanes per context, eight waves of 32-context tiles (two per SIMD), when the staging fits
  // next to the filter, parameter and gradient images (every reference config); otherwise 64-context tiles, one wave per SIMD.
  // `waves` names the form (8 / <= 4).  BEAR_CNN_BACKWARD=1 forces the second form (developer A/B runs, tests).
  const size_t lds2 = sizeof(double) * (cnnq_fixed_doubles(D) + (size_t)cnnq<2>::WAVES * cnnq<2>::WAVE_DOUBLES);
  const char *force = getenv("BEAR_CNN_BACKWARD");
  // BEAR_AMD_DETERMINISTIC: ONE wave per block.  The block's gradient image takes LDS floating-point atomics from all its waves,
  // in whatever order they get there; with one wave the adds happen in program order, the blocks' images are summed in a fixed
  // order anyway (cnn_finalize_kernel) -- two runs of a step are bit-identical, at an eighth of the waves per CU.
  const bool det = bear_deterministic();
  int waves = det ? 1 : 4, parts = 0;
  while (waves > 1 && fixed + (size_t)waves * CNN_WAVE_DOUBLES * sizeof(double) > 160u * 1024u) waves >>= 1;
  size_t lds = fixed + (size_t)waves * CNN_WAVE_DOUBLES * sizeof(double);
  uint64_t per_block = (uint64_t)64 * waves;
  if (lds2 <= 160u * 1024u && !(force && force[0] == '1')) {
    parts = 1;
    waves = det ? 1 : cnnq<2>::WAVES;
    lds = sizeof(double) * (cnnq_fixed_doubles(D) + (size_t)waves * cnnq<2>::WAVE_DOUBLES);
    per_block = (uint64_t)cnnq<2>::TILE * waves;
  }
  if (lds > 160u * 1024u) return BEAR_ERR_INVALID_ARG;
  uint64_t blocks = (n_rows + per_block - 1) / per_block;
  if (blocks > (uint64_t)ws->num_cu) blocks = (uint64_t)ws->num_cu;
  if (blocks == 0) blocks = 1;
  const size_t need = (size_t)blocks * D.total;
  if (ws->cnn_partials_cap < need) {
    if (!may_alloc) return BEAR_ERR_INVALID_ARG;
    HIP_TRY(hipStreamSynchronize(s));
    if (ws->cnn_partials) (void)hipFree(ws->cnn_partials);
    ws->cnn_partials = nullptr;
    ws->cnn_partials_cap = 0;
    HIP_TRY(hipMalloc(&ws->cnn_partials, sizeof(double) * need));
    ws->cnn_partials_cap = need;
  }
  *waves_out = waves;
  *parts_out = parts;
  *lds_out = lds;
  *blocks_out = blocks;
  return BEAR_OK;
}

// live_plan [nullable]: a five-column plan of the table -- the part kernel then walks its lists of contexts that hold counts
static int launch_cnn_backward(bear_ws *ws, const cnn_dims &D, const uint64_t *kmer_code, uint64_t n_rows, int filter_width,
                               const double *params, const double *t1_save, const double *prior, const double *grad_prior,
                               double *grad_params, hipStream_t s, int may_alloc, const bear_plan *live_plan = nullptr) {
  int waves = 0, parts = 0;
  size_t lds = 0;
  uint64_t blocks = 0;
  int st = cnn_backward_grid(ws, D, n_rows, filter_width, &waves, &parts, &lds, &blocks, s, may_alloc);
  if (st != BEAR_OK) return st;
  const bool parts2 = parts != 0;
  const void *fn = parts2 ? reinterpret_cast<const void *>(cnn_backward_parts_kernel<2>) : reinterpret_cast<const void *>(cnn_backward_kernel);
  if (may_alloc) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned long long *kc = reinterpret_cast<const unsigned long long *>(kmer_code);
  if (parts2) {
    const bool lists = live_plan && live_plan->live && live_plan->n_live_rows < n_rows;   // all rows live: plain groups of rows
    const uint64_t groups = lists ? live_plan->n_tiles : (n_rows + cnnq<2>::TILE - 1) / cnnq<2>::TILE;
    hipLaunchKernelGGL(cnn_backward_parts_kernel<2>, dim3((unsigned)blocks), dim3(64 * waves), lds, s, kc, n_rows, D, params, t1_save, prior,
                       grad_prior, ws->cnn_partials, lists ? live_plan->tiles : nullptr, lists ? live_plan->live : nullptr, groups,
                       cnn_all_positions(D));
  }
  else
    hipLaunchKernelGGL(cnn_backward_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds, s, kc, n_rows, D, params, t1_save, prior,
                       grad_prior, ws->cnn_partials);
  hipLaunchKernelGGL(cnn_finalize_kernel, dim3((D.total + 3) / 4), dim3(256), 0, s, ws->cnn_partials, (int)blocks, D.total, grad_params);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_cnn_backward_f64(bear_ws *ws, const uint64_t *kmer_code, uint64_t n_rows, int lag, int filter_width, int num_filters,
                          int layer1_width, const double *params, const double *t1_save, const double *prior,
                          const double *grad_prior, double *grad_params, void *stream) {
  int st = cnn_check(ws, lag, filter_width, num_filters, layer1_width);
  if (st != BEAR_OK) return st;
  if (!params || !grad_params) return BEAR_ERR_INVALID_ARG;
  if (n_rows && (!kmer_code || !t1_save || !prior || !grad_prior || misaligned(t1_save))) return BEAR_ERR_INVALID_ARG;
  const cnn_dims D = cnn_make_dims(lag, filter_width);
  return launch_cnn_backward(ws, D, kmer_code, n_rows, filter_width, params, t1_save, prior, grad_prior, grad_params,
                             static_cast<hipStream_t>(stream), 1);
}

int bear_cnn_reserve(bear_ws *ws, uint64_t n_rows, int lag, int filter_width, int num_filters, int layer1_width) {
  int st = cnn_check(ws, lag, filter_width, num_filters, layer1_width);
  if (st != BEAR_OK) return st;
  const cnn_dims D = cnn_make_dims(lag, filter_width);
  int waves = 0, parts = 0;
  size_t lds = 0;
  uint64_t blocks = 0;
  st = cnn_backward_grid(ws, D, n_rows, filter_width, &waves, &parts, &lds, &blocks, nullptr, 1);
  if (st != BEAR_OK) return st;
  HIP_TRY(hipFuncSetAttribute(parts ? reinterpret_cast<const void *>(cnn_backward_parts_kernel<2>)
                                                      : reinterpret_cast<const void *>(cnn_backward_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return BEAR_OK;
}

static int cnn_forward_levels(bear_ws *ws, const bear_plan *plan, const cnn_dims &D, const uint64_t *kmer_code, uint64_t n_rows,
                              const double *params, double *prior, double *t1_buf, hipStream_t s);
// Level k's rows are prefixes of L_k letters (L_0 = lag: the contexts): it evaluates the positions whose window [p, p + fw) lies
// inside its prefix but not inside the next level's shorter one -- p + fw in (L_{k+1}, L_k]; the last level takes what is left.
static cnn_level_io cnn_level_positions(const bear_plan *plan, const cnn_dims &D, int k) {
  const int K = plan->n_cnn_levels;
  const int L = k == 0 ? D.lag : plan->cnn_levels[k - 1].letters;
  cnn_level_io io = cnn_all_positions(D);
  io.p_hi = L - D.fw + 1;
  io.p_lo = k == K ? 0 : plan->cnn_levels[k].letters - D.fw + 1;
  io.head = k == 0;
  io.p_hi -= plan->n_cnn_win[k];          // the level's last positions come from its window tables
  return io;
}
// the launch over the rows of a window table: its one position, no head, no parent
static cnn_level_io cnn_window_positions(const cnn_dims &D, const bear_window_dev &wt) {
  cnn_level_io io = cnn_all_positions(D);
  io.p_lo = wt.pos;
  io.p_hi = io.p_lo + 1;
  io.head = 0;
  return io;
}
// what level k's rows put their layer-1 sums together from besides their own positions: the parent level's rows and the window tables
static void cnn_level_sources(const bear_plan *plan, int k, cnn_level_io &io) {
  if (k < plan->n_cnn_levels) {
    io.t1_parent = plan->cnn_levels[k].rows;                 // level k + 1
    io.parent = plan->cnn_levels[k].parent_of_below;
  }
  io.n_win = plan->n_cnn_win[k];
  for (int q = 0; q < plan->n_cnn_win[k]; ++q) {
    io.win_rows[q] = plan->cnn_win[k][q].rows;
    io.win_row_of[q] = plan->cnn_win[k][q].row_of_context;
  }
}
static bool cnn_parts_form_forced_off() {      // BEAR_CNN_BACKWARD=1 (cnn_backward_grid): the 64-context form of the backward kernel, which has no position range
  const char *force = getenv("BEAR_CNN_BACKWARD");
  return force && force[0] == '1';
}
// ---- the convolutional step over prefix levels (kernels_cnn.h, cnn_level_io): the forward kernel once per level from the shortest
// prefixes down to the contexts, the planned DM kernel with gradient rows, the backward kernel once per level the other way with a
// row-sum launch in between; block partials accumulate in the workspace's buffer (stream order), one finalize at the end.
// Level k < K evaluates position P - 1 - k, the last level K the positions [0, P - K).
static int cnn_train_reduce_levels(bear_ws *ws, const bear_plan *plan, const cnn_dims &D, const uint64_t *kmer_code, uint64_t n_rows,
                                   const double *theta, double *prior_buf, double *t1_buf, double *grad_rows_buf, double eps, int train_ar,
                                   double *packed, hipStream_t s) {
  const int K = plan->n_cnn_levels;
  const double *params = theta + 1;
  int bw_waves = 0, bw_parts = 0;
  size_t bw_lds = 0;
  uint64_t bw_blocks = 0;
  int st = cnn_backward_grid(ws, D, n_rows, D.fw, &bw_waves, &bw_parts, &bw_lds, &bw_blocks, s, 0);
  if (st != BEAR_OK) return st;
  if (!bw_parts) return BEAR_ERR_INVALID_ARG;      // (attach refuses shapes whose staging does not fit: not reached)
  auto level_codes = [&](int k) { return k == 0 ? reinterpret_cast<const unsigned long long *>(kmer_code) : plan->cnn_levels[k - 1].codes; };
  auto level_rows = [&](int k) { return k == 0 ? n_rows : plan->cnn_levels[k - 1].n; };
  auto level_table = [&](int k) { return k == 0 ? t1_buf : plan->cnn_levels[k - 1].rows; };
  auto level_io = [&](int k) { return cnn_level_positions(plan, D, k); };
  // Level 0 evaluates no position itself (all of them come from its parent level and the window tables): the forward pass then keeps
  // no layer-1 sums of the contexts -- the backward pass puts them together again from the same rows (cnn_backward_parts_kernel)
  const cnn_level_io io0 = level_io(0);
  const bool recompute_t1 = io0.p_lo >= io0.p_hi && !getenv("BEAR_AMD_CNN_KEEP_T1");
  st = cnn_forward_levels(ws, plan, D, kmer_code, n_rows, params, prior_buf, recompute_t1 ? nullptr : t1_buf, s);
  if (st != BEAR_OK) return st;
  bear_params only_eps;
  memset(&only_eps, 0, sizeof(only_eps));
  only_eps.eps = eps;
  st = launch_prior_plan_grad(ws, plan, prior_buf, only_eps, theta, train_ar, 1, packed, grad_rows_buf, s);   // softmax rows: normalised
  if (st != BEAR_OK) return st;
  const uint64_t per_block = (uint64_t)cnnq<2>::TILE * (uint64_t)bw_waves;
  bool first_launch = true;      // the first backward launch writes every row of the partial buffer the finalize reads, the others add
  auto backward = [&](const unsigned long long *codes, uint64_t n, const double *t1_in, cnn_level_io io) {
    uint64_t blocks = (n + per_block - 1) / per_block;
    if (blocks > bw_blocks) blocks = bw_blocks;
    if (blocks == 0) blocks = 1;
    if (first_launch) blocks = bw_blocks;
    io.accumulate = first_launch ? 0 : 1;
    first_launch = false;
    hipLaunchKernelGGL(cnn_backward_parts_kernel<2>, dim3((unsigned)blocks), dim3(64 * bw_waves), bw_lds, s, codes, n, D, params, t1_in, prior_buf,
                       grad_rows_buf, ws->cnn_partials, static_cast<const pln_tile *>(nullptr), static_cast<const uint16_t *>(nullptr),
                       (n + cnnq<2>::TILE - 1) / cnnq<2>::TILE, io);
  };
  for (int k = 0; k <= K; ++k) {
    cnn_level_io io = level_io(k);
    const uint64_t n = level_rows(k);
    const int W = plan->n_cnn_win[k];
    if (k > 0) {        // this level's dT1 rows = the sums of its children's
      uint64_t sb = (n * (CNN_L1 / 2) + 255) / 256;
      if (sb > (uint64_t)ws->num_cu * 32) sb = (uint64_t)ws->num_cu * 32;
      hipLaunchKernelGGL(cnn_level_sum_kernel, dim3((unsigned)sb), dim3(256), 0, s, level_table(k - 1), plan->cnn_levels[k - 1].child_start,
                         n, level_table(k));
    }
    io.dT1 = (k == 0 && (K > 0 || W > 0)) ? t1_buf : (k > 0 ? level_table(k) : nullptr);   // level 0 leaves its dT1 rows where its t1 rows were
    if (k == 0 && recompute_t1) cnn_level_sources(plan, 0, io);
    // (a level of prefixes whose positions all come from window tables has nothing to do itself: its dT1 rows feed the tables below)
    if (k == 0 && io.p_lo >= io.p_hi && !getenv("BEAR_AMD_CNN_NO_HEAD_KERNEL")) {
      // the contexts evaluate no position themselves: the head-only kernel (kernels_cnn.h); always the step's first backward launch
      io.accumulate = first_launch ? 0 : 1;
      first_launch = false;
      hipLaunchKernelGGL(cnn_backward_head_kernel, dim3((unsigned)bw_blocks), dim3(CNH_WAVES * 64), cnh_lds_bytes(), s, n, D, params,
                         recompute_t1 ? static_cast<const double *>(nullptr) : t1_buf, prior_buf, grad_rows_buf, ws->cnn_partials, io);
    } else if (k == 0 || io.p_lo < io.p_hi) {
      backward(level_codes(k), n, (k == 0 && recompute_t1) ? static_cast<const double *>(nullptr) : t1_buf, io);
    }
    for (int q = 0; q < W; ++q) {      // the level's window tables: a window's dT1 row = the sum of its rows' (anywhere in the level)
      const bear_window_dev &wt = plan->cnn_win[k][q];
      uint64_t sb = (wt.n + 3) / 4;
      if (sb > (uint64_t)ws->num_cu * 8) sb = (uint64_t)ws->num_cu * 8;
      hipLaunchKernelGGL(cnn_window_sum_kernel, dim3((unsigned)sb), dim3(256), 0, s, level_table(k), wt.perm, wt.child_start, wt.n, wt.rows);
      cnn_level_io wio = cnn_window_positions(D, wt);
      wio.dT1 = wt.rows;
      backward(wt.codes, wt.n, t1_buf, wio);
    }
  }
  hipLaunchKernelGGL(cnn_finalize_kernel, dim3((D.total + 3) / 4), dim3(256), 0, s, ws->cnn_partials, (int)bw_blocks, D.total, packed + 2);
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

// Prefix levels of the plan's (k-mer-sorted) contexts for the convolutional step: see include/bear_hip.h.
static void plan_drop_cnn_levels(bear_plan *plan) {
  for (int k = 0; k < plan->n_cnn_levels; ++k) {
    plan->bytes -= plan->cnn_levels[k].bytes;
    bear_level_free(&plan->cnn_levels[k]);
  }
  for (int k = 0; k <= CNN_MAX_LAG; ++k) {
    for (int q = 0; q < plan->n_cnn_win[k]; ++q) {
      plan->bytes -= plan->cnn_win[k][q].bytes;
      bear_window_free(&plan->cnn_win[k][q]);
    }
    plan->n_cnn_win[k] = 0;
  }
  plan->n_cnn_levels = 0;
  plan->n_cnn_windows = 0;
  plan->cnn_codes = nullptr;
}

int bear_plan_attach_cnn_levels(bear_plan *plan, const uint64_t *kmer_code, int lag, int filter_width, int *n_levels, void *stream) {
  if (n_levels) *n_levels = 0;
  if (!plan || plan->ncol != 5 || lag < 1 || lag > CNN_MAX_LAG || filter_width < 1 || filter_width > lag) return BEAR_ERR_INVALID_ARG;
  if (plan->n_rows && (!kmer_code || misaligned(kmer_code))) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipStreamSynchronize(s));
  plan_drop_cnn_levels(plan);                          // a plan holds one set of levels: the new one replaces it
  const cnn_dims D = cnn_make_dims(lag, filter_width);
  // (the part form of the backward kernel is the one with a position range: shapes whose staging does not fit keep the plain step)
  if (sizeof(double) * (cnnq_fixed_doubles(D) + (size_t)cnnq<2>::WAVES * cnnq<2>::WAVE_DOUBLES) > 160u * 1024u) return BEAR_OK;
  if (plan->n_rows < 2 || plan->n_live_rows != plan->n_rows) return BEAR_OK;   // (the step walks the plan's lists instead: no levels)
  const unsigned long long *below = reinterpret_cast<const unsigned long long *>(kmer_code);
  uint64_t n_below = plan->n_rows;
  int misses = 0;
  double keep_ratio = 0.6;
  if (const char *r = getenv("BEAR_AMD_CNN_LEVEL_RATIO")) keep_ratio = atof(r);     // developer switch (scripts/dev/cnn_levels_time.py)
  for (int k = 1; k <= D.P - 1 && misses < 3; ++k) {
    bear_level_dev lv;
    const int st = bear_level_build(below, n_below, lag - k, &lv, s);
    if (st != BEAR_OK) {
      if (st == BEAR_ERR_HIP) g_last_hip_error = bear_count_last_hip_error();
      plan_drop_cnn_levels(plan);                      // the levels built so far go with it: the plan is as it was without levels
      return st;
    }
    // a level pays when it is clearly smaller than the last one kept (a position per row either way, plus the row traffic); a
    // prefix length that does not (a sparser table: its prefixes of lag - 1 letters hardly repeat) is skipped -- the level below
    // then evaluates that position too -- and the next shorter one is tried against the same rows
    if ((double)lv.n > keep_ratio * (double)n_below) {
      bear_level_free(&lv);
      ++misses;
      continue;
    }
    misses = 0;
    lv.bytes = lv.n * (8 + 4 + 16 * 8) + 4 * n_below;
    plan->cnn_levels[plan->n_cnn_levels++] = lv;
    plan->bytes += lv.bytes;
    below = lv.codes;
    n_below = lv.n;
  }
  // Window tables, level by level (level 0 = the contexts): for the level's own positions, from the last one up, CNN_MAX_WIN at most,
  // each only while the level holds at least eight rows per distinct window (a table costs one position per window plus a 128-byte
  // gather per row and direction; a position evaluated per row costs ~25 times that gather).
  if (!getenv("BEAR_AMD_CNN_NO_WINDOWS")) {
    const int K = plan->n_cnn_levels;
    for (int k = 0; k <= K; ++k) {
      const unsigned long long *codes_k = k == 0 ? reinterpret_cast<const unsigned long long *>(kmer_code) : plan->cnn_levels[k - 1].codes;
      const uint64_t n_k = k == 0 ? plan->n_rows : plan->cnn_levels[k - 1].n;
      const int L = k == 0 ? lag : plan->cnn_levels[k - 1].letters;
      const int p_hi = L - filter_width + 1, p_lo = k == K ? 0 : plan->cnn_levels[k].letters - filter_width + 1;
      bear_window_dev built[CNN_MAX_WIN];
      int nb = 0;
      for (int p = p_hi - 1; p >= p_lo && nb < CNN_MAX_WIN && n_k >= 64; --p) {
        bear_window_dev wt;
        const int st = bear_window_build(codes_k, n_k, p, filter_width, &wt, s);
        if (st != BEAR_OK) {
          if (st == BEAR_ERR_HIP) g_last_hip_error = bear_count_last_hip_error();
          for (int q = 0; q < nb; ++q) bear_window_free(&built[q]);
          plan_drop_cnn_levels(plan);
          return st;
        }
        if (wt.n * 8 > n_k) {
          bear_window_free(&wt);
          break;
        }
        built[nb++] = wt;
      }
      for (int q = 0; q < nb; ++q) {         // ascending positions
        plan->cnn_win[k][q] = built[nb - 1 - q];
        plan->bytes += plan->cnn_win[k][q].bytes;
      }
      plan->n_cnn_win[k] = nb;
      plan->n_cnn_windows += nb;
    }
  }
  if (plan->n_cnn_levels || plan->n_cnn_windows) {
    plan->cnn_codes = kmer_code;
    plan->cnn_lag = lag;
    plan->cnn_fw = filter_width;
  }
  if (n_levels) *n_levels = plan->n_cnn_levels;
  return BEAR_OK;
}

int bear_plan_cnn_window_rows(const bear_plan *plan, uint64_t *rows_out, int *pos_out, int *level_out, int capacity) {
  if (!plan || (capacity > 0 && !rows_out)) return BEAR_ERR_INVALID_ARG;
  int t = 0;
  for (int k = 0; k <= plan->n_cnn_levels; ++k)
    for (int q = 0; q < plan->n_cnn_win[k]; ++q, ++t)
      if (t < capacity) {
        rows_out[t] = plan->cnn_win[k][q].n;
        if (pos_out) pos_out[t] = plan->cnn_win[k][q].pos;
        if (level_out) level_out[t] = k;
      }
  return plan->n_cnn_windows;
}

int bear_plan_cnn_level_rows(const bear_plan *plan, uint64_t *rows_out, int *letters_out, int capacity) {
  if (!plan || (capacity > 0 && !rows_out)) return BEAR_ERR_INVALID_ARG;
  for (int k = 0; k < plan->n_cnn_levels && k < capacity; ++k) {
    rows_out[k] = plan->cnn_levels[k].n;
    if (letters_out) letters_out[k] = plan->cnn_levels[k].letters;
  }
  return plan->n_cnn_levels;
}

// The forward pass alone over a plan's prefix levels (evaluation-style callers, bench.py): prior rows and the contexts' layer-1 sums.
static int cnn_forward_levels(bear_ws *ws, const bear_plan *plan, const cnn_dims &D, const uint64_t *kmer_code, uint64_t n_rows,
                              const double *params, double *prior, double *t1_buf, hipStream_t s) {
  const int K = plan->n_cnn_levels;
  const size_t fwd_lds = sizeof(double) * (BEAR_EXPTAB_N + (size_t)D.fw * 6 * CNN_NF + (CNN_THREADS / 64) * CNN_FWD_SCRATCH);
  for (int k = K; k >= 0; --k) {
    for (int q = 0; q < plan->n_cnn_win[k]; ++q) {          // the level's window tables first: one position over its distinct windows
      const bear_window_dev &wt = plan->cnn_win[k][q];
      const uint64_t groups = (wt.n + 63) / 64;
      uint64_t blocks = (groups + CNN_THREADS / 64 - 1) / (CNN_THREADS / 64);
      if (blocks > (uint64_t)ws->num_cu * 16) blocks = (uint64_t)ws->num_cu * 16;
      hipLaunchKernelGGL(cnn_forward_kernel, dim3((unsigned)blocks), dim3(CNN_THREADS), fwd_lds, s, wt.codes, wt.n, D, params,
                         static_cast<double *>(nullptr), wt.rows, static_cast<const pln_tile *>(nullptr), static_cast<const uint16_t *>(nullptr),
                         groups, cnn_window_positions(D, wt));
    }
    cnn_level_io io = cnn_level_positions(plan, D, k);
    cnn_level_sources(plan, k, io);
    const uint64_t n = k == 0 ? n_rows : plan->cnn_levels[k - 1].n, groups = (n + 63) / 64;
    uint64_t blocks = (groups + CNN_THREADS / 64 - 1) / (CNN_THREADS / 64);
    if (blocks > (uint64_t)ws->num_cu * 16) blocks = (uint64_t)ws->num_cu * 16;
    if (k == 0 && io.p_lo >= io.p_hi && !getenv("BEAR_AMD_CNN_NO_HEAD_KERNEL")) {      // the contexts evaluate no position themselves: the head alone
      uint64_t hb = (n + 32 * 16 - 1) / (32 * 16);
      if (hb > (uint64_t)ws->num_cu * 2) hb = (uint64_t)ws->num_cu * 2;
      hipLaunchKernelGGL(cnn_forward_head_kernel, dim3((unsigned)hb), dim3(1024), 0, s, n, D, params, prior, t1_buf, io);
      continue;
    }
    hipLaunchKernelGGL(cnn_forward_kernel, dim3((unsigned)blocks), dim3(CNN_THREADS), fwd_lds, s,
                       k == 0 ? reinterpret_cast<const unsigned long long *>(kmer_code) : plan->cnn_levels[k - 1].codes, n, D, params,
                       k == 0 ? prior : static_cast<double *>(nullptr), k == 0 ? t1_buf : plan->cnn_levels[k - 1].rows,
                       static_cast<const pln_tile *>(nullptr), static_cast<const uint16_t *>(nullptr), groups, io);
  }
  HIP_TRY(hipGetLastError());
  return BEAR_OK;
}

int bear_cnn_forward_plan_f64(bear_ws *ws, const bear_plan *plan, const uint64_t *kmer_code, uint64_t n_rows, int lag, int filter_width,
                              int num_filters, int layer1_width, const double *params, double *prior, double *t1_save, void *stream) {
  int st = cnn_check(ws, lag, filter_width, num_filters, layer1_width);
  if (st != BEAR_OK) return st;
  if (!plan || plan->n_rows != n_rows || plan->device != ws->device) return BEAR_ERR_INVALID_ARG;
  if (n_rows == 0) return BEAR_OK;
  if (!kmer_code || !params || !prior || !t1_save || misaligned(t1_save) || (reinterpret_cast<uintptr_t>(prior) & 7u)) return BEAR_ERR_INVALID_ARG;
  if (!((plan->n_cnn_levels > 0 || plan->n_cnn_windows > 0) && plan->cnn_codes == kmer_code && plan->cnn_lag == lag && plan->cnn_fw == filter_width) ||
      getenv("BEAR_AMD_CNN_NO_LEVELS"))
    return bear_cnn_forward_f64(ws, kmer_code, n_rows, lag, filter_width, num_filters, layer1_width, params, prior, t1_save, stream);
  return cnn_forward_levels(ws, plan, cnn_make_dims(lag, filter_width), kmer_code, n_rows, params, prior, t1_save, static_cast<hipStream_t>(stream));
}

int bear_net_cnn_train_reduce_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_code, uint64_t n_rows,
                                  int lag, int filter_width, int num_filters, int layer1_width, const double *theta, double *prior_buf,
                                  double *t1_buf, double *grad_rows_buf, double eps, int train_ar, double *packed, void *stream) {
  int st = cnn_check(ws, lag, filter_width, num_filters, layer1_width);
  if (st != BEAR_OK) return st;
  if (!plan || !packed || !theta || !prior_buf || !t1_buf || !grad_rows_buf || !n_rows || !kmer_code) return BEAR_ERR_INVALID_ARG;
  if (plan->counts != counts || plan->n_rows != n_rows || plan->ncol != 5 || plan->device != ws->device) return BEAR_ERR_INVALID_ARG;
  if (misaligned(prior_buf) || misaligned(t1_buf) || misaligned(grad_rows_buf) || (reinterpret_cast<uintptr_t>(packed) & 7u))
    return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const cnn_dims D = cnn_make_dims(lag, filter_width);
  const double *params = theta + 1;
  if ((plan->n_cnn_levels > 0 || plan->n_cnn_windows > 0) && plan->cnn_codes == kmer_code && plan->cnn_lag == lag && plan->cnn_fw == filter_width &&
      plan->n_live_rows == n_rows && !getenv("BEAR_AMD_CNN_NO_LEVELS") && !cnn_parts_form_forced_off())
    return cnn_train_reduce_levels(ws, plan, D, kmer_code, n_rows, theta, prior_buf, t1_buf, grad_rows_buf, eps, train_ar, packed, s);
  {
    const size_t lds = sizeof(double) * (BEAR_EXPTAB_N + (size_t)filter_width * 6 * CNN_NF + (CNN_THREADS / 64) * CNN_FWD_SCRATCH);
    // only the contexts that hold training counts: the DM kernel reads nobody else's prior row (their gradient rows are zero).
    // Only together with the backward kernel that walks the same lists (shapes whose LDS does not fit take all rows in both).
    int bw_waves = 0, bw_parts = 0;
    size_t bw_lds = 0;
    uint64_t bw_blocks = 0;
    st = cnn_backward_grid(ws, D, n_rows, filter_width, &bw_waves, &bw_parts, &bw_lds, &bw_blocks, s, 0);
    if (st != BEAR_OK) return st;
    const bool lists = plan->live && bw_parts && plan->n_live_rows < n_rows;   // all rows live: plain groups of rows
    const uint64_t groups = lists ? plan->n_tiles : (n_rows + 63) / 64;
    uint64_t blocks = (groups + CNN_THREADS / 64 - 1) / (CNN_THREADS / 64);
    if (blocks > (uint64_t)ws->num_cu * 16) blocks = (uint64_t)ws->num_cu * 16;
    hipLaunchKernelGGL(cnn_forward_kernel, dim3((unsigned)blocks), dim3(CNN_THREADS), lds, s,
                       reinterpret_cast<const unsigned long long *>(kmer_code), n_rows, D, params, prior_buf, t1_buf,
                       lists ? plan->tiles : nullptr, lists ? plan->live : nullptr, groups, cnn_all_positions(D));
  }
  bear_params only_eps;
  memset(&only_eps, 0, sizeof(only_eps));
  only_eps.eps = eps;
  st = launch_prior_plan_grad(ws, plan, prior_buf, only_eps, theta, train_ar, 1, packed, grad_rows_buf, s);   // softmax rows: normalised
  if (st != BEAR_OK) return st;
  return launch_cnn_backward(ws, D, kmer_code, n_rows, filter_width, params, t1_buf, prior_buf, grad_rows_buf, packed + 2, s, 0, plan);
}

int bear_net_cnn_train_step_f64(bear_ws *ws, const bear_plan *plan, const uint32_t *counts, const uint64_t *kmer_code, uint64_t n_rows,
                                int lag, int filter_width, int num_filters, int layer1_width, double *theta, double *adam_m,
                                double *adam_v, double *adam_t, double *prior_buf, double *t1_buf, double *grad_rows_buf,
                                double *packed, double eps, int train_ar, double learning_rate, double scale, double *loss_buf,
                                uint64_t loss_cap, void *stream) {
  if (!adam_m || !adam_v || !adam_t) return BEAR_ERR_INVALID_ARG;
  int st = bear_net_cnn_train_reduce_f64(ws, plan, counts, kmer_code, n_rows, lag, filter_width, num_filters, layer1_width, theta, prior_buf,
                                         t1_buf, grad_rows_buf, eps, train_ar, packed, stream);
  if (st != BEAR_OK) return st;
  const cnn_dims D = cnn_make_dims(lag, filter_width);
  return launch_train_apply(theta, 1 + D.total, packed, adam_m, adam_v, adam_t, learning_rate, scale, train_ar, loss_buf, loss_cap,
                            static_cast<hipStream_t>(stream));
}

int bear_stream_read(bear_ws *ws, const void *src, uint64_t n_bytes, void *stream) {
  int st = check_ws(ws);
  if (st != BEAR_OK) return st;
  if (!src || misaligned(src)) return BEAR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(stream_read_kernel, dim3(ws->num_cu * 16), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4 *>(src), n_bytes / 16, reinterpret_cast<uint32_t *>(ws->dbg));
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

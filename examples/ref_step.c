/* examples/ref_step.c -- the C ABI from plain C (no Python, no torch): one bear_ref training-step evaluation
 * (bear_model/bear_ref.py:207-259 with the stop net function) on a synthetic k=13 table generated on the device.
 *
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/ref_step.c -Lbear_amd -lbear_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/bear_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/ref_step && /tmp/ref_step 100000000
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "bear_hip.h"

#define CHECK(st, what)                                                          \
  do {                                                                           \
    int _s = (st);                                                               \
    if (_s != BEAR_OK) {                                                         \
      fprintf(stderr, "%s: %s (hip %d)\n", what, bear_strerror(_s), bear_last_hip_error()); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char **argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 1000000ull;
  bear_ws *ws = NULL;
  CHECK(bear_ws_create(0, &ws), "bear_ws_create");
  uint32_t *train = NULL, *ref = NULL;
  double *out = NULL, host[4];
  if (hipMalloc((void **)&train, n * 20) != hipSuccess || hipMalloc((void **)&ref, n * 20) != hipSuccess ||
      hipMalloc((void **)&out, 4 * sizeof(double)) != hipSuccess) {
    fprintf(stderr, "hipMalloc failed\n");
    return 1;
  }
  CHECK(bear_synth_counts_u32(20211012, 0, n, 0, train, NULL, ref, NULL), "bear_synth_counts_u32");
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  bear_plan *plan = NULL;
  CHECK(bear_plan_create(ws, train, n, 4, &plan), "bear_plan_create");
  const double h_s = 0.0, tau_s = log(1.0 / 30.0), nu_s = -log(100.0);      /* the reference's initial values */
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  CHECK(bear_dm_ref_plan_f64(ws, plan, train, ref, n, h_s, tau_s, nu_s, 1e-7, 0, out, NULL), "bear_dm_ref_plan_f64");
  hipEventRecord(e0, NULL);
  for (int k = 0; k < 20; ++k)
    CHECK(bear_dm_ref_plan_f64(ws, plan, train, ref, n, h_s, tau_s, nu_s, 1e-7, 0, out, NULL), "bear_dm_ref_plan_f64");
  hipEventRecord(e1, NULL);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  printf("n = %llu contexts: sum LL = %.12e, d/dh_s = %.6e, d/dtau_s = %.6e, d/dnu_s = %.6e; %.3f ms per step (%.1f Gctx/s)\n",
         (unsigned long long)n, host[0], host[1], host[2], host[3], ms / 20, (double)n / (ms / 20 * 1e-3) / 1e9);
  bear_plan_destroy(plan);
  bear_ws_destroy(ws);
  hipFree(train);
  hipFree(ref);
  hipFree(out);
  return 0;
}

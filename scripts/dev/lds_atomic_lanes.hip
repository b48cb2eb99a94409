// Dev micro-benchmark: cost of one LDS fp64 atomic wave-instruction by the number of ACTIVE lanes and by address pattern
// (one 1024-thread block per CU).  Prints cycles of CU time per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(1024) void k(double *out, int iters, int active, int mode) {
  __shared__ double tab[8192];
  for (int i = threadIdx.x; i < 8192; i += 1024) tab[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  if (lane < active) {
    for (int i = 0; i < iters; ++i) {
      h = h * 1664525u + 1013904223u;
      uint32_t a;
      if (mode == 0) a = (lane * 8u + ((h >> 10) & 7u)) & 8191u;             // distinct addresses, distinct banks
      else if (mode == 1) a = ((h >> 10) % 36u) * 5u;                        // random among 36 rows (pair-table pattern)
      else if (mode == 2) a = (lane & 3u) * 2048u + ((h >> 10) & 1u);        // 4 addresses: heavy same-address collisions
      else a = ((lane & 15u) + 16u * ((h >> 10) & 3u) * 4u) & 8191u;         // replicas on consecutive banks
      atomicAdd(&tab[a], 1.0);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[0] + tab[1];
}
int main() {
  double *out;
  (void)hipMalloc(&out, 256 * sizeof(double));
  const int iters = 2048;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode)
    for (int active : {1, 4, 16, 64}) {
      hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, iters, active, mode);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, iters, active, mode);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      // 16 waves per CU issue `iters` wave-instructions each
      printf("mode %d active %2d: %.3f ms -> %.1f cycles of CU time per wave-instruction (2.4 GHz)\n", mode, active, ms,
             ms * 1e-3 * 2.4e9 / (16.0 * iters));
    }
  return 0;
}

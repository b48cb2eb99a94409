// Dev micro-benchmark (round 4): ds_add_f64 wave-instructions, 64 active lanes, by ADDRESS PATTERN of the lanes' 8-byte slots.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, int iters) {
  __shared__ double tab[8192];
  for (int i = threadIdx.x; i < 8192; i += 1024) tab[i] = 0.0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    const uint32_t r = (h >> 10);
    uint32_t slot;
    if (MODE == 0) slot = lane;                                             // consecutive
    else if (MODE == 1) slot = 2u * lane;                                   // every second slot (a pair's first contexts, dense rows)
    else if (MODE == 2) slot = (lane >> 1) * 6u + (lane & 1u) * 2u + ((lane >> 1) & 0u);   // base-6 gaps, every second: 0,2,6,8,12,...
    else if (MODE == 3) slot = (lane / 4u) * 6u + (lane & 3u);              // base-6 gaps, consecutive contexts: 0,1,2,3,6,7,8,9,...
    else if (MODE == 4) slot = (r ^ (lane * 2654435761u)) % 216u;           // random among 216 rows (collisions possible)
    else slot = (lane & 31u) * 2u + (lane >> 5) * 1u;                       // two interleaved blocks of 32 (two prefix blocks in a wave)
    atomicAdd(&tab[(slot + 256u * (r & 15u)) & 8191u], 1.0);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[0] + tab[1];
}
template <int MODE>
void run(const char *name, double *out) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms -> %.1f cycles per wave-instruction\n", name, ms, ms * 1e-3 * 2.4e9 / (16.0 * iters));
}
int main() {
  double *out;
  (void)hipMalloc(&out, 256 * sizeof(double));
  run<0>("consecutive slots", out);
  run<1>("every second slot", out);
  run<2>("base-6 gaps, every second (0,2,6,8,..)", out);
  run<3>("base-6 gaps, consecutive (0,1,2,3,6,..)", out);
  run<4>("random among 216 rows", out);
  run<5>("two interleaved blocks of 32", out);
  return 0;
}

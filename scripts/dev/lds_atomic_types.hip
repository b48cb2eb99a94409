// Dev micro-benchmark (round 4): cost of one LDS atomic wave-instruction by OPERAND TYPE -- fp64 add, u64 add, f32 add, u32 add --
// 64 active lanes on distinct consecutive 8-byte (4-byte) slots, one 1024-thread block per CU.  Cycles of CU time per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int TYPE>
__global__ __launch_bounds__(1024) void k(double *out, int iters) {
  __shared__ unsigned long long tab[8192];
  for (int i = threadIdx.x; i < 8192; i += 1024) tab[i] = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    const uint32_t a = (lane + 64u * ((h >> 10) & 63u)) & 8191u;   // a wave's 64 lanes: 64 consecutive slots, a random group of them per step
    if (TYPE == 0) atomicAdd(reinterpret_cast<double *>(&tab[a]), 1.0);
    else if (TYPE == 1) atomicAdd(&tab[a], 1ull);
    else if (TYPE == 2) atomicAdd(reinterpret_cast<float *>(&tab[a]), 1.0f);
    else if (TYPE == 3) atomicAdd(reinterpret_cast<unsigned int *>(&tab[a]), 1u);
    else {          // plain read-modify-write (not atomic: what a private table would cost)
      double *p = reinterpret_cast<double *>(&tab[a]);
      *p = *p + 1.0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (double)tab[0] + (double)tab[1];
}
template <int TYPE>
void run(const char *name, double *out) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<TYPE>, dim3(256), dim3(1024), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<TYPE>, dim3(256), dim3(1024), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms -> %.1f cycles of CU time per wave-instruction (2.4 GHz, 16 waves per CU)\n", name, ms, ms * 1e-3 * 2.4e9 / (16.0 * iters));
}
int main() {
  double *out;
  (void)hipMalloc(&out, 256 * sizeof(double));
  run<0>("ds_add_f64", out);
  run<1>("ds_add_u64", out);
  run<2>("ds_add_f32", out);
  run<3>("ds_add_u32", out);
  run<4>("read + add + write (f64)", out);
  return 0;
}

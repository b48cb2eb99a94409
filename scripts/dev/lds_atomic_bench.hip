// Dev micro-benchmark: LDS atomic-add throughput by operand type on gfx950 (one 1024-thread block per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <typename T>
__global__ __launch_bounds__(1024) void k(T *out, int iters, int span, int same) {
  __shared__ T tab[4096];
  for (int i = threadIdx.x; i < 4096; i += 1024) tab[i] = (T)0;
  __syncthreads();
  uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    const uint32_t a = same ? ((h >> 10) % span) : ((threadIdx.x & 63u) * 8u + ((h >> 10) & 7u)) % 4096u;
    atomicAdd(&tab[a], (T)1);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[0] + tab[1];
}
template <typename T>
void run(const char *name, int span, int same) {
  T *out;
  hipMalloc(&out, 256 * sizeof(T));
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<T>, dim3(256), dim3(1024), 0, 0, out, iters, span, same);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<T>, dim3(256), dim3(1024), 0, 0, out, iters, span, same);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // lane-atomics per clock per CU at 2.4 GHz
  printf("%-8s span %4d mode %d: %.3f ms  -> %.2f lane-atomics/clk/CU\n", name, span, same, ms, 1024.0 * iters / (ms * 1e-3 * 2.4e9));
  hipFree(out);
}
int main() {
  for (int same = 0; same < 2; ++same) {
    const int span = 1260;
    run<unsigned int>("u32", span, same);
    run<unsigned long long>("u64", span, same);
    run<float>("f32", span, same);
    run<double>("f64", span, same);
  }
  return 0;
}

// Dev micro-benchmark: how long does a wave spend ISSUING global_load_lds_dwordx4 (LDS-DMA) instructions
// compared with plain global_load_dwordx4 into registers?  One 1024-thread block per CU, every wave issues
// K loads of 1 KiB back to back, then waits.  Reports s_memtime ticks per instruction for issue and for the wait.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define K 8
__global__ __launch_bounds__(1024) void k_dma(const unsigned char *src, size_t stride, unsigned long long *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long t_issue = 0, t_wait = 0;
  const unsigned char *base = src + ((size_t)blockIdx.x * 16 + wave) * stride + lane * 16;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const unsigned char *g = base + ((size_t)it * K + p) * 1024;
      const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds + (wave * K + p) * 1024);
      {
      // M0 is compiler-reserved: saved and restored inside the statement that uses it (no "m0" clobber: that is undefined behaviour)
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0)
                   : "v"(g), "s"(m)
                   : "memory");
    }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    t_issue += t1 - t0;
    t_wait += t2 - t1;
  }
  if (lane == 0) {
    out[(blockIdx.x * 16 + wave) * 2] = t_issue;
    out[(blockIdx.x * 16 + wave) * 2 + 1] = t_wait;
  }
}
__global__ __launch_bounds__(1024) void k_reg(const unsigned char *src, size_t stride, unsigned long long *out, int iters, uint4 *sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long t_issue = 0, t_wait = 0;
  const unsigned char *base = src + ((size_t)blockIdx.x * 16 + wave) * stride + lane * 16;
  uint4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    uint4 r[K];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int p = 0; p < K; ++p) {
      const uint4 *g = reinterpret_cast<const uint4 *>(base + ((size_t)it * K + p) * 1024);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[p]) : "v"(g) : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int p = 0; p < K; ++p) {
      reinterpret_cast<uint4 *>(lds + (wave * K + p) * 1024)[lane] = r[p];
      acc.x ^= r[p].x;
    }
    t_issue += t1 - t0;
    t_wait += t2 - t1;
  }
  if (lane == 0) {
    out[(blockIdx.x * 16 + wave) * 2] = t_issue;
    out[(blockIdx.x * 16 + wave) * 2 + 1] = t_wait;
  }
  if (acc.x == 0x12345678u) sink[0] = acc;
}
// Only the first W waves of the block issue LDS-DMA (the others exit): can a few dedicated waves saturate HBM?
// Each DMA wave streams `per_wave` KiB pieces through a ring of 32 KiB of LDS, never more than 24 in flight.
__global__ __launch_bounds__(1024) void k_few(const unsigned char *src, size_t bytes_per_block, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if ((int)wave >= W) return;
  const size_t pieces = bytes_per_block / 1024;
  const unsigned char *base = src + (size_t)blockIdx.x * bytes_per_block + lane * 16;
  uint32_t k = 0;
  for (size_t p = wave; p < pieces; p += W, ++k) {
    const unsigned char *g = base + p * 1024;
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds + (wave * 8 + (k & 7u)) * 1024);
    {
      // M0 is compiler-reserved: saved and restored inside the statement that uses it (no "m0" clobber: that is undefined behaviour)
      uint32_t keep_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep_m0)
                   : "v"(g), "s"(m)
                   : "memory");
    }
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)
int main() {
  const int iters = 64, blocks = 256;
  const int lds_bytes = 16 * K * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dma), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_reg), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const size_t stride = (size_t)iters * K * 1024;
  unsigned char *src;
  unsigned long long *out;
  uint4 *sink;
  CK(hipMalloc(&src, stride * blocks * 16 + 4096));
  CK(hipMemset(src, 1, stride * blocks * 16));
  CK(hipMalloc(&out, sizeof(unsigned long long) * blocks * 32));
  CK(hipMalloc(&sink, 64));
  std::vector<unsigned long long> h(blocks * 32);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_dma, dim3(blocks), dim3(1024), lds_bytes, 0, src, stride, out, iters);
      else hipLaunchKernelGGL(k_reg, dim3(blocks), dim3(1024), lds_bytes, 0, src, stride, out, iters, sink);
      CK(hipGetLastError());
      hipEventRecord(e1);
      CK(hipEventSynchronize(e1));
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), out, sizeof(unsigned long long) * blocks * 32, hipMemcpyDeviceToHost);
      double ti = 0, tw = 0;
      for (int i = 0; i < blocks * 16; ++i) { ti += h[2 * i]; tw += h[2 * i + 1]; }
      ti /= blocks * 16.0 * iters * K;
      tw /= blocks * 16.0 * iters;
      printf("%s rep %d: %.3f ms, %.1f GB/s | issue %.0f ticks per load instruction, wait %.0f ticks per batch of %d\n",
             mode == 0 ? "lds-dma " : "register", rep, ms, stride * blocks * 16 / (ms * 1e-3) / 1e9, ti, tw, K);
    }
  }
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_few), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const size_t bpb = stride * 16;
  for (int W : {1, 2, 4, 8, 16}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_few, dim3(blocks), dim3(1024), lds_bytes, 0, src, bpb, W);
      CK(hipGetLastError());
      hipEventRecord(e1);
      CK(hipEventSynchronize(e1));
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("dma waves per CU %2d: %.3f ms, %.1f GB/s\n", W, ms, bpb * blocks / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}

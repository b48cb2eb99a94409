// bear_count.hip -- k-mer transition counting on the device: the count table of summarize.py built straight from the
// sequences.
//
// The reference gets there in three stages (bear_model/summarize.py): (1) write prefix / suffix / full FASTQ files per
// input, (2) run the external KMC counter on each, (3) heap-merge the sorted KMC dumps into rows
// `kmer \t [[A,C,G,T,$ per group]...]` (Register / Consolidate, summarize.py:380-622).  What the three stages compute is
// stated by the reference's own test (bear_model/tests/test_summarize.py:88-115): for every sequence and lag L,
//     full = '[' * L + seq + ']' ;  for j in [L, len(full)):  counts[full[j-L:j]][group][full[j]] += 1 .
// On an MI355X that is one pass per lag over the resident text: every transition becomes a (context code, group * 5 +
// next letter) pair, the pairs are radix-sorted by context (rocPRIM radix_sort_pairs: a library sort is the right tool for
// the sort itself), and a run-length pass turns runs of equal contexts into rows and scatters the pair values into the
// planar uint32 [group][row][5] slabs the training kernels consume -- no KMC, no intermediate files, and the table can
// go to training without ever being text.
//
// Text layout [dev]: per sequence  5 (start marker), letters 0..3 (6 = any other character), 4 (stop).  A transition sits
// at every position holding 0..4; its context is the L codes before it, read back until the start marker, the rest
// filled with the start symbol.  Transitions whose context or next letter contains a 6 are dropped (KMC drops k-mers
// with non-ACGT letters likewise).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>

#include "../../include/bear_hip.h"
#include "bear_levels.h"

namespace {
thread_local int g_count_hip_error = 0;
#define CNT_TRY(expr)                    \
  do {                                   \
    hipError_t _e = (expr);              \
    if (_e != hipSuccess) {              \
      g_count_hip_error = (int)_e;       \
      st = BEAR_ERR_HIP;                 \
      goto done;                         \
    }                                    \
  } while (0)

// dropped transitions carry a key with only bit 3*lag set: it sorts behind every context, and the radix sort needs to
// look at 3*lag + 1 bits only (5 passes instead of 8 at lag 13)
__host__ __device__ inline uint64_t cnt_invalid(int lag) { return 1ull << (3 * lag); }

// key: the context as the packed k-mer code of bear_pack_kmers_u64 (letter l of the k-mer in bits [3l, 3l+3); 4 = '[')
__global__ __launch_bounds__(256) void cnt_emit_kernel(const uint8_t *__restrict__ text, const uint8_t *__restrict__ grp,
                                                       uint64_t n_pos, int lag, uint64_t *__restrict__ keys,
                                                       uint32_t *__restrict__ vals) {
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < n_pos; t += (uint64_t)gridDim.x * 256) {
    const uint32_t nx = text[t];
    const uint64_t CNT_INVALID = cnt_invalid(lag);
    uint64_t key = CNT_INVALID;
    if (nx <= 4u) {
      key = 0;
      bool started = false, bad = false;
      for (int i = 1; i <= lag; ++i) {                 // letter lag - i of the k-mer
        uint32_t c = 4u;
        if (!started) {
          c = (t >= (uint64_t)i) ? text[t - i] : 5u;
          if (c == 5u) {
            started = true;
            c = 4u;
          }
        }
        bad |= c > 4u;
        key |= (uint64_t)c << (3 * (lag - i));
      }
      if (bad) key = CNT_INVALID;
    }
    keys[t] = key;
    vals[t] = (uint32_t)grp[t] * 5u + (nx <= 4u ? nx : 0u);
  }
}

__global__ __launch_bounds__(256) void cnt_flag_kernel(const uint64_t *__restrict__ keys, uint64_t n, int lag,
                                                       uint32_t *__restrict__ flags) {
  const uint64_t CNT_INVALID = cnt_invalid(lag);
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (uint64_t)gridDim.x * 256) {
    const uint64_t k = keys[t];
    flags[t] = (k != CNT_INVALID && (t == 0 || keys[t - 1] != k)) ? 1u : 0u;
  }
}

// rows: inclusive scan of the run-start flags (row index + 1).  One thread per sorted pair.
__global__ __launch_bounds__(256) void cnt_scatter_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                          const uint32_t *__restrict__ rows, uint64_t n, uint64_t n_rows, int lag,
                                                          uint32_t n_groups, uint8_t *__restrict__ kmers, uint64_t *__restrict__ codes,
                                                          uint32_t *__restrict__ counts) {
  const uint64_t CNT_INVALID = cnt_invalid(lag);
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (uint64_t)gridDim.x * 256) {
    const uint64_t k = keys[t];
    if (k == CNT_INVALID) continue;
    const uint64_t row = (uint64_t)rows[t] - 1u;
    const uint32_t v = vals[t];
    if (v / 5u < n_groups) atomicAdd(&counts[((uint64_t)(v / 5u) * n_rows + row) * 5u + (v % 5u)], 1u);   // group ids beyond n_groups are ignored
    if (t == 0 || keys[t - 1] != k) {                // run start: name the row
      if (codes) {
        uint64_t packed = k;
        for (int l = lag; l < 21; ++l) packed |= 5ull << (3 * l);     // positions >= lag hold 5 (bear_pack_kmers_u64)
        codes[row] = packed;
      }
      if (kmers)
        for (int l = 0; l < lag; ++l) kmers[row * (uint64_t)lag + l] = (uint8_t)("ACGT["[(k >> (3 * l)) & 7ull]);
    }
  }
}

unsigned grid_for(uint64_t n) {
  uint64_t b = (n + 255) / 256;
  return (unsigned)(b > (1u << 20) ? (1u << 20) : (b ? b : 1));
}
}  // namespace

struct bear_kmer_sort {
  uint64_t n_pos, n_rows;
  int lag;
  uint64_t *keys;   // sorted
  uint32_t *vals;   // sorted with the keys
  uint32_t *rows;   // inclusive scan of run starts
};

extern "C" {

int bear_count_last_hip_error(void) { return g_count_hip_error; }

int bear_kmer_sort_create(const uint8_t *text, const uint8_t *group, uint64_t n_pos, int lag, bear_kmer_sort **out,
                          uint64_t *n_rows_out, void *stream) {
  if (!out || !n_rows_out || lag < 1 || lag > 21) return BEAR_ERR_INVALID_ARG;
  *out = nullptr;
  *n_rows_out = 0;
  if (n_pos && (!text || !group)) return BEAR_ERR_INVALID_ARG;
  if (n_pos >= 0xffffffffull) return BEAR_ERR_INVALID_ARG;   // row indices are 32-bit: shard the text above 4e9 positions
  int st = BEAR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bear_kmer_sort *h = new (std::nothrow) bear_kmer_sort();
  if (!h) return BEAR_ERR_NOMEM;
  h->n_pos = n_pos;
  h->lag = lag;
  uint64_t *keys_in = nullptr;
  uint32_t *vals_in = nullptr, *flags = nullptr;
  void *temp = nullptr;
  size_t tb_sort = 0, tb_scan = 0;
  uint32_t last = 0;
  if (n_pos == 0) {
    *out = h;
    return BEAR_OK;
  }
  CNT_TRY(hipMalloc(&keys_in, n_pos * 8));
  CNT_TRY(hipMalloc(&vals_in, n_pos * 4));
  CNT_TRY(hipMalloc(&h->keys, n_pos * 8));
  CNT_TRY(hipMalloc(&h->vals, n_pos * 4));
  hipLaunchKernelGGL(cnt_emit_kernel, dim3(grid_for(n_pos)), dim3(256), 0, s, text, group, n_pos, lag, keys_in, vals_in);
  CNT_TRY(hipGetLastError());
  CNT_TRY(rocprim::radix_sort_pairs(nullptr, tb_sort, keys_in, h->keys, vals_in, h->vals, n_pos, 0u, (unsigned)(3 * lag + 1), s));
  CNT_TRY(hipMalloc(&temp, tb_sort ? tb_sort : 8));
  CNT_TRY(rocprim::radix_sort_pairs(temp, tb_sort, keys_in, h->keys, vals_in, h->vals, n_pos, 0u, (unsigned)(3 * lag + 1), s));
  CNT_TRY(hipStreamSynchronize(s));
  (void)hipFree(temp);
  temp = nullptr;
  (void)hipFree(keys_in);
  keys_in = nullptr;
  flags = vals_in;   // reuse
  vals_in = nullptr;
  CNT_TRY(hipMalloc(&h->rows, n_pos * 4));
  hipLaunchKernelGGL(cnt_flag_kernel, dim3(grid_for(n_pos)), dim3(256), 0, s, h->keys, n_pos, lag, flags);
  CNT_TRY(hipGetLastError());
  CNT_TRY(rocprim::inclusive_scan(nullptr, tb_scan, flags, h->rows, n_pos, rocprim::plus<uint32_t>(), s));
  CNT_TRY(hipMalloc(&temp, tb_scan ? tb_scan : 8));
  CNT_TRY(rocprim::inclusive_scan(temp, tb_scan, flags, h->rows, n_pos, rocprim::plus<uint32_t>(), s));
  CNT_TRY(hipMemcpyAsync(&last, h->rows + (n_pos - 1), 4, hipMemcpyDeviceToHost, s));
  CNT_TRY(hipStreamSynchronize(s));
  h->n_rows = last;
done:
  if (temp) (void)hipFree(temp);
  if (keys_in) (void)hipFree(keys_in);
  if (vals_in) (void)hipFree(vals_in);
  if (flags) (void)hipFree(flags);
  if (st != BEAR_OK) {
    if (h->keys) (void)hipFree(h->keys);
    if (h->vals) (void)hipFree(h->vals);
    if (h->rows) (void)hipFree(h->rows);
    delete h;
    return st;
  }
  *out = h;
  *n_rows_out = h->n_rows;
  return BEAR_OK;
}

int bear_kmer_sort_reduce(const bear_kmer_sort *h, int n_groups, uint8_t *kmers, uint64_t *kmer_code, uint32_t *counts,
                          void *stream) {
  if (!h || n_groups < 1 || n_groups > 255) return BEAR_ERR_INVALID_ARG;
  if (h->n_rows == 0) return BEAR_OK;
  if (!counts) return BEAR_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(counts, 0, (size_t)n_groups * h->n_rows * 5 * 4, s) != hipSuccess) return BEAR_ERR_HIP;
  hipLaunchKernelGGL(cnt_scatter_kernel, dim3(grid_for(h->n_pos)), dim3(256), 0, s, h->keys, h->vals, h->rows, h->n_pos, h->n_rows,
                     h->lag, (uint32_t)n_groups, kmers, kmer_code, counts);
  if (hipGetLastError() != hipSuccess) return BEAR_ERR_HIP;
  return BEAR_OK;
}

int bear_kmer_sort_destroy(bear_kmer_sort *h) {
  if (!h) return BEAR_OK;
  if (h->keys) (void)hipFree(h->keys);
  if (h->vals) (void)hipFree(h->vals);
  if (h->rows) (void)hipFree(h->rows);
  delete h;
  return BEAR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ k-mer order of a batch (fused AR-function kernels)
// The sums of a training step do not depend on the order of a batch's rows, and the fused linear / convolutional kernels run
// about twice as fast when consecutive contexts share their leading letters (kernels_linear.h phase C, kernels_cnn.h shared
// windows).  bear_kmer_order_u64 returns the permutation that sorts packed contexts lexicographically, FIRST letter most
// significant (a radix sort over 3 lag bits of the letter-reversed code); bear_gather_rows applies a permutation to any
// row-major slab (count rows, packed contexts, k-mer bytes) -- once per batch, before its plan is built.
namespace {
__global__ __launch_bounds__(256) void order_keys_kernel(const unsigned long long *__restrict__ code, uint64_t n, int lag,
                                                         unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const unsigned long long w = code[i];
    unsigned long long k = 0ull;
    for (int l = 0; l < lag; ++l) k = (k << 3) | ((w >> (3 * l)) & 7ull);      // letter 0 ends up in the top field
    keys[i] = k;
    vals[i] = (uint32_t)i;
  }
}
__global__ __launch_bounds__(256) void gather_rows_kernel(const unsigned char *__restrict__ src, const uint32_t *__restrict__ perm,
                                                          unsigned char *__restrict__ dst, uint64_t n_rows, uint32_t row_bytes) {
  if ((row_bytes & 3u) == 0u) {      // word rows: one thread per 4-byte word, consecutive threads on consecutive words of a row
    const uint32_t wpr = row_bytes >> 2;
    const uint64_t total = n_rows * wpr;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
      const uint64_t r = t / wpr;
      const uint32_t k = (uint32_t)(t - r * wpr);
      reinterpret_cast<uint32_t *>(dst)[t] = reinterpret_cast<const uint32_t *>(src)[(uint64_t)perm[r] * wpr + k];
    }
  } else {
    const uint64_t total = n_rows * row_bytes;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
      const uint64_t r = t / row_bytes;
      dst[t] = src[(uint64_t)perm[r] * row_bytes + (t - r * row_bytes)];
    }
  }
}
}  // namespace

extern "C" int bear_kmer_order_u64(const uint64_t *kmer_code, uint64_t n_rows, int lag, uint32_t *perm, void *scratch,
                                   uint64_t *scratch_bytes, void *stream) {
  if (!scratch_bytes || lag < 1 || lag > 21 || n_rows > 0xffffffffull) return BEAR_ERR_INVALID_ARG;
  // caller-owned scratch: two key arrays, the identity values, rocPRIM's temporary storage (each piece 256-byte aligned)
  const uint64_t keys_b = (n_rows * 8 + 255) & ~255ull, vals_b = (n_rows * 4 + 255) & ~255ull;
  size_t tb = 0;
  int st = BEAR_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_rows)
    CNT_TRY(rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint32_t *)nullptr,
                                      (uint32_t *)nullptr, n_rows, 0u, (unsigned)(3 * lag), s));
  {
    const uint64_t need = n_rows ? 2 * keys_b + vals_b + (tb ? tb : 8) : 0;
    if (!scratch) {            // size query
      *scratch_bytes = need;
      return BEAR_OK;
    }
    if (n_rows == 0) return BEAR_OK;
    if (!kmer_code || !perm || *scratch_bytes < need || (reinterpret_cast<uintptr_t>(scratch) & 255)) return BEAR_ERR_INVALID_ARG;
    unsigned char *base = static_cast<unsigned char *>(scratch);
    unsigned long long *keys_in = reinterpret_cast<unsigned long long *>(base);
    unsigned long long *keys_out = reinterpret_cast<unsigned long long *>(base + keys_b);
    uint32_t *vals_in = reinterpret_cast<uint32_t *>(base + 2 * keys_b);
    void *temp = base + 2 * keys_b + vals_b;
    hipLaunchKernelGGL(order_keys_kernel, dim3(grid_for(n_rows)), dim3(256), 0, s, reinterpret_cast<const unsigned long long *>(kmer_code),
                       n_rows, lag, keys_in, vals_in);
    CNT_TRY(hipGetLastError());
    // stream-ordered from here on: the scratch is the caller's, nothing is freed and nothing waits on the host
    CNT_TRY(rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, vals_in, perm, n_rows, 0u, (unsigned)(3 * lag), s));
  }
done:
  return st;
}

extern "C" int bear_gather_rows(const void *src, const uint32_t *perm, void *dst, uint64_t n_rows, uint32_t row_bytes, void *stream) {
  if ((n_rows && (!src || !perm || !dst)) || row_bytes == 0 || src == dst) return BEAR_ERR_INVALID_ARG;
  if (n_rows == 0) return BEAR_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n_rows * ((row_bytes & 3u) ? row_bytes : row_bytes >> 2))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const unsigned char *>(src), perm, static_cast<unsigned char *>(dst),
                     n_rows, row_bytes);
  return hipGetLastError() == hipSuccess ? BEAR_OK : BEAR_ERR_HIP;
}

// ------------------------------------------------------------------ prefix levels of a sorted batch (bear_levels.h, kernels_cnn.h)
namespace {
__global__ __launch_bounds__(256) void level_flag_kernel(const unsigned long long *__restrict__ codes, uint64_t n, unsigned long long mask,
                                                         uint32_t *__restrict__ flag) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
    flag[i] = (i == 0 || ((codes[i] ^ codes[i - 1]) & mask) != 0ull) ? 1u : 0u;
}
// scan[i] = number of runs that start at or before row i: row i belongs to run scan[i] - 1; a run's first row writes the run's record
__global__ __launch_bounds__(256) void level_compact_kernel(const unsigned long long *__restrict__ codes, uint64_t n, unsigned long long mask,
                                                            unsigned long long fill, const uint32_t *__restrict__ flag,
                                                            uint32_t *__restrict__ scan_to_parent, unsigned long long *__restrict__ out_codes,
                                                            uint32_t *__restrict__ child_start, uint64_t n_runs) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint32_t run = scan_to_parent[i] - 1u;
    if (flag[i]) {
      out_codes[run] = (codes[i] & mask) | fill;
      child_start[run] = (uint32_t)i;
    }
    scan_to_parent[i] = run;
    if (i == 0) child_start[n_runs] = (uint32_t)n;
  }
}
}  // namespace

namespace {
struct rec16 { unsigned long long a, b; };
struct rec16_less {
  __device__ __host__ bool operator()(const rec16 &x, const rec16 &y) const { return x.a < y.a || (x.a == y.a && x.b < y.b); }
};
template <typename T, typename Less>
int canonical_sort(T *d, uint64_t n, Less less, hipStream_t s) {
  int st = BEAR_OK;
  void *temp = nullptr;
  size_t tb = 0;
  T *out = nullptr;
  CNT_TRY(hipMalloc(&out, n * sizeof(T)));
  CNT_TRY(rocprim::merge_sort(nullptr, tb, d, out, n, less, s));
  CNT_TRY(hipMalloc(&temp, tb ? tb : 8));
  CNT_TRY(rocprim::merge_sort(temp, tb, d, out, n, less, s));
  CNT_TRY(hipMemcpyAsync(d, out, n * sizeof(T), hipMemcpyDeviceToDevice, s));
  CNT_TRY(hipStreamSynchronize(s));
done:
  if (temp) (void)hipFree(temp);
  if (out) (void)hipFree(out);
  return st;
}
}  // namespace

int bear_canonical_order(void *records, uint64_t n, int width, hipStream_t s) {
  if (n < 2) return BEAR_OK;
  if (!records) return BEAR_ERR_INVALID_ARG;
  if (width == 16) return canonical_sort(static_cast<rec16 *>(records), n, rec16_less(), s);
  if (width == 8) return canonical_sort(static_cast<unsigned long long *>(records), n, rocprim::less<unsigned long long>(), s);
  if (width == 4) return canonical_sort(static_cast<uint32_t *>(records), n, rocprim::less<uint32_t>(), s);
  return BEAR_ERR_INVALID_ARG;
}

// ------------------------------------------------------------------ window tables (bear_levels.h, kernels_cnn.h)
namespace {
__global__ __launch_bounds__(256) void window_keys_kernel(const unsigned long long *__restrict__ codes, uint64_t n, int shift, unsigned long long mask,
                                                          unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    keys[i] = (codes[i] >> shift) & mask;
    vals[i] = (uint32_t)i;
  }
}
// sorted position j belongs to run scan[j] - 1; a run's first position writes the window's record; every position its context's row
__global__ __launch_bounds__(256) void window_compact_kernel(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ perm, uint64_t n,
                                                             int shift, unsigned long long fill, const uint32_t *__restrict__ flag,
                                                             const uint32_t *__restrict__ scan, unsigned long long *__restrict__ out_codes,
                                                             uint32_t *__restrict__ child_start, uint32_t *__restrict__ row_of_context,
                                                             uint64_t n_runs) {
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) {
    const uint32_t run = scan[j] - 1u;
    if (flag[j]) {
      out_codes[run] = (keys[j] << shift) | fill;
      child_start[run] = (uint32_t)j;
    }
    row_of_context[perm[j]] = run;
    if (j == 0) child_start[n_runs] = (uint32_t)n;
  }
}
}  // namespace

void bear_window_free(bear_window_dev *wt) {
  if (!wt) return;
  (void)hipFree(wt->codes);
  (void)hipFree(wt->row_of_context);
  (void)hipFree(wt->perm);
  (void)hipFree(wt->child_start);
  (void)hipFree(wt->rows);
  wt->codes = nullptr;
  wt->row_of_context = wt->perm = wt->child_start = nullptr;
  wt->rows = nullptr;
  wt->n = 0;
  wt->bytes = 0;
}

int bear_window_build(const unsigned long long *codes, uint64_t n_rows, int pos, int fw, bear_window_dev *out, hipStream_t s) {
  if (!codes || !out || n_rows == 0 || n_rows > 0xfffffffeull || pos < 0 || fw < 1 || pos + fw > 21) return BEAR_ERR_INVALID_ARG;
  int st = BEAR_OK;
  const int shift = 3 * pos;
  const unsigned long long mask = (1ull << (3 * fw)) - 1ull;      // (fw <= 21: at most 63 bits)
  unsigned long long fill = 0ull;
  for (int l = 0; l < 21; ++l)
    if (l < pos || l >= pos + fw) fill |= 5ull << (3 * l);
  unsigned long long *keys_in = nullptr, *keys = nullptr;
  uint32_t *vals_in = nullptr, *flag = nullptr, *scan = nullptr, n_runs = 0;
  void *temp = nullptr;
  size_t tb = 0, tb2 = 0;
  out->pos = pos;
  out->n = 0;
  out->codes = nullptr;
  out->row_of_context = out->perm = out->child_start = nullptr;
  out->rows = nullptr;
  out->bytes = 0;
  CNT_TRY(hipMalloc(&keys_in, n_rows * 8));
  CNT_TRY(hipMalloc(&keys, n_rows * 8));
  CNT_TRY(hipMalloc(&vals_in, n_rows * 4));
  CNT_TRY(hipMalloc(&out->perm, n_rows * 4));
  CNT_TRY(hipMalloc(&out->row_of_context, n_rows * 4));
  hipLaunchKernelGGL(window_keys_kernel, dim3(grid_for(n_rows)), dim3(256), 0, s, codes, n_rows, shift, mask, keys_in, vals_in);
  CNT_TRY(hipGetLastError());
  CNT_TRY(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys, vals_in, out->perm, n_rows, 0u, (unsigned)(3 * fw), s));
  CNT_TRY(hipMalloc(&temp, tb ? tb : 8));
  CNT_TRY(rocprim::radix_sort_pairs(temp, tb, keys_in, keys, vals_in, out->perm, n_rows, 0u, (unsigned)(3 * fw), s));   // stable: ties keep row order
  CNT_TRY(hipStreamSynchronize(s));
  (void)hipFree(temp);
  temp = nullptr;
  (void)hipFree(keys_in);
  keys_in = nullptr;
  // runs of equal windows (the flag / scan of the prefix levels, on the sorted keys)
  flag = vals_in;          // (the unsorted row numbers are no longer needed)
  vals_in = nullptr;
  CNT_TRY(hipMalloc(&scan, n_rows * 4));
  hipLaunchKernelGGL(level_flag_kernel, dim3(grid_for(n_rows)), dim3(256), 0, s, keys, n_rows, ~0ull, flag);
  CNT_TRY(hipGetLastError());
  CNT_TRY(rocprim::inclusive_scan(nullptr, tb2, flag, scan, n_rows, rocprim::plus<uint32_t>(), s));
  CNT_TRY(hipMalloc(&temp, tb2 ? tb2 : 8));
  CNT_TRY(rocprim::inclusive_scan(temp, tb2, flag, scan, n_rows, rocprim::plus<uint32_t>(), s));
  CNT_TRY(hipMemcpyAsync(&n_runs, scan + (n_rows - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  CNT_TRY(hipStreamSynchronize(s));
  CNT_TRY(hipMalloc(&out->codes, (size_t)n_runs * 8));
  CNT_TRY(hipMalloc(&out->child_start, ((size_t)n_runs + 1) * 4));
  CNT_TRY(hipMalloc(&out->rows, (size_t)n_runs * 16 * sizeof(double)));
  hipLaunchKernelGGL(window_compact_kernel, dim3(grid_for(n_rows)), dim3(256), 0, s, keys, out->perm, n_rows, shift, fill, flag, scan, out->codes,
                     out->child_start, out->row_of_context, (uint64_t)n_runs);
  CNT_TRY(hipGetLastError());
  CNT_TRY(hipStreamSynchronize(s));
  out->n = n_runs;
  out->bytes = (uint64_t)n_runs * (8 + 4 + 128) + n_rows * 8;
done:
  if (temp) (void)hipFree(temp);
  if (keys_in) (void)hipFree(keys_in);
  if (keys) (void)hipFree(keys);
  if (vals_in) (void)hipFree(vals_in);
  if (flag) (void)hipFree(flag);
  if (scan) (void)hipFree(scan);
  if (st != BEAR_OK) bear_window_free(out);
  return st;
}

void bear_level_free(bear_level_dev *lv) {
  if (!lv) return;
  (void)hipFree(lv->codes);
  (void)hipFree(lv->parent_of_below);
  (void)hipFree(lv->child_start);
  (void)hipFree(lv->rows);
  lv->codes = nullptr;
  lv->parent_of_below = nullptr;
  lv->child_start = nullptr;
  lv->rows = nullptr;
  lv->n = 0;
  lv->bytes = 0;
}

int bear_level_build(const unsigned long long *codes_below, uint64_t n_below, int letters, bear_level_dev *out, hipStream_t s) {
  if (!codes_below || !out || n_below == 0 || n_below > 0xfffffffeull || letters < 1 || letters > 21) return BEAR_ERR_INVALID_ARG;
  int st = BEAR_OK;
  const unsigned long long mask = (1ull << (3 * letters)) - 1ull;
  unsigned long long fill = 0ull;
  for (int l = letters; l < 22 && 3 * l < 64; ++l) fill |= 5ull << (3 * l);     // (bear_pack_kmers_u64 fills positions >= lag the same way)
  uint32_t *flag = nullptr, *scan = nullptr, n_runs = 0;
  void *temp = nullptr;
  size_t tb = 0;
  out->n = 0;
  out->letters = letters;
  out->codes = nullptr;
  out->parent_of_below = nullptr;
  out->child_start = nullptr;
  out->rows = nullptr;
  out->bytes = 0;
  CNT_TRY(hipMalloc(&flag, n_below * sizeof(uint32_t)));
  CNT_TRY(hipMalloc(&scan, n_below * sizeof(uint32_t)));
  hipLaunchKernelGGL(level_flag_kernel, dim3(grid_for(n_below)), dim3(256), 0, s, codes_below, n_below, mask, flag);
  CNT_TRY(hipGetLastError());
  CNT_TRY(rocprim::inclusive_scan(nullptr, tb, flag, scan, n_below, rocprim::plus<uint32_t>(), s));
  CNT_TRY(hipMalloc(&temp, tb ? tb : 8));
  CNT_TRY(rocprim::inclusive_scan(temp, tb, flag, scan, n_below, rocprim::plus<uint32_t>(), s));
  CNT_TRY(hipMemcpyAsync(&n_runs, scan + (n_below - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  CNT_TRY(hipStreamSynchronize(s));
  CNT_TRY(hipMalloc(&out->codes, (size_t)n_runs * sizeof(unsigned long long)));
  CNT_TRY(hipMalloc(&out->child_start, ((size_t)n_runs + 1) * sizeof(uint32_t)));
  CNT_TRY(hipMalloc(&out->rows, (size_t)n_runs * 16 * sizeof(double)));
  hipLaunchKernelGGL(level_compact_kernel, dim3(grid_for(n_below)), dim3(256), 0, s, codes_below, n_below, mask, fill, flag, scan, out->codes,
                     out->child_start, (uint64_t)n_runs);
  CNT_TRY(hipGetLastError());
  CNT_TRY(hipStreamSynchronize(s));
  out->n = n_runs;
  out->parent_of_below = scan;
  scan = nullptr;
done:
  if (temp) (void)hipFree(temp);
  if (flag) (void)hipFree(flag);
  if (scan) (void)hipFree(scan);
  if (st != BEAR_OK) bear_level_free(out);
  return st;
}

// bear_common.h -- workspace, launch parameters and block-level reduction shared by all kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/bear_hip.h"
#include "bear_release_guard.h"
#include "bear_math.h"

#define BEAR_THREADS 256
#define BEAR_TILE_ROWS 1024
#define BEAR_ROWS_PER_THREAD (BEAR_TILE_ROWS / BEAR_THREADS)
#define BEAR_MAX_OUT 4

static thread_local int g_last_hip_error = 0;

#define HIP_TRY(expr)                      \
  do {                                     \
    hipError_t _e = (expr);                \
    if (_e != hipSuccess) {                \
      g_last_hip_error = (int)_e;          \
      return BEAR_ERR_HIP;                 \
    }                                      \
  } while (0)

struct bear_params;
struct bear_ws {
  int device;
  int num_cu;
  int max_blocks;
  double *partials;  // [max_blocks][BEAR_MAX_OUT]
  double *logtab;    // [BEAR_LOGTAB_N][2] = {r_i, -log r_i} (bear_math.h, bear_log_tab)
  unsigned long long *dbg;  // developer timing buffer [max_blocks][8][6]
  double *eval_partials;    // [eval_blocks][EVL_MAX_OUT] (kernels_eval.h)
  double *eval_out;         // [EVL_MAX_OUT] scratch result vector (bear_bmm_f64)
  int eval_blocks;
  double *lin_partials;     // [num_cu][LIN_MAX_GRAD] d/d mat partials (kernels_linrows.h)
  double *lin_accum;        // [LIN_MAX_GRAD] d/d mat accumulator of the fused linear step (kernels_linear.h): zero between launches
  unsigned long long *arrive;      // arrival word of the launch that owns `partials` (bear_arrival: epoch << 24 | blocks arrived)
  unsigned epoch;                  // host side: the stamp of the last launch that used `arrive` (never 0)
  double *cnn_partials;     // [cnn_blocks][cnn total] parameter-gradient partials (kernels_cnn.h), grown on demand
  size_t cnn_partials_cap;  // doubles
};

struct bear_params {
  double inv_h;   // 1 / exp(h_signed)
  double eps;
  // mode R (bear_ref.py:63-68 with the stop net function)
  double E;       // exp(-tau)
  double tauE;    // tau * exp(-tau)
  double tau;     // exp(tau_signed)
  double V;       // 1 / (nw + 1)
  double nw;      // exp(net_weight_signed)
};

// Partials leave the blocks that store them as AGENT-scope relaxed atomic stores (sc1: visible device-wide once acknowledged)
// -- not behind __threadfence(): an agent-scope release fence writes back the XCD's whole L2, and one per wave of a 2048-block
// grid cost 0.25 ms per launch (measured: dm_ref_items_kernel 0.135 -> 0.39 ms).  Only the last block pays one acquire.
__device__ __forceinline__ void bear_store_agent(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ 5-wide fp64 rows out of a wave
// One row per lane (rows i0 .. i0 + valid of a [n, 5] table, i0 a multiple of 64): a lane's own 40 bytes are five partial-line
// stores; through R (64 * 5 doubles of this wave's LDS) the wave's valid * 40 bytes leave as consecutive 16-byte pieces
// (i0 * 40 is a multiple of 16 when dst is 16-byte aligned); an odd count leaves one double.
__device__ __forceinline__ void bear_wave_store_rows5(double *R, const double (&f)[5], double *__restrict__ dst, uint64_t i0,
                                                      uint32_t valid, uint32_t lane) {
#pragma unroll
  for (int b = 0; b < 5; ++b) R[lane * 5u + b] = f[b];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t n16 = (valid * 5u) >> 1;
  typedef double bear_v2d __attribute__((ext_vector_type(2)));
  bear_v2d *out = reinterpret_cast<bear_v2d *>(dst + i0 * 5u);
  const bear_v2d *src = reinterpret_cast<const bear_v2d *>(R);
  // nontemporal: the rows are not read again by this kernel (linear rows forward: 0.963 -> 0.914 ms per 1e8 contexts)
  for (uint32_t k = lane; k < n16; k += 64u) __builtin_nontemporal_store(src[k], out + k);
  if ((valid & 1u) && lane == 0) dst[(i0 + valid) * 5u - 1u] = R[valid * 5u - 1u];
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();   // the reads are done before the wave's next rows land in R
}

// The reverse: rows i0 .. i0 + valid of a [n, 5] table into one row per lane, as consecutive 16-byte loads through R (a lane's own
// 40 bytes would be five 8-byte loads at a 40-byte stride: every line is touched by several instructions and part of it is
// fetched again).  Lanes beyond `valid` get the last row.  src 16-byte aligned.
__device__ __forceinline__ void bear_wave_load_rows5(double *R, const double *__restrict__ src, uint64_t i0, uint32_t valid,
                                                     uint32_t lane, double (&f)[5]) {
  typedef double bear_v2d __attribute__((ext_vector_type(2)));
  const uint32_t n16 = (valid * 5u) >> 1;
  const bear_v2d *in = reinterpret_cast<const bear_v2d *>(src + i0 * 5u);
  bear_v2d *dst = reinterpret_cast<bear_v2d *>(R);
  for (uint32_t k = lane; k < n16; k += 64u) dst[k] = __builtin_nontemporal_load(in + k);
  if ((valid & 1u) && lane == 0) R[valid * 5u - 1u] = src[(i0 + valid) * 5u - 1u];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t row = lane < valid ? lane : valid - 1u;
#pragma unroll
  for (int b = 0; b < 5; ++b) f[b] = R[row * 5u + b];
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();   // the reads are done before R is reused
}

// ------------------------------------------------------------------ block reduction
// Sums NOUT per-thread accumulators over the block (any block size that is a multiple of 64, up to
// 1024) and stores one partial per block.
template <int NOUT, bool AGENT = false>
__device__ __forceinline__ void block_store_partials(double (&acc)[NOUT], double *partials) {
  __shared__ double red[16][BEAR_MAX_OUT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    double v = bear_wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NOUT) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w][threadIdx.x];
    if (AGENT) bear_store_agent(&partials[(size_t)blockIdx.x * BEAR_MAX_OUT + threadIdx.x], s);
    else partials[(size_t)blockIdx.x * BEAR_MAX_OUT + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------ one launch per reduce
// How a launch of a planned kernel gets its constants and leaves its sums.  A training step on a small shard is bound by its
// launches, not by its kernels (scripts/dev/step_latency.py), so neither the constants nor the final sum take a launch of
// their own: every block derives the constants from the device-resident parameters in its prologue (a few exponentials), and
// the LAST block to finish sums the per-block partials -- in the fixed order of finalize_kernel, whichever block that is.
#define BEAR_THETA_NET 1   // theta = {h_signed, ...}                      (bear_net.py:43)
#define BEAR_THETA_REF 2   // theta = {h_signed, tau_signed, net_weight_signed}  (bear_ref.py:45-47, 106)
// The arrival word of a workspace and the stamp of THIS launch.  The word is `epoch << 24 | blocks arrived`, zero between
// launches; every launch is handed the next epoch by the host (ws_arrival, bear_hip.hip).  A block first raises the word to its
// own stamp (atomic max: a count left behind by an EARLIER launch -- one that faulted half way, or that overlapped on this
// workspace -- is discarded there instead of silently breaking the last-block detection of every later launch), then counts
// itself in: two one-way atomics per block, no retry loop (a compare-and-swap loop over 256 blocks that finish together cost
// 0.19 ms per launch).  A launch replayed from a HIP graph keeps the stamp it was captured with: between launches the word is
// back at zero (left there by the last block), so an older stamp counts from zero like any other.
// LIMIT of the self-healing: it works for EAGER launches only (each carries a newer stamp than anything left behind).  A graph
// replay carries the stamp of its capture: if an eager launch with a NEWER stamp has left a partial count behind (it faulted, or
// overlapped on this workspace against the header's rule), later replays never see their last block and io.out keeps stale sums;
// the host-side counter also wraps after 2^32 launches.  After a failed launch or synchronisation a caller therefore destroys
// the workspace (bear_ws_destroy / bear_ws_create zero the words) and captures its graph again -- the Python loop does not
// survive a BearError either way (it propagates).
struct bear_arrival {
  unsigned long long *word;
  unsigned epoch;
};
struct bear_step_io {
  const double *theta;   // non-NULL: constants from these parameters (kind), else the by-value bear_params of the launch
  int kind;
  unsigned epoch;        // this launch's stamp on the arrival word (in the padding behind `kind`: the argument block keeps its size)
  double *out;           // non-NULL: the last block writes the fixed-order sums here; NULL: a finalize_kernel launch follows
  unsigned long long *arrive_word;   // bear_ws::arrive
  __host__ __device__ bear_arrival arrive() const { return bear_arrival{arrive_word, epoch}; }
};
static_assert(sizeof(bear_step_io) == 32, "bear_step_io: kernel-argument size");

__device__ __forceinline__ double bear_uniform_f64(double v) {   // a wave-uniform value back into scalar registers
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// What bear_dm_ref_plan_f64 / bear_dm_prior_plan_f64 derive on the host, from parameters the optimizer moves on the device.
__device__ __forceinline__ bear_params bear_params_of(const bear_params &arg, const bear_step_io &io) {
  if (!io.theta) return arg;
  bear_params p;
  p.eps = arg.eps;
  p.inv_h = bear_uniform_f64(1.0 / exp(io.theta[0]));
  p.E = p.tauE = p.tau = p.V = p.nw = 0.0;
  if (io.kind == BEAR_THETA_REF) {
    const double tau = exp(io.theta[1]), nw = exp(io.theta[2]);
    const double E = exp(-tau);
    p.E = bear_uniform_f64(E);
    p.tauE = bear_uniform_f64(tau * E);
    p.tau = bear_uniform_f64(tau);
    p.V = bear_uniform_f64(1.0 / (nw + 1.0));
    p.nw = bear_uniform_f64(nw);
  }
  return p;
}

// After its partials are stored (bear_store_agent) a block announces itself; true for the block that arrived last.  Every
// thread waits for the acknowledgement of its own stores (s_waitcnt vmcnt(0): gfx9 counts stores there; a workgroup-scope
// release fence compiles to nothing on gfx950) before the barrier, so the arrival counter is bumped only after all of this
// block's partials are visible to the device.
// Two levels: a block arrives at one of BEAR_ARRIVE_SUBS counters (its number mod 16, each on a cache line of its own), the last one
// of a counter at the top word.  Atomics on ONE address serialise (~12 ns each): 2048 blocks that finish together spent 25 us there
// with one counter -- half the duration of a launch-bound step (dm_ref_items_kernel at 1e7 contexts: 59 us, 15 of them work).
#define BEAR_ARRIVE_SUBS 16
#define BEAR_ARRIVE_STRIDE 16            // words between counters (128 bytes)
#define BEAR_ARRIVE_WORDS ((BEAR_ARRIVE_SUBS + 1) * BEAR_ARRIVE_STRIDE)
__device__ __forceinline__ bool bear_arrive_count(unsigned long long *word, unsigned long long tag, unsigned expected) {
  __hip_atomic_fetch_max(word, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag + (expected - 1u);
}
__device__ __forceinline__ bool bear_arrive_last(const bear_arrival &arrive) {
  __shared__ unsigned s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long tag = (unsigned long long)arrive.epoch << 24;
    const unsigned sub = blockIdx.x % BEAR_ARRIVE_SUBS, grid = gridDim.x;
    const unsigned in_sub = (grid - sub + BEAR_ARRIVE_SUBS - 1u) / BEAR_ARRIVE_SUBS;      // blocks with this residue
    const unsigned subs = grid < BEAR_ARRIVE_SUBS ? grid : BEAR_ARRIVE_SUBS;               // counters in use
    bool last = bear_arrive_count(arrive.word + (1u + sub) * BEAR_ARRIVE_STRIDE, tag, in_sub);
    if (last) last = bear_arrive_count(arrive.word, tag, subs);
    s_last = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return false;
  // the one block that goes on to read: a single agent-scope acquire (invalidates this CU's L1 and the XCD's L2 lines that may
  // still hold the previous launch's partials), then plain, pipelined loads
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

// The last block is done with the partials: the words go back to zero (a replay of this launch starts its count afresh).
__device__ __forceinline__ void bear_arrive_reset(const bear_arrival &arrive) {
#pragma unroll
  for (int k = 0; k <= BEAR_ARRIVE_SUBS; ++k)
    __hip_atomic_store(arrive.word + k * BEAR_ARRIVE_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The arithmetic of finalize_kernel (256 threads, same order) inside the last block of the producing launch.
__device__ __forceinline__ void bear_finalize_in_block(const double *partials, int n_out, double *out, const bear_arrival &arrive, bool accumulate = false) {
  __shared__ double fred[4][BEAR_MAX_OUT];
  const int n_blocks = (int)gridDim.x;
  if (threadIdx.x < 256) {
    double acc[BEAR_MAX_OUT] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < n_blocks; b += 256)
#pragma unroll
      for (int k = 0; k < BEAR_MAX_OUT; ++k)
        if (k < n_out) acc[k] += partials[(size_t)b * BEAR_MAX_OUT + k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < BEAR_MAX_OUT; ++k) {
      double v = bear_wave_sum(acc[k]);
      if (lane == 0) fred[wave][k] = v;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_out) {
    const double v = (fred[0][threadIdx.x] + fred[1][threadIdx.x]) + (fred[2][threadIdx.x] + fred[3][threadIdx.x]);
    out[threadIdx.x] = accumulate ? out[threadIdx.x] + v : v;    // (accumulate: the second launch of a step that takes two)
  }
  if (threadIdx.x == 0) bear_arrive_reset(arrive);
}

// ---- the optimizer update (tf.keras Adam, bear_net.py:278-282) of a step whose sums need no all-reduce, INSIDE the launch that
// formed them: a launch-bound step (configs[1]: ~15 us of kernel) is then ONE launch instead of two.  The last block -- every other
// block has arrived, i.e. is done reading theta -- runs the update on `packed` = io.out = [sum LL, d/d theta...] it has just written.
// theta == NULL: no update (the two-launch form: reduce, [all-reduce], bear_train_apply_f64).
struct bear_apply_io {
  double *theta;            // [n_theta], updated in place
  double *m, *v;            // Adam moments [n_theta]
  double *t_state;          // [1]: steps taken so far
  double *loss_buf;         // [loss_cap] or NULL: loss_buf[step] = -scale * packed[0]
  unsigned long long loss_cap;
  double lr, scale;         // gradients are scale * packed[1 + k]
  int n_theta, train_ar;    // train_ar: theta[0] = h_signed gets no update (bear_net.py:194-196)
};

// tf.keras Adam on theta[k], k = tid, tid + n_threads, ... -- the body of adam_vec_kernel and of the fused step alike (one source:
// the two forms end in the same bits).  Every thread reads the step counter before the barrier, thread 0 advances it behind it.
__device__ __forceinline__ void bear_adam_update(const bear_apply_io &A, const double *packed, int tid, int n_threads) {
  const double t0 = A.t_state[0], t = t0 + 1.0;
  const double b1 = 0.9, b2 = 0.999, aeps = 1e-7;
  const double lr_t = A.lr * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t));
  for (int k = tid; k < A.n_theta; k += n_threads) {
    if (A.train_ar && k == 0) continue;
    const double g = A.scale * packed[1 + k];
    const double mk = b1 * A.m[k] + (1.0 - b1) * g, vk = b2 * A.v[k] + (1.0 - b2) * g * g;
    A.m[k] = mk;
    A.v[k] = vk;
    A.theta[k] -= lr_t * mk / (sqrt(vk) + aeps);
  }
  if (tid == 0) {
    const unsigned long long step = (unsigned long long)t0;
    if (A.loss_buf && step < A.loss_cap) A.loss_buf[step] = -A.scale * packed[0];
  }
  __syncthreads();
  if (tid == 0) A.t_state[0] = t;
}

// ... by the last block of a reduce launch, behind its fixed-order sums (block-uniform call)
__device__ __forceinline__ void bear_apply_in_block(const bear_apply_io &A, const double *packed) {
  if (!A.theta) return;
  __threadfence_block();
  __syncthreads();            // packed[] was written by other threads of this block
  bear_adam_update(A, packed, (int)threadIdx.x, (int)blockDim.x);
}

template <int NOUT>
__device__ __forceinline__ void block_finish(double (&acc)[NOUT], double *partials, const bear_step_io &io) {
  if (!io.out) {
    block_store_partials<NOUT>(acc, partials);
    return;
  }
  block_store_partials<NOUT, true>(acc, partials);
  if (bear_arrive_last(io.arrive())) bear_finalize_in_block(partials, NOUT, io.out, io.arrive());
}
template <int NOUT>
__device__ __forceinline__ void block_finish(double (&acc)[NOUT], double *partials, const bear_step_io &io, const bear_apply_io &apply) {
  if (!io.out) {                                        // (a finalize_kernel launch follows: never together with an update)
    block_store_partials<NOUT>(acc, partials);
    return;
  }
  block_store_partials<NOUT, true>(acc, partials);
  if (bear_arrive_last(io.arrive())) {
    bear_finalize_in_block(partials, NOUT, io.out, io.arrive());
    bear_apply_in_block(apply, io.out);
  }
}

// ------------------------------------------------------------------ finalize: fixed-order sum of block partials
__global__ __launch_bounds__(256) void finalize_kernel(const double *__restrict__ partials, int n_blocks,
                                                       int n_out, double *__restrict__ out) {
  __shared__ double red[4][BEAR_MAX_OUT];
  double acc[BEAR_MAX_OUT] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < n_blocks; b += 256)
#pragma unroll
    for (int k = 0; k < BEAR_MAX_OUT; ++k)
      if (k < n_out) acc[k] += partials[(size_t)b * BEAR_MAX_OUT + k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < BEAR_MAX_OUT; ++k) {
    double v = bear_wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < n_out) out[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}


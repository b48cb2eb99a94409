// kernels_linear.h -- the whole bear_net training step for the linear AR function, fused on the plan.
//
//   f_i = softmax(sum_l mat[l, kmer_i[l], :])                      (ar_funcs.py:41-45)
//   sum LL, d sum LL / d h_signed                                   (bear_net.py:177-191, core.py:73-74)
//   d sum LL / d mat[l, a, b]                                       (bear_net.py:193 through the softmax)
//
// Contexts arrive as one 64-bit word each (bear_linear_index_u64: the k-mer as row numbers of the letter-group tables below),
// 8 bytes per context instead of the 40-byte prior row, and nothing is written per context.  Per tile of the plan, two barriers:
//   A  one thread per LIVE context (entries tid, tid + 896 of the plan's per-tile list of contexts that hold counts; ~30% of
//      the rows of a count table hold none in the training column and are never looked at): logits from GROUP tables -- pairs
//      of letters T[g][a, a'] = mat[2g][a] + mat[2g+1][a'] over the leading letters and one triple over the last three, rows of
//      four doubles relative to the fifth letter's logit and in units of ln2/128 (6 table rows of 32 bytes for lag 13, read
//      back to back: the group count is a template parameter) -- softmax; the row goes to LDS for the items AND stays in the
//      thread's registers;
//   B  ticketed item units as in dm_prior_plan_kernel: D, P per item; with q = dLL/df at the item's cell the item leaves
//      -w = -f q in its own cell of the LDS row (a cell belongs to at most one item; f > 0, so the sign marks it);
//   C  the same thread as in A reads its rows' cells back: w_b from the marked cells, s = sum_b w_b, and the softmax backward
//      g_b = w_b - f_b s with f from its registers (the common base -u P(A,n) of the five cells drops out because the row
//      sums to one).  d/d mat[l][a][:] = sum over contexts with letter a at position l of g: the 64 contexts of a wave are
//      consecutive live rows, and in a k-mer-sorted table (bear_net.train sorts a batch's rows by k-mer at upload: the sums do
//      not depend on the order) they share all but the last few letters.  What an LDS fp64 atomic costs is its INSTRUCTION
//      (~20 clocks of CU time whatever the number of active lanes, scripts/dev/lds_atomic_lanes.hip), so lanes are mapped to
//      (group, letter) pairs: one instruction adds the wave's sums (DPP reductions) to every group the wave shares, one adds
//      the row-of-16 sums to the groups a row of 16 shares, and only what is left (sorted: the triple) takes one add per
//      context and letter, into letter-major tables so that the 64 adds of an instruction spread over all banks.  (Round 1:
//      28 + 8 atomic instructions per 64 contexts = 2.2 of its 3.85 ms; now 6.)
//   C of tile t and A of tile t + 1 are one phase (a thread only touches its own row slots).
// The context terms -D(A, n) come from the plan's histogram (A = u + 5 eps: softmax rows are normalised).
// After the last tile the group tables fold into d/d mat partials; the last block to finish sums the blocks in fixed order.
// Note: LDS floating-point atomics make the summation order inside a block run-dependent (last-bit jitter in
// grad_mat); the ELBO and d/dh sums keep the fixed-order reduction of the other kernels.
#pragma once
#include "kernels_plan.h"
#include <type_traits>
#ifndef LIN_MAX_LAG
#define LIN_MAX_LAG 21
#endif
#ifndef LIN_CH
#define LIN_CH 4
#endif
// Letter groups: pairs over the leading letters, ONE triple over the last three.  In a k-mer-sorted table the last letters
// vary fastest: with them in a single group, the 64 consecutive contexts of a wave differ in that group only (plus, now and
// then, the pair before it), so every other group takes one block-level add (see phase C).
// Table rows hold four doubles: the logits relative to the fifth letter's (a softmax does not see the shift), 32-byte aligned.
#define LIN_PAIR_COMBOS 36
#define LIN_TRI_COMBOS 216
#define LIN_PSTRIDE (LIN_PAIR_COMBOS * 4)
#define LIN_MAX_PAIRS ((LIN_MAX_LAG - 3 + 1) / 2)
#define LIN_MAX_GROUPS (LIN_MAX_PAIRS + 1)
#define LIN_TAB_DOUBLES (LIN_MAX_PAIRS * LIN_PSTRIDE + LIN_TRI_COMBOS * 4)
#define LIN_GT_PLANE (LIN_TAB_DOUBLES / 4)   // rows of all group tables
#define LIN_MAX_GRAD (LIN_MAX_LAG * 25)

// The PAIRED form of a tile's list (plan_pair_kernel, built once per batch from the plan's list and the index words): the same
// rows in the same order, with an empty entry (0xffff) in front of every run of equal leading letters (all pair groups) that
// would otherwise start at an odd position -- so entries 2 j and 2 j + 1 always share every pair group, and thread j takes both:
// the pair rows are read and multiplied once for two contexts (phase A), and one reduction / run detection / wave- and row-level
// add serves 128 contexts instead of 64 (phase C).  A tile whose list would grow beyond what the row threads take (2 per thread: a
// stretch of the table where a run is one context) keeps its plain list: the step is then two launches, the paired form of the
// kernel over the paired tiles and the plain form over the rest (both forms in one kernel ran out of registers).
#ifndef LIN_DMA_WAVES
#define LIN_DMA_WAVES 2
#endif
#ifndef LIN_ROW_THREADS
#define LIN_ROW_THREADS (PLN_THREADS - 64 * LIN_DMA_WAVES)     // (see "The waves that issue a tile's DMA" below)
#endif
#define LIN_PAIR_CAP (2 * LIN_ROW_THREADS)                  // entries the row threads of a block take (LIN_RPT * LIN_ROW_THREADS, asserted below)
// Behind a tile's entries the list carries one LEVEL word per pair, padded to whole units of 64 pairs: how many LEADING pair groups
// the pair's unit of 64 / its row of 16 / its quad share (bits 0-3 / 4-7 / 8-11).  They decide which of phase C's adds a lane makes
// and depend on the k-mers alone, so the builder works them out once per batch instead of every wave of every launch (~35
// vector instructions of cross-lane ORs and bit scans per 128 contexts: round 6).
#define LIN_LEV_CAP (((LIN_PAIR_CAP / 2) + 63) / 64 * 64)
#define LIN_LIVE2_STRIDE ((2 + LIN_PAIR_CAP + LIN_LEV_CAP + 7) / 8 * 8)   // uint16 per tile of the paired lists (a multiple of 8: 16-byte rows)
__host__ __device__ inline uint32_t lin_lev_len(uint32_t n_ent) { return ((n_ent >> 1) + 63u) & ~63u; }      // level words behind n_ent entries
#define LIN_EMPTY 0xffffu

struct lin_buf {
  __attribute__((aligned(16))) unsigned long long codes[PLN_RMAX + 2];
  __attribute__((aligned(16))) unsigned char blk[PLN_BLOCK_MAX];
  __attribute__((aligned(16))) uint16_t live[LIN_LIVE2_STRIDE];  // the plan's list of contexts with counts: [0] = how many, then rows
                                                                 // (paired form, bear_plan_pair_contexts: [0] = entries, [1] unused, then entries)
};
struct pln_lds_lin {
  double pri[PLN_RMAX * 5 + 2];  // [PLN_SENTINEL] = 1.0
  lin_buf buf[2];
  __attribute__((aligned(32))) double T[LIN_TAB_DOUBLES];   // [pair g][a * 6 + a'][b < 4] ... | [triple][(a * 6 + a') * 6 + a''][b < 4]
  double GT[LIN_TAB_DOUBLES];   // gradient tables, letter-major: [b < 4][row]: the 64 adds of a wave (one letter, 64 rows) spread over all banks
  double2 logtab[BEAR_LOGTAB_N];
  double tabD[SRT_NKEY];
  double tabP[SRT_NKEY];
  double exptab[BEAR_EXPTAB_N];
  __attribute__((aligned(16))) pln_tile desc[4];   // tile descriptors, landed by LDS-DMA two iterations ahead of their use (see the tile loop)
  uint32_t ticket[2];
  uint32_t c_done;               // += 1 by every wave that has read its rows back in phase C (see the tile loop)
  unsigned long long t_max;      // bits of the largest |logit| in the group tables (table build)
};
static_assert(sizeof(pln_lds_lin) <= 160 * 1024, "linear-head kernel: LDS budget");
static_assert(PLN_RMAX * 5 >= LIN_MAX_LAG * 25 + 75 * 9, "pln_lds_lin::pri doubles as the scratch of the table build and the fold");

// group geometry of a lag: npair pair groups over the letters before the triple (the last pair holds one letter when their
// number is odd), then the triple over letters [tri, tri + 3) (positions >= lag hold the "unknown" letter: they add nothing)
struct lin_geom {
  int lag, npair, tri, ng;   // ng = npair + 1
  uint32_t last_pair_mask;   // 7: the last pair holds one letter; 63: two
};
__host__ __device__ inline lin_geom lin_make_geom(int lag) {
  lin_geom G;
  G.lag = lag;
  G.tri = lag >= 3 ? lag - 3 : 0;
  G.npair = (G.tri + 1) / 2;
  G.ng = G.npair + 1;
  G.last_pair_mask = (G.tri & 1) ? 7u : 63u;
  return G;
}

// int8 codes [n, lag] (core.encode_kmers: 0..A letters / start symbol, anything else unknown) -> packed words
__global__ void pack_kmers_kernel(const int8_t *__restrict__ codes, uint64_t n, int lag, unsigned long long *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long w = 0;
  for (int l = 0; l < LIN_MAX_LAG + 1; ++l) {
    unsigned long long v = 5;
    if (l < lag) {
      const int c = codes[i * lag + l];
      v = (c >= 0 && c <= 4) ? (unsigned long long)c : 5ull;
    }
    w |= v << (3 * l);
  }
  out[i] = w;
}

// ASCII k-mers as they sit in the count file -> int8 letter codes (core.tf_one_hot's alphabet order, core.py:146-153:
// 0..3 letters, 4 = start symbol '[', -1 = anything else) -- the host LUT of a 1e9-row table took longer than an epoch.
__global__ void encode_kmers_kernel(const uint8_t *__restrict__ ascii, uint64_t n_bytes, int rna, int8_t *__restrict__ codes) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t ch = ascii[i];
    int8_t c = -1;
    if (ch == 'A') c = 0;
    else if (ch == 'C') c = 1;
    else if (ch == 'G') c = 2;
    else if (ch == (rna ? 'U' : 'T')) c = 3;
    else if (ch == '[') c = 4;
    codes[i] = c;
  }
}

// ---- group indices of a context.  The linear head does not read packed letters (3 bits each, bear_pack_kmers_u64) but one
// word of table-row indices per context, built once per batch by bear_linear_index_u64: pair g at bits [6 g, 6 g + 6)
// (a * 6 + a' < 36), the triple behind the pairs at bits [6 npair, 6 npair + 8) (< 216).  In the kernel the number of groups
// NG = npair + 1 is a template parameter: every shift is a constant and the table reads of a context issue back to back.
__host__ __device__ inline unsigned long long lin_index_word(unsigned long long raw, const lin_geom &G) {
  unsigned long long cv = 0ull;
  for (int g = 0; g < G.npair; ++g) {
    uint32_t field = (uint32_t)(raw >> (6 * g)) & 63u;
    if (g == G.npair - 1) field &= G.last_pair_mask;   // a pair of one letter: its partner belongs to the triple
    cv |= (unsigned long long)((field & 7u) * 6u + (field >> 3)) << (6 * g);
  }
  const uint32_t f9 = (uint32_t)(raw >> (3 * G.tri)) & 511u;
  cv |= (unsigned long long)((f9 & 7u) * 36u + ((f9 >> 3) & 7u) * 6u + (f9 >> 6)) << (6 * G.npair);
  return cv;
}
__global__ void linear_index_kernel(const unsigned long long *__restrict__ raw, uint64_t n, int lag, unsigned long long *__restrict__ out) {
  const lin_geom G = lin_make_geom(lag);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = lin_index_word(raw[i], G);
}
// ---- bear_plan_pair_contexts: the paired lists of every tile (see LIN_PAIR_CAP).  One thread per tile walks the tile's list (set-up
// path: once per batch); the number of entries goes to the list's first word and to n_ent[tile] (0xffff: the list does not fit, the
// tile keeps its plain list -- the host sorts the tiles into two descriptor arrays, one per form of the kernel).
//
// WHICH contexts of a run of equal leading letters (a "block": all pair groups equal) share a lane, and in which order the lanes
// follow each other inside the block, is free -- every sum is the same -- and decides what the triple's gradient adds cost: an LDS
// fp64 atomic of 64 lanes is carried out in four passes of 16 lanes, and a pass takes as many turns as its fullest bank pair holds
// lanes (measured with timing-only builds, profiles/NOTES_r05.md: rows = lane % 16 cost what 64 distinct rows cost, rows = lane / 2
// cost 0.1 ms more per 1e8 contexts).  The bank pair of a context's add is its triple row mod 16, so the builder deals a block's
// contexts to the (16 lanes x slot) groups greedily: per lane and slot the context whose class is not yet in the group and has the
// most contexts left in the block; when every class left is taken, the second slot stays empty (while the list has room) or the
// fullest class goes anyway.  Two copies of one k-mer (the benchmark table draws its k-mers with replacement: 28 % of them occur
// twice or more) take ONE lane: the kernel adds the sum of their gradients to the triple row once (lin_scatter_grad_paired).
// Ties go to the context whose ROW class (row mod 16: the bank pair of its softmax row in phases A and
// C) is new to the group.  Measured on the 1e8 table, same box: 1.479 -> 1.455 ms from the order alone, 1.447 with the empty slots,
// 1.432 with the row classes.
// One wave per tile: the lanes fetch the tile's list and what the dealing needs of each context's index word into LDS (a byte: class,
// first of its block, copy of its predecessor); then LANE q KEEPS CLASS q (q < 16: its bucket of the block, how many contexts it has
// left, its next context's row class and whether that context has a copy behind it) and a pick is three 16-lane maxima instead
// of a scan over the classes; the lanes write the list out.  (A thread per tile out of global memory was latency-bound: 43 ms per
// 1e8 contexts and 13 ms for any small batch; one lane per tile out of LDS 54 ms: ~400 dependent instructions per pick.)
__device__ __forceinline__ uint32_t lin_pair_levels(unsigned long long cv, uint32_t lane, uint32_t np);      // (below, next to its consumer)
__device__ __forceinline__ uint32_t lin_max16(uint32_t v) {       // the maximum over lanes 0..15 (the other lanes hold 0), in every one of them
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__global__ __launch_bounds__(64) void plan_pair_kernel(const pln_tile *__restrict__ tiles, uint64_t n_tiles, const uint16_t *__restrict__ live,
                                                       const unsigned long long *__restrict__ kmer_index, int lag, uint16_t *__restrict__ live2,
                                                       uint16_t *__restrict__ n_ent, int empty_slots) {
  const lin_geom G = lin_make_geom(lag);
  const unsigned long long pair_mask = (1ull << (6 * G.npair)) - 1ull;      // every pair group of the index word (bear_linear_index_u64)
  const int tri_shift = 6 * G.npair;
  __shared__ uint16_t row_l[PLN_RMAX];        // the tile's list: rows of the contexts with counts, in sorted order
  __shared__ uint8_t meta[PLN_RMAX + 1];      // per list position: class (4 bits) | 16: first of its block | 32: same k-mer as its predecessor
  __shared__ uint16_t bucket[PLN_RMAX];       // a block's list positions bucketed by class
  __shared__ uint16_t out_l[LIN_LIVE2_STRIDE];
  __shared__ uint32_t cnt_s[16];
  const uint32_t lane = threadIdx.x, q = lane & 15u;
  const bool keeper = lane < 16u;
  for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint16_t *in = live + t * PLN_LIVE_STRIDE;
    const uint64_t row0 = tiles[t].row0;
    const uint32_t n = in[0];
    for (uint32_t i = lane; i < n; i += 64) {
      const uint32_t row = in[1 + i];
      const unsigned long long w = kmer_index[row0 + row];
      const unsigned long long wp = i ? kmer_index[row0 + in[i]] : ~w;
      row_l[i] = (uint16_t)row;
      meta[i] = (uint8_t)(((uint32_t)(w >> tri_shift) & 15u) | (((w ^ wp) & pair_mask) != 0ull || i == 0u ? 16u : 0u) | (w == wp ? 32u : 0u));
    }
    if (lane == 0) meta[n] = 16u;             // (a block ends where the next one starts)
    __syncthreads();
    // everything below is wave-uniform control flow; m, used*, need, left are the same numbers in every lane
    uint32_t m = 0;                       // entries written (or that would have been: the list is dropped when m > LIN_PAIR_CAP)
    uint32_t used[2] = {0u, 0u};          // classes present in the current group of 16 lanes, per slot
    uint32_t used_row[2] = {0u, 0u};      // ... and the classes of the contexts' ROWS (row mod 16: the bank pair of the lane's softmax row in
                                          // phases A and C, five doubles per row and 5 is odd)
    // entries the blocks need at least (a block's contexts, rounded up to whole lanes): empty slots beyond that only while the list
    // is sure to fit without them too
    uint32_t need = 0;
    for (uint32_t s = 0; s < n;) {
      uint32_t e = s + 1u;
      while (!(meta[e] & 16u)) ++e;
      need += (e - s + 1u) & ~1u;
      s = e;
    }
    auto put = [&](uint32_t v) {           // (called by ONE lane)
      if (m < LIN_PAIR_CAP) out_l[2 + m] = (uint16_t)v;
    };
    for (uint32_t s = 0; s < n;) {
      // the block [s, e): contexts with the leading letters of entry s; my class's share of it
      uint32_t e = s + 1u;
      while (!(meta[e] & 16u)) ++e;
      uint32_t cnt = 0;
      if (keeper)
        for (uint32_t i = s; i < e; ++i) cnt += (meta[i] & 15u) == q ? 1u : 0u;
      if (keeper) cnt_s[q] = cnt;
      __syncthreads();
      uint32_t end = 0;                    // my bucket is [end - cnt, end) once it is filled
      for (uint32_t k = 0; k < 16u; ++k) end += k <= q ? cnt_s[k] : 0u;
      if (keeper) {
        uint32_t at = end - cnt;
        for (uint32_t i = s; i < e; ++i)
          if ((meta[i] & 15u) == q) bucket[at++] = (uint16_t)i;       // (order inside a class: as listed)
      }
      __syncthreads();
      // my class's next context: its row class, and whether the one behind it in the bucket is the SAME k-mer (the positions of a
      // class keep their sorted order, so copies of a k-mer are neighbours in the list and in the bucket): such a pair shares a
      // lane and the kernel adds the sum of the two gradients to the triple row once
      uint32_t head_rc = 0u;
      bool head_dbl = false;
      auto refresh = [&]() {
        head_rc = 0u;
        head_dbl = false;
        if (!keeper || cnt == 0u) return;
        const uint32_t a0 = bucket[end - cnt];
        head_rc = (uint32_t)row_l[a0] & 15u;
        if (cnt >= 2u) {
          const uint32_t a1 = bucket[end - cnt + 1u];
          head_dbl = a1 == a0 + 1u && (meta[a1] & 32u);
        }
      };
      refresh();
      uint32_t left = e - s;
      need -= (left + 1u) & ~1u;                               // (of the blocks behind this one)
      while (left) {
        if ((m & 31u) == 0u) used[0] = used[1] = used_row[0] = used_row[1] = 0u;         // a new group of 16 lanes
        for (int slot = 0; slot < 2; ++slot) {
          if (!left) {
            if (lane == 0) put(LIN_EMPTY);                     // (only ever behind a first slot: m is odd here)
            ++m;
            continue;
          }
          // of the classes not in the group yet the one with the most contexts left -- among those whose next context also brings a
          // new row class, if there is one (score: 2 x count + 1); the second slot leaves the doubles to the first slots while it can;
          // equal scores: the lowest class
          const bool fresh = keeper && cnt != 0u && !((used[slot] >> q) & 1u);
          // (a third criterion -- class mod 8 new to the current EIGHT lanes, for the 16-byte table reads of phase A -- changed nothing)
          const uint32_t score = fresh ? (2u * cnt + (((used_row[slot] >> head_rc) & 1u) ? 0u : 1u)) * 16u + (15u - q) : 0u;
          const bool dbl2 = slot == 1 && head_dbl;
          uint32_t key = lin_max16(dbl2 ? 0u : score);
          if (key == 0u) key = lin_max16(dbl2 ? score : 0u);
          if (key == 0u) {
            // every class left is in this group already: an empty second slot while the tile's list keeps room for what is left
            if (empty_slots && slot == 1 && m + 1u + ((left + 1u) & ~1u) + need <= LIN_PAIR_CAP) {
              if (lane == 0) put(LIN_EMPTY);
              ++m;
              continue;
            }
            key = lin_max16(keeper && cnt != 0u ? cnt * 16u + (15u - q) : 0u);       // ... or the fullest class goes anyway
          }
          const uint32_t cl = 15u - (key & 15u);
          uint32_t row = 0u, twin = 0u;
          if (keeper && q == cl) {
            const uint32_t pos = bucket[end - cnt];
            row = row_l[pos];
            put(row);
            if (slot == 0 && head_dbl) {                       // the lane is full: the copy adds nothing of its own in the second slot's instruction
              twin = 1u;
              ++m;
              put(row_l[bucket[end - cnt + 1u]]);
              --m;
              --cnt;
            }
            --cnt;
            refresh();
          }
          row = (uint32_t)__builtin_amdgcn_readlane((int)row, (int)cl);
          twin = (uint32_t)__builtin_amdgcn_readlane((int)twin, (int)cl);
          used[slot] |= 1u << cl;
          used_row[slot] |= 1u << (row & 15u);
          m += 1u + twin;
          left -= 1u + twin;
          if (twin) break;
        }
      }
      s = e;
      __syncthreads();                     // (the buckets are refilled for the next block)
    }
    __syncthreads();
    const bool fits = m <= LIN_PAIR_CAP;
    uint16_t *out = live2 + t * LIN_LIVE2_STRIDE;
    if (fits)
      for (uint32_t i = lane; i < m; i += 64) out[2 + i] = out_l[2 + i];
    if (fits && m) {
      // the level words (lin_pair_levels), a unit of 64 pairs at a time; lanes beyond the list's last pair repeat it
      const uint32_t n_pairs = m >> 1;
      for (uint32_t u0 = 0; u0 < n_pairs; u0 += 64u) {
        const uint32_t pr = u0 + lane < n_pairs ? u0 + lane : n_pairs - 1u;
        const unsigned long long cv = kmer_index[row0 + out_l[2 + 2u * pr]] & pair_mask;
        out[2 + m + u0 + lane] = (uint16_t)lin_pair_levels(cv, lane, (uint32_t)G.npair);
      }
    }
    if (lane == 0) {
      out[0] = fits ? (uint16_t)m : (uint16_t)0;
      out[1] = 0;
      n_ent[t] = fits ? (uint16_t)m : (uint16_t)0xffffu;
    }
    __syncthreads();
  }
}
// offset (in doubles) of group g's table row; g a constant after unrolling
template <int NG>
__device__ __forceinline__ uint32_t lin_off(unsigned long long cv, int g) {
  const uint32_t idx = (uint32_t)(cv >> (6 * g)) & (g == NG - 1 ? 255u : 63u);
  return (uint32_t)g * LIN_PSTRIDE + idx * 4u;
}
// the same for a per-lane group number
__device__ __forceinline__ uint32_t lin_off_any(unsigned long long cv, uint32_t g, uint32_t ng) {
  const uint32_t idx = (uint32_t)(cv >> (6u * g)) & (g == ng - 1u ? 255u : 63u);
  return g * LIN_PSTRIDE + idx * 4u;
}

// A group's table row of a context (g a constant after unrolling).  The pairs' rows are four doubles side by side; the TRIPLE's rows
// are split into two planes of 16-byte halves (letters 0, 1 | letters 2, 3): the lanes of a wave read different triple rows, and a
// 16-byte LDS read serves eight lanes per pass -- rows of 32 bytes put them on four of the eight 16-byte bank groups, halves that
// follow each other on all eight.
template <int NG>
__device__ __forceinline__ void lin_table_row(const double *T, unsigned long long cv, int g, double2 &lo, double2 &hi) {
  if (g == NG - 1) {
    const uint32_t idx = (uint32_t)(cv >> (6 * g)) & 255u;
    const double *base = T + (NG - 1) * LIN_PSTRIDE;
    lo = *reinterpret_cast<const double2 *>(base + idx * 2u);
    hi = *reinterpret_cast<const double2 *>(base + 2 * LIN_TRI_COMBOS + idx * 2u);
  } else {
    const double2 *t = reinterpret_cast<const double2 *>(T + lin_off<NG>(cv, g));
    lo = t[0];
    hi = t[1];
  }
}

// exp(d ln2 / 128): the tables hold logits in units of ln2 / 128, so the argument reduction of bear_exp_tab is a
// rounding and an exact subtraction; 2^(j/128) from the table, degree-5 polynomial on |r| <= 1/2 unit, v_ldexp for the rest.
#define LIN_EXP_UNIT 184.66496523378731    // 128 / ln 2
__device__ __forceinline__ double lin_exp_units(double d, const double *__restrict__ tab) {
  const double kf = __builtin_rint(d), r = d - kf;
  const int ki = (int)kf;
  const double t = tab[ki & (BEAR_EXPTAB_N - 1)];
  constexpr double c1 = 0.0054152123481245727, c2 = c1 * c1 / 2.0, c3 = c1 * c1 * c1 / 6.0, c4 = c1 * c1 * c1 * c1 / 24.0,
                   c5 = c1 * c1 * c1 * c1 * c1 / 120.0;
  double p = __builtin_fma(r, c5, c4);
  p = __builtin_fma(r, p, c3);
  p = __builtin_fma(r, p, c2);
  p = __builtin_fma(r, p, c1);
  return __builtin_ldexp(__builtin_fma(t, r * p, t), ki >> 7);
}

// softmax row of one context from the group tables (the fifth logit is the zero the tables are relative to)
template <int NG, bool EXP>
__device__ __forceinline__ void lin_row(const double *T, const double *exptab, unsigned long long cv, double (&f)[5]) {
  constexpr int CH = LIN_CH;   // table rows in flight (registers: 8 per row)
  if (EXP) {              // the tables hold exp(logit) (see the table build): a product per letter and no exponential
    double e[4] = {1.0, 1.0, 1.0, 1.0};
    unsigned long long cw = cv;
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += CH) {
      double2 lo[CH], hi[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j)
        if (g0 + j < NG) {
          lin_table_row<NG>(T, cw, g0 + j, lo[j], hi[j]);
        }
#pragma unroll
      for (int j = 0; j < CH; ++j)
        if (g0 + j < NG) {
          if (g0 + j == 0) {
            e[0] = lo[j].x;
            e[1] = lo[j].y;
            e[2] = hi[j].x;
            e[3] = hi[j].y;
          } else {
            e[0] *= lo[j].x;
            e[1] *= lo[j].y;
            e[2] *= hi[j].x;
            e[3] *= hi[j].y;
          }
        }
      // the next rows' addresses wait for this product: otherwise all the rows are fetched at once (48 registers at six groups)
      if (g0 + CH < NG) asm("" : "+v"(cw) : "v"(e[0]));
    }
    const double r = bear_rcp(1.0 + ((e[0] + e[1]) + (e[2] + e[3])));
#pragma unroll
    for (int b = 0; b < 4; ++b) f[b] = e[b] * r;
    f[4] = r;
    return;
  }
  double z[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int g0 = 0; g0 < NG; g0 += CH) {
    double2 lo[CH], hi[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (g0 + j < NG) {
        lin_table_row<NG>(T, cv, g0 + j, lo[j], hi[j]);
      }
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (g0 + j < NG) {
        z[0] += lo[j].x;
        z[1] += lo[j].y;
        z[2] += hi[j].x;
        z[3] += hi[j].y;
      }
  }
  // Logits within +-600 (in natural units) for the whole wave -- anything a fitted model produces: no maximum to subtract and
  // the fifth exponential is 1.  Otherwise the shifted form (a saturated softmax needs it).
  const double za = __builtin_fmax(__builtin_fmax(__builtin_fabs(z[0]), __builtin_fabs(z[1])), __builtin_fmax(__builtin_fabs(z[2]), __builtin_fabs(z[3])));
  double s;
  if (__builtin_amdgcn_ballot_w64(!(za < 600.0 * LIN_EXP_UNIT)) == 0ull) {
    s = 1.0;
    f[4] = 1.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f[b] = lin_exp_units(z[b], exptab);
      s += f[b];
    }
  } else {
    const double m = __builtin_fmax(__builtin_fmax(__builtin_fmax(z[0], z[1]), __builtin_fmax(z[2], z[3])), 0.0);
    s = 0.0;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      f[b] = lin_exp_units(__builtin_fmax((b < 4 ? z[b] : 0.0) - m, -700.0 * LIN_EXP_UNIT), exptab);
      s += f[b];
    }
  }
  const double r = bear_rcp(s);
#pragma unroll
  for (int b = 0; b < 5; ++b) f[b] *= r;
}

// Two contexts that share every pair group (a thread's two entries of the PAIRED list): the pair rows are read and combined once,
// each context adds its own triple row.  has1 = false: there is no second context (f1 is then not meaningful).
template <int NG, bool EXP>
__device__ __forceinline__ void lin_row2(const double *T, const double *exptab, unsigned long long c0, unsigned long long c1,
                                         double (&f0)[5], double (&f1)[5]) {
  constexpr int CH = LIN_CH;
  constexpr int NP = NG - 1;                 // pair groups
  double e[4] = {EXP ? 1.0 : 0.0, EXP ? 1.0 : 0.0, EXP ? 1.0 : 0.0, EXP ? 1.0 : 0.0};
  unsigned long long cw = c0;
#pragma unroll
  for (int g0 = 0; g0 < NP; g0 += CH) {
    double2 lo[CH], hi[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (g0 + j < NP) {
        const double2 *t = reinterpret_cast<const double2 *>(T + lin_off<NG>(cw, g0 + j));
        lo[j] = t[0];
        hi[j] = t[1];
      }
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (g0 + j < NP) {
        if (EXP && g0 + j == 0) {
          e[0] = lo[j].x;
          e[1] = lo[j].y;
          e[2] = hi[j].x;
          e[3] = hi[j].y;
        } else if (EXP) {
          e[0] *= lo[j].x;
          e[1] *= lo[j].y;
          e[2] *= hi[j].x;
          e[3] *= hi[j].y;
        } else {
          e[0] += lo[j].x;
          e[1] += lo[j].y;
          e[2] += hi[j].x;
          e[3] += hi[j].y;
        }
      }
    if (g0 + CH < NP) asm("" : "+v"(cw) : "v"(e[0]));     // as in lin_row: the next rows' addresses wait for this product
  }
  double2 a0, b0, a1, b1;
  lin_table_row<NG>(T, c0, NG - 1, a0, b0);
  lin_table_row<NG>(T, c1, NG - 1, a1, b1);
  if (EXP) {
    const double x0[4] = {e[0] * a0.x, e[1] * a0.y, e[2] * b0.x, e[3] * b0.y};
    const double x1[4] = {e[0] * a1.x, e[1] * a1.y, e[2] * b1.x, e[3] * b1.y};
    const double r0 = bear_rcp(1.0 + ((x0[0] + x0[1]) + (x0[2] + x0[3])));
    const double r1 = bear_rcp(1.0 + ((x1[0] + x1[1]) + (x1[2] + x1[3])));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f0[b] = x0[b] * r0;
      f1[b] = x1[b] * r1;
    }
    f0[4] = r0;
    f1[4] = r1;
    return;
  }
  double z[2][4] = {{e[0] + a0.x, e[1] + a0.y, e[2] + b0.x, e[3] + b0.y}, {e[0] + a1.x, e[1] + a1.y, e[2] + b1.x, e[3] + b1.y}};
  double za = 0.0;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int b = 0; b < 4; ++b) za = __builtin_fmax(za, __builtin_fabs(z[k][b]));
  const bool plain = __builtin_amdgcn_ballot_w64(!(za < 600.0 * LIN_EXP_UNIT)) == 0ull;      // as in lin_row, for both contexts
  double fo[2][5];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double (&f)[5] = fo[k];
    double sum;
    if (plain) {
      sum = 1.0;
      f[4] = 1.0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f[b] = lin_exp_units(z[k][b], exptab);
        sum += f[b];
      }
    } else {
      const double m = __builtin_fmax(__builtin_fmax(__builtin_fmax(z[k][0], z[k][1]), __builtin_fmax(z[k][2], z[k][3])), 0.0);
      sum = 0.0;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        f[b] = lin_exp_units(__builtin_fmax((b < 4 ? z[k][b] : 0.0) - m, -700.0 * LIN_EXP_UNIT), exptab);
        sum += f[b];
      }
    }
    const double r = bear_rcp(sum);
#pragma unroll
    for (int b = 0; b < 5; ++b) f[b] *= r;
  }
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    f0[b] = fo[0][b];
    f1[b] = fo[1][b];
  }
}

// ---- BEAR_AMD_DETERMINISTIC: the gradient tables in 64-bit FIXED POINT.  LDS floating-point atomics from several waves land in
// whatever order the waves get there, so d/d mat is reproducible to rounding only; integer adds commute exactly.  A context's
// g_b = d L / d logit_b is bounded by its row total (x P(x, c) <= c, so w_b = f_b u P <= c_b and |g_b| <= n), hence every sum
// of them by `bound` >= the sum of all counts that enter one gradient (bear_plan_set_count_bound; default: the plan's table).
// With scale = 2^62 / bound (a power of two) every g becomes an integer BEFORE anything is added, no sum can leave 63 bits, and
// the table sums are the exact integer sums -- whatever the order of the adds, the cut of the table into tiles and blocks, the
// form of the lists.  The price: each g is rounded to bound 2^-62 once (2e-19 of the total count) instead of to its own last
// bit; a block's / launch's integer sum becomes a double at the very end (one rounding).
typedef long long lin_fx;
__device__ __forceinline__ lin_fx lin_to_fixed(double v, double scale) { return (lin_fx)__builtin_rint(v * scale); }
// The bound on the sum of |g_b| over everything that enters one gradient, from what a plan knows of its table (and of the
// other shards of its batch, bear_plan_set_count_bound): the sum of all counts -- and, in BEAR mode, where
// w_b = f u P(x, c) <= sum_j min(1, f u / j) <= 1 + u (1 + ln c), also cells (1 + u (1 + ln c_max)): tables of huge counts
// under a large h (small u) have gradients far below their counts, and the tighter bound keeps the rounding unit
// bound 2^-62 far below them.  Every block derives the same power of two from the launch's u.
struct lin_fx_bound {
  double counts, cells, ln_cmax;
};
__device__ __forceinline__ double lin_fx_scale(const lin_fx_bound &B, double u, bool ar) {
  double bound = B.counts;
  if (!ar) bound = __builtin_fmin(bound, B.cells * (1.0 + u * (1.0 + B.ln_cmax)));
  bound = __builtin_fmax(bound, 1.0);
  int e;
  (void)frexp(bound, &e);                    // bound < 2^e
  return bear_uniform_f64(ldexp(1.0, 62 - e));
}
template <typename T> __device__ __forceinline__ long long lin_bits(T v);
template <> __device__ __forceinline__ long long lin_bits<double>(double v) { return __double_as_longlong(v); }
template <> __device__ __forceinline__ long long lin_bits<lin_fx>(lin_fx v) { return v; }
template <typename T> __device__ __forceinline__ T lin_unbits(long long q);
template <> __device__ __forceinline__ double lin_unbits<double>(long long q) { return __longlong_as_double(q); }
template <> __device__ __forceinline__ lin_fx lin_unbits<lin_fx>(long long q) { return q; }
// one add into a gradient table (the table's words hold doubles or fixed-point integers, one kind per launch)
__device__ __forceinline__ void lin_gt_add(double *p, double v) { atomicAdd(p, v); }
__device__ __forceinline__ void lin_gt_add(double *p, lin_fx v) { atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }

// ---- cross-lane sums on DPP (a double = two 32-bit moves + one add per step; the generic __shfl_xor of a double costs
// two ds_bpermute round trips: measured 0.76 ms per 1e8 contexts for the 24 of them a wave needs here)
template <int CTRL, int ROW_MASK, typename T>
__device__ __forceinline__ T lin_dpp(T v) {
  const long long q = lin_bits<T>(v);
  // (mov_dpp: no "old" value -- every pattern used here gives every lane a source lane, and update_dpp's old = 0 cost a v_mov per move)
  static_assert(ROW_MASK == 0xf, "lin_dpp: all rows");
  const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)q, CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(q >> 32), CTRL, ROW_MASK, 0xf, true);
  return lin_unbits<T>(((long long)hi << 32) | (uint32_t)lo);
}
// Sums of the four gradient letters over quads, rows of 16 and the wave, TRANSPOSED: lane i ends up with the sums of ONE letter,
// lin_letter(i) -- which is how the block-level adds want them (lane <-> (group, letter)).  A quad exchanges two letters, then
// one (the other lane of a pair keeps the other half), and from there every level is one add: 33 instructions instead of the
// 80 of four separate butterflies.
__device__ __forceinline__ uint32_t lin_letter(uint32_t lane) { return ((lane & 1u) << 1) | ((lane >> 1) & 1u); }
template <typename T>
__device__ __forceinline__ T lin_quad_letter_sum(const T (&g)[4], uint32_t lane) {
  const bool o1 = lane & 1u, o2 = lane & 2u;
  T k0 = o1 ? g[2] : g[0], k1 = o1 ? g[3] : g[1];
  k0 += lin_dpp<0xB1, 0xf, T>(o1 ? g[0] : g[2]);   // quad_perm [1,0,3,2]: the pair's sums of letters 2 o1, 2 o1 + 1
  k1 += lin_dpp<0xB1, 0xf, T>(o1 ? g[1] : g[3]);
  return (o2 ? k1 : k0) + lin_dpp<0x4E, 0xf, T>(o2 ? k0 : k1);   // quad_perm [2,3,0,1]
}
template <typename T>
__device__ __forceinline__ T lin_row16_sum(T quad_sum) {
  T v = quad_sum;
  v += lin_dpp<0x124, 0xf, T>(v);   // row_ror:4: the same letter of the next quad
  v += lin_dpp<0x128, 0xf, T>(v);   // row_ror:8
  return v;
}
// v_permlane16_swap / v_permlane32_swap (gfx950) of a value with itself: every lane gets the value of the same lane of the
// neighbouring row of 16 / of the other half of the wave in the second result
template <typename T>
__device__ __forceinline__ T lin_wave_sum(T row_sum) {
  T v = row_sum;
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const long long q = lin_bits<T>(v);
    const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
    const auto a = step == 0 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto c = step == 0 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = lin_unbits<T>(((long long)c[0] << 32) | a[0]) + lin_unbits<T>(((long long)c[1] << 32) | a[1]);
  }
  return v;
}
__device__ __forceinline__ unsigned long long lin_first_lane(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// The waves that issue a tile's DMA (the last LIN_DMA_WAVES of the block) take no rows in phases A / C: with HBM busy every DMA
// instruction parks its wave for ~700 clocks, a tile is ~25 of them, and a wave that also had its share of rows reached the
// phase barrier ~4000 clocks (a fifth of a tile's time) after the others.  Rows go to the first LIN_ROW_THREADS threads.
#ifndef LIN_DMA_WAVES
#define LIN_DMA_WAVES 2
#endif
#ifndef LIN_ROW_THREADS
#define LIN_ROW_THREADS (PLN_THREADS - 64 * LIN_DMA_WAVES)
#endif
constexpr int LIN_RPT = (PLN_RMAX + LIN_ROW_THREADS - 1) / LIN_ROW_THREADS;   // contexts of a tile per row thread

// ---- phase A over the tile's live contexts (entry tid + 1024 k of the plan's list): softmax rows into fA (kept until phase C
// of the same tile) and, by lin_phase_a_store, into LDS for the items.  Contexts without counts are never looked at: nothing
// reads their rows.  The two halves are separate because a thread's rows differ from tile to tile: the LDS rows may only be
// overwritten once EVERY wave has read its rows of the previous tile back (phase C) -- the computation does not have to wait.
template <int NG, bool EXP>
__device__ __forceinline__ uint32_t lin_phase_a(pln_lds_lin &S, const lin_buf &B, uint32_t n_live, uint32_t tid_in,
                                                double (&fA)[LIN_RPT][5], unsigned long long (&cA)[LIN_RPT]) {
  static_assert(LIN_RPT == 2 && PLN_RMAX < 0xffff, "two 16-bit row numbers in one register");
  uint32_t rows = 0xffffffffu;     // 0xffff: no row
  uint32_t tid = tid_in;
  asm volatile("" : "+v"(tid));    // as in phase C: nothing derived from the thread number is worth a register across the tile loop
#pragma unroll
  for (int k = 0; k < LIN_RPT; ++k) {
    const uint32_t j = tid + LIN_ROW_THREADS * k;
    if (tid < LIN_ROW_THREADS && j < n_live) {
      const uint32_t row = B.live[1 + j];
      rows = k == 0 ? (rows & 0xffff0000u) | row : (rows & 0xffffu) | (row << 16);
      cA[k] = B.codes[row];
      lin_row<NG, EXP>(S.T, S.exptab, cA[k], fA[k]);
    }
  }
  return rows;
}
__device__ __forceinline__ void lin_phase_a_store(pln_lds_lin &S, const double (&fA)[LIN_RPT][5], uint32_t rows) {
#pragma unroll
  for (int k = 0; k < LIN_RPT; ++k) {
    const uint32_t row = (rows >> (16 * k)) & 0xffffu;
    if (row != 0xffffu) {
#pragma unroll
      for (int b = 0; b < 5; ++b) S.pri[row * 5 + b] = fA[k][b];
    }
  }
}

// ---- phase A over the PAIRED list: thread j takes entries 2 j and 2 j + 1 (they share every pair group, or the second is empty).
// Returns the two rows as lin_phase_a does (0xffff: none).
template <int NG, bool EXP>
__device__ __forceinline__ uint32_t lin_phase_a_paired(pln_lds_lin &S, const lin_buf &B, uint32_t n_ent, uint32_t tid_in,
                                                       double (&fA)[LIN_RPT][5], unsigned long long (&cA)[LIN_RPT]) {
  static_assert(LIN_RPT == 2 && LIN_RPT * LIN_ROW_THREADS == LIN_PAIR_CAP, "a thread takes one pair of entries");
  uint32_t rows = 0xffffffffu;
  uint32_t tid = tid_in;
  asm volatile("" : "+v"(tid));
  if (tid < LIN_ROW_THREADS && 2u * tid < n_ent) {
    rows = reinterpret_cast<const uint32_t *>(B.live)[1 + tid];       // entries 2 tid, 2 tid + 1 behind the two header words
    const uint32_t r0 = rows & 0xffffu, r1 = rows >> 16;              // (plan_pair_kernel: an empty entry only ever sits in the odd slot)
    cA[0] = B.codes[r0];
    cA[1] = B.codes[r1 != LIN_EMPTY ? r1 : r0];                       // (both reads in flight: a read behind a branch waited for the first)
    lin_row2<NG, EXP>(S.T, S.exptab, cA[0], cA[1], fA[0], fA[1]);
  }
  return rows;
}

// ---- the gradient of 64 consecutive contexts (one per lane: index word cv, g_b = d L / d logit_b for b < 4, nz = the lane has
// anything to add; lanes beyond the end of a list repeat its last context's word with g = 0) into the letter-major gradient tables
// GT.  One LDS fp64 atomic wave-instruction costs ~20 clocks of CU time whatever the number of active lanes (measured,
// scripts/dev/lds_atomic_lanes.hip), so what counts is the number of INSTRUCTIONS: lanes are mapped to (group, letter) pairs
//   1. all groups whose row the whole wave shares: lane 4 g + b adds the wave's sum of g_b              (one instruction)
//   2. up to four groups that a lane's row of 16 shares: lane (slot, b) of each row adds the row's sum   (one instruction)
//   3. what is left (in a sorted table: the triple of the last letters): one add per context and letter (four per group)
template <int NG, typename T = double>
__device__ __forceinline__ void lin_scatter_grad(double *GT, unsigned long long cv, const T (&g)[4], bool nz, uint32_t lane,
                                                 double (&acc)[2]) {
  const T tq = lin_quad_letter_sum<T>(g, lane), th = lin_row16_sum<T>(tq), tw = lin_wave_sum<T>(th);   // of letter lin_letter(lane)
  const uint32_t bl = lin_letter(lane);
  // How many LEADING groups does my row of 16 share (l_row), how many the whole wave (l_wave)?  In a sorted table the shared
  // groups are the leading ones; anything else is merely handled one level lower than it could be.  e = my index word xor my
  // predecessor's (wave_ror:1); a row shares the groups below the lowest set bit of the OR of e over its lanes 1..15, the
  // wave those below the lowest set bit of the OR over all lanes (lane 0's e = word 0 xor word 63 is the xor of all the
  // others: it cannot lower that bit).
  const uint32_t clo = (uint32_t)cv, chi = (uint32_t)(cv >> 32);
  const uint32_t elo = clo ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)clo, 0x13C, 0xf, 0xf, false);
  const uint32_t ehi = chi ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)chi, 0x13C, 0xf, 0xf, false);
  const bool row_first = (lane & 15u) == 0u;
  uint32_t rlo = row_first ? 0u : elo, rhi = row_first ? 0u : ehi;
  rlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rlo, 0x121, 0xf, 0xf, false);   // row_ror:1, 2, 4, 8: OR over the row in every lane
  rhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rhi, 0x121, 0xf, 0xf, false);
  rlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rlo, 0x122, 0xf, 0xf, false);
  rhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rhi, 0x122, 0xf, 0xf, false);
  rlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rlo, 0x124, 0xf, 0xf, false);
  rhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rhi, 0x124, 0xf, 0xf, false);
  rlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rlo, 0x128, 0xf, 0xf, false);
  rhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rhi, 0x128, 0xf, 0xf, false);
  uint32_t wlo = rlo | elo, whi = rhi | ehi;                     // lanes 0, 16, 32, 48 carry the steps between rows
  {
    auto a = __builtin_amdgcn_permlane16_swap(wlo, wlo, false, false);
    auto c = __builtin_amdgcn_permlane16_swap(whi, whi, false, false);
    wlo = a[0] | a[1];
    whi = c[0] | c[1];
    a = __builtin_amdgcn_permlane32_swap(wlo, wlo, false, false);
    c = __builtin_amdgcn_permlane32_swap(whi, whi, false, false);
    wlo = a[0] | a[1];
    whi = c[0] | c[1];
  }
  const unsigned long long wave_or = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)whi) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)wlo);
  const unsigned long long row_or = ((unsigned long long)rhi << 32) | rlo;
  // ... and my quad: the OR of e over its lanes 1..3
  const bool quad_first = (lane & 3u) == 0u;
  uint32_t qlo = quad_first ? 0u : elo, qhi = quad_first ? 0u : ehi;
  qlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qlo, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2], [2,3,0,1]
  qhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qhi, 0xB1, 0xf, 0xf, false);
  qlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qlo, 0x4E, 0xf, 0xf, false);
  qhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qhi, 0x4E, 0xf, 0xf, false);
  const unsigned long long quad_or = ((unsigned long long)qhi << 32) | qlo;
  // group of bit position p: p / 6 (the last group is wider: clamp); 43 / 256 ~ 1 / 6 is exact for p < 64
  const uint32_t l_wave = wave_or ? min((uint32_t)(__builtin_ctzll(wave_or) * 43) >> 8, (uint32_t)NG - 1u) : (uint32_t)NG;
  const uint32_t l_row = row_or ? min((uint32_t)(__builtin_ctzll(row_or) * 43) >> 8, (uint32_t)NG - 1u) : (uint32_t)NG;   // >= l_wave
  const uint32_t l_quad = quad_or ? min((uint32_t)(__builtin_ctzll(quad_or) * 43) >> 8, (uint32_t)NG - 1u) : (uint32_t)NG; // >= l_row
  // 1. the groups the whole wave shares
  {
    const uint32_t gq = lane >> 2;
    if (gq < l_wave && tw != T(0)) lin_gt_add(&GT[bl * LIN_GT_PLANE + (lin_off_any(cv, gq, NG) >> 2)], tw);
  }
  if (l_wave == (uint32_t)NG) return;
  // 2. the next (up to four) groups, shared by a row of 16: slot s of the row takes group l_wave + s
  {
    const uint32_t pick = l_wave + ((lane & 15u) >> 2);
    if (pick < l_row && th != T(0)) lin_gt_add(&GT[bl * LIN_GT_PLANE + (lin_off_any(cv, pick, NG) >> 2)], th);
  }
  // 2b. the first group a row does NOT share, where my quad still does: in a sorted table a row of 16 that straddles two prefix
  //     blocks has one such group (the last pair), and per context its adds hit ONE address with eight lanes at a time (LDS
  //     atomics on one address serialise).  Lane (quad, letter) adds the quad's sum instead; only the quad on the boundary is
  //     left to step 3.
  const bool quad_covers = l_row < l_quad && l_row < l_wave + 4u;    // (groups at or beyond l_wave + 4 are step 3's anyway)
  if (__builtin_amdgcn_ballot_w64(quad_covers)) {
    if (quad_covers && tq != T(0)) lin_gt_add(&GT[bl * LIN_GT_PLANE + (lin_off_any(cv, l_row, NG) >> 2)], tq);
  }
  // 3. one add per context and letter for every group not covered above
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) {
    if ((uint32_t)gq < l_wave) continue;                   // wave-uniform test
    const bool mine = nz && ((uint32_t)gq >= l_row || (uint32_t)gq >= l_wave + 4u) && !(quad_covers && (uint32_t)gq == l_row);
    if (!__builtin_amdgcn_ballot_w64(mine)) continue;
    if (mine) {
      double *gt = &GT[lin_off<NG>(cv, gq) >> 2];
#pragma unroll
      for (int b = 0; b < 4; ++b) lin_gt_add(&gt[b * LIN_GT_PLANE], g[b]);
    }
  }
}

// ---- phase C for the thread's rows: g_b = w_b - f_b s into the gradient tables (lin_scatter_grad), whole waves at a time.
template <int NG, bool DET>
__device__ __forceinline__ void lin_phase_c(pln_lds_lin &S, uint32_t n_live, uint32_t tid, uint32_t lane_in, const double (&fA)[LIN_RPT][5],
                                            const unsigned long long (&cA)[LIN_RPT], uint32_t rowA, double (&acc)[2], double gt_scale) {
  uint32_t lane = lane_in;
#pragma unroll
  for (int k = 0; k < LIN_RPT; ++k) {
    const uint32_t j0 = (tid & ~63u) + LIN_ROW_THREADS * k;  // first of this wave's 64 consecutive list entries
    if (tid >= LIN_ROW_THREADS || j0 >= n_live) continue;    // wave-uniform
    // keep LLVM from hoisting every lane-derived value of the ten NG variants out of the tile loop (that spilled 30 registers)
    asm volatile("" : "+v"(lane));
    const bool live = j0 + lane < n_live;
    const uint32_t row = live ? (rowA >> (16 * k)) & 0xffffu : 0u;      // row and index word from phase A's registers: the tile's
    const unsigned long long code = cA[k];                               // buffers are being refilled already
    double g[4] = {0.0, 0.0, 0.0, 0.0}, sw = 0.0;
    if (live) {
      double w[5];
#pragma unroll
      for (int b = 0; b < 5; ++b)           // cells an item has marked hold -w; the others still hold f_b >= 0 (one v_max_f64: see the paired form)
        asm("v_max_f64 %0, -%1, 0" : "=v"(w[b]) : "v"(S.pri[row * 5 + b]));
      sw = (((w[0] + w[1]) + w[2]) + w[3]) + w[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) g[b] = __builtin_fma(-fA[k][b], sw, w[b]);
    }
    const bool nz = sw > 0.0;              // the context holds an item (w >= 0)
    if (__builtin_amdgcn_ballot_w64(nz) == 0ull) continue;   // no item in these 64 contexts
    // entries beyond the list end repeat its last context: they add nothing and never break a run
    const uint32_t last = n_live - j0 < 64u ? n_live - j0 - 1u : 63u;   // wave-uniform
    unsigned long long cv = code;
    if (last != 63u) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)code, (int)last), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(code >> 32), (int)last);
      if (!live) cv = ((unsigned long long)hi << 32) | lo;
    }
    if (DET) {
      const lin_fx gi[4] = {lin_to_fixed(g[0], gt_scale), lin_to_fixed(g[1], gt_scale), lin_to_fixed(g[2], gt_scale), lin_to_fixed(g[3], gt_scale)};
      lin_scatter_grad<NG, lin_fx>(S.GT, cv, gi, nz, lane, acc);
    } else {
      lin_scatter_grad<NG, double>(S.GT, cv, g, nz, lane, acc);
    }
  }
}

// ---- the level word of a lane's pair inside its unit of 64 pairs (cv: the pair groups' bits of its index word; lanes beyond the end
// of a list hold the last pair's): leading pair groups shared by the unit | by my row of 16 << 4 | by my quad << 8.  e = my word xor my
// predecessor's (wave_ror:1); a row shares the groups below the lowest set bit of the OR of e over its lanes 1..15, the unit those
// below the lowest set bit of the OR over all lanes (lane 0's e = word 0 xor word 63 is the xor of all the others: it cannot
// lower that bit), a quad likewise over its lanes 1..3.  Run by plan_pair_kernel (one wave per tile) when the lists are built.
__device__ __forceinline__ uint32_t lin_pair_levels(unsigned long long cv, uint32_t lane, uint32_t np) {
  const uint32_t clo = (uint32_t)cv, chi = (uint32_t)(cv >> 32);
  const uint32_t elo = clo ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)clo, 0x13C, 0xf, 0xf, false);
  const uint32_t ehi = chi ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)chi, 0x13C, 0xf, 0xf, false);
  const bool row_first = (lane & 15u) == 0u;
  uint32_t rlo = row_first ? 0u : elo, rhi = row_first ? 0u : ehi;
  rlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rlo, 0x121, 0xf, 0xf, false);   // row_ror:1, 2, 4, 8: OR over the row in every lane
  rhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rhi, 0x121, 0xf, 0xf, false);
  rlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rlo, 0x122, 0xf, 0xf, false);
  rhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rhi, 0x122, 0xf, 0xf, false);
  rlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rlo, 0x124, 0xf, 0xf, false);
  rhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rhi, 0x124, 0xf, 0xf, false);
  rlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rlo, 0x128, 0xf, 0xf, false);
  rhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rhi, 0x128, 0xf, 0xf, false);
  uint32_t wlo = rlo | elo, whi = rhi | ehi;                     // lanes 0, 16, 32, 48 carry the steps between rows
  {
    auto a = __builtin_amdgcn_permlane16_swap(wlo, wlo, false, false);
    auto c = __builtin_amdgcn_permlane16_swap(whi, whi, false, false);
    wlo = a[0] | a[1];
    whi = c[0] | c[1];
    a = __builtin_amdgcn_permlane32_swap(wlo, wlo, false, false);
    c = __builtin_amdgcn_permlane32_swap(whi, whi, false, false);
    wlo = a[0] | a[1];
    whi = c[0] | c[1];
  }
  const unsigned long long wave_or = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)whi) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)wlo);
  const unsigned long long row_or = ((unsigned long long)rhi << 32) | rlo;
  const bool quad_first = (lane & 3u) == 0u;
  uint32_t qlo = quad_first ? 0u : elo, qhi = quad_first ? 0u : ehi;
  qlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qlo, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2], [2,3,0,1]
  qhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qhi, 0xB1, 0xf, 0xf, false);
  qlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qlo, 0x4E, 0xf, 0xf, false);
  qhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)qhi, 0x4E, 0xf, 0xf, false);
  const unsigned long long quad_or = ((unsigned long long)qhi << 32) | qlo;
  // group of bit position p: p / 6 (only pair bits are in the words: no clamp beyond np); 43 / 256 ~ 1 / 6 is exact for p < 64
  const uint32_t l_wave = wave_or ? ((uint32_t)(__builtin_ctzll(wave_or) * 43) >> 8) : np;
  const uint32_t l_row = row_or ? ((uint32_t)(__builtin_ctzll(row_or) * 43) >> 8) : np;
  const uint32_t l_quad = quad_or ? ((uint32_t)(__builtin_ctzll(quad_or) * 43) >> 8) : np;
  return l_wave | (l_row << 4) | (l_quad << 8);
}

// ---- lin_scatter_grad for 64 PAIRS of contexts (c0, c1 share every pair group; g0 / g1 their gradients, zero where there is no
// context or no item): the pair groups take the pair's SUM gs through the same wave / row-of-16 / quad levels -- one reduction, one
// run detection and one add per level for 128 contexts -- and each context adds its own g to its triple row.
template <int NG, typename T>
__device__ __forceinline__ void lin_scatter_grad_paired(double *GT, unsigned long long c0, unsigned long long c1, const T (&g0)[4],
                                                        const T (&g1)[4], bool nz0, bool nz1, uint32_t lane, uint32_t lev) {
  constexpr uint32_t NP = NG - 1;          // pair groups: the only ones the levels may share
  if (NP > 0) {
    const T gs[4] = {g0[0] + g1[0], g0[1] + g1[1], g0[2] + g1[2], g0[3] + g1[3]};
    const T tq = lin_quad_letter_sum<T>(gs, lane), th = lin_row16_sum<T>(tq), tw = lin_wave_sum<T>(th);
    const uint32_t bl = lin_letter(lane);
    const unsigned long long pm = (1ull << (6 * NP)) - 1ull;
    const unsigned long long cv = c0 & pm;
    // leading PAIR groups shared by the wave / my row of 16 / my quad: the list's level word (lin_pair_levels, at build time)
    const uint32_t l_wave = srt_uniform(lev & 15u), l_row = (lev >> 4) & 15u, l_quad = (lev >> 8) & 15u;
    // (measured and not kept: the three row addresses below through 24-bit multiplies by name and a plane offset formed once --
    // 11 -> 6 vector instructions per add, and 0.7 % SLOWER, ABAB)
    {
      const uint32_t gq = lane >> 2;
      if (gq < l_wave && tw != T(0)) lin_gt_add(&GT[bl * LIN_GT_PLANE + (lin_off_any(cv, gq, NG) >> 2)], tw);
    }
    if (l_wave < NP) {
      {
        const uint32_t pick = l_wave + ((lane & 15u) >> 2);
        if (pick < l_row && th != T(0)) lin_gt_add(&GT[bl * LIN_GT_PLANE + (lin_off_any(cv, pick, NG) >> 2)], th);
      }
      const bool quad_covers = l_row < l_quad && l_row < l_wave + 4u;
      // (no ballot in front of the divergent adds below: the branch over an empty exec mask is the wave-uniform skip, and a ballot
      // of a condition the compiler already holds as a lane mask went through a register and back -- two instructions a test)
      if (quad_covers && tq != T(0)) lin_gt_add(&GT[bl * LIN_GT_PLANE + (lin_off_any(cv, l_row, NG) >> 2)], tq);
#pragma unroll
      for (int gq = 0; gq < (int)NP; ++gq) {
        if ((uint32_t)gq < l_wave) continue;
        const bool mine = (nz0 || nz1) && ((uint32_t)gq >= l_row || (uint32_t)gq >= l_wave + 4u) && !(quad_covers && (uint32_t)gq == l_row);
        if (mine) {
          double *gt = &GT[lin_off<NG>(cv, gq) >> 2];
#pragma unroll
          for (int b = 0; b < 4; ++b) lin_gt_add(&gt[b * LIN_GT_PLANE], gs[b]);
        }
      }
    }
  }
  // the triple rows: one add per context and letter
  // (two copies of one k-mer in a lane -- the builder pairs them -- are one row: their sum goes up once)
  const bool twin = nz1 && c1 == c0;
  if (nz0 || twin) {
    double *gt = &GT[lin_off<NG>(c0, NG - 1) >> 2];
#pragma unroll
    for (int b = 0; b < 4; ++b) lin_gt_add(&gt[b * LIN_GT_PLANE], twin ? g0[b] + g1[b] : g0[b]);
  }
  const bool own1 = nz1 && !twin;
  if (own1) {
    double *gt = &GT[lin_off<NG>(c1, NG - 1) >> 2];
#pragma unroll
    for (int b = 0; b < 4; ++b) lin_gt_add(&gt[b * LIN_GT_PLANE], g1[b]);
  }
}

// ---- phase C over the PAIRED list: a thread reads both of its rows back; a wave scatters 64 pairs at once.
template <int NG, bool DET>
__device__ __forceinline__ void lin_phase_c_paired(pln_lds_lin &S, uint32_t n_ent, uint32_t tid, uint32_t lane_in,
                                                   const double (&fA)[LIN_RPT][5], const unsigned long long (&cA)[LIN_RPT], uint32_t rowA,
                                                   uint32_t levA, double gt_scale) {
  uint32_t lane = lane_in;
  const uint32_t j0 = 2u * (tid & ~63u);                      // first entry of this wave's 64 pairs
  if (tid >= LIN_ROW_THREADS || j0 >= n_ent) return;          // wave-uniform
  asm volatile("" : "+v"(lane));
  double g[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, sw[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t row = (rowA >> (16 * k)) & 0xffffu;
    if (row != LIN_EMPTY) {
      // a cell holds the item's mark -w < 0 or the untouched f >= 0: w = max(-cell, 0) as ONE v_max_f64 (fmax() canonicalises -cell
      // with a second one first)
      double w[5];
#pragma unroll
      for (int b = 0; b < 5; ++b) asm("v_max_f64 %0, -%1, 0" : "=v"(w[b]) : "v"(S.pri[row * 5 + b]));
      sw[k] = (((w[0] + w[1]) + w[2]) + w[3]) + w[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) g[k][b] = __builtin_fma(-fA[k][b], sw[k], w[b]);
    }
  }
  const bool nz[2] = {sw[0] > 0.0, sw[1] > 0.0};
  if (__builtin_amdgcn_ballot_w64(nz[0] || nz[1]) == 0ull) return;
  // lanes beyond the end of the list repeat the last pair's leading letters: they add nothing and never break a run
  const uint32_t n_pairs = (n_ent - j0) >> 1;                 // wave-uniform (both even)
  const uint32_t last = n_pairs < 64u ? n_pairs - 1u : 63u;
  unsigned long long c0 = cA[0], c1 = cA[1];
  if (last != 63u) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cA[0], (int)last), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cA[0] >> 32), (int)last);
    if (lane > last) c0 = c1 = ((unsigned long long)hi << 32) | lo;
  }
  if (DET) {
    lin_fx gi[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int b = 0; b < 4; ++b) gi[k][b] = lin_to_fixed(g[k][b], gt_scale);
    lin_scatter_grad_paired<NG, lin_fx>(S.GT, c0, c1, gi[0], gi[1], nz[0], nz[1], lane, levA);
  } else {
    lin_scatter_grad_paired<NG, double>(S.GT, c0, c1, g[0], g[1], nz[0], nz[1], lane, levA);
  }
}

// ---- the group tables of a launch, built by the whole block (n_threads threads; *t_max: a word of LDS).  A row is the sum of mat[l][a_l][b] - mat[l][a_l][4] over the group's letters;
// letter value 5 = unknown and positions beyond the group or the lag contribute nothing.
// If no context's partial sums of table rows can leave +-600 (ng rows of at most t_max each -- anything a fitted model
// produces), the tables hold exp(logit) and a context's softmax numerators are PRODUCTS of table entries: no exponential
// per context (about a third of phase A's instructions) -- returns true.  Otherwise logits in units of ln2 / 128
// (lin_exp_units) and the per-context form.  Ends with a barrier.
// `mat_lds` (NULL or lag * 25 doubles of LDS nobody else uses yet): the parameters are fetched ONCE, one element a thread, and the
// rows are formed from the copy -- a table entry is up to six parameter reads behind conditions, which the compiler turned into
// three dependent round trips to memory per entry (~7 us of every launch of the fused step: the `group tables` stamp of round 6).
__device__ __forceinline__ bool lin_build_tables(double *T, unsigned long long *t_max, const double *__restrict__ mat_g, const lin_geom &G,
                                                 int tid, int n_threads, double *mat_lds = nullptr) {
  const int lag = G.lag, n_tab = G.npair * LIN_PSTRIDE + LIN_TRI_COMBOS * 4;
  if (tid == 0) *t_max = 0ull;
  const double *mat = mat_g;
  if (mat_lds) {
    for (int k = tid; k < lag * 25; k += n_threads) mat_lds[k] = mat_g[k];
    mat = mat_lds;
    __syncthreads();
  }
  double t_abs = 0.0;
  for (int k = tid; k < n_tab; k += n_threads) {
    double v = 0.0;
    auto letter = [&](int l, int a2, int b) {
      if (a2 < 5 && l < lag) v += mat[(l * 5 + a2) * 5 + b] - mat[(l * 5 + a2) * 5 + 4];
    };
    if (k < G.npair * LIN_PSTRIDE) {
      const int g = k / LIN_PSTRIDE, r = k - g * LIN_PSTRIDE, combo = r >> 2, b = r & 3;
      const int a0 = combo / 6, a1 = combo - a0 * 6;
      letter(2 * g, a0, b);
      if (2 * g + 1 < G.tri) letter(2 * g + 1, a1, b);
    } else {
      const int r = k - G.npair * LIN_PSTRIDE, combo = r >> 2, b = r & 3;
      letter(G.tri, combo / 36, b);
      letter(G.tri + 1, (combo / 6) % 6, b);
      letter(G.tri + 2, combo % 6, b);
    }
    int kk = k;                            // (the triple's rows in two planes: lin_table_row)
    if (k >= G.npair * LIN_PSTRIDE) {
      const int r = k - G.npair * LIN_PSTRIDE, combo = r >> 2, b = r & 3;
      kk = G.npair * LIN_PSTRIDE + (b < 2 ? 0 : 2 * LIN_TRI_COMBOS) + combo * 2 + (b & 1);
    }
    T[kk] = v;
    t_abs = __builtin_fmax(t_abs, __builtin_fabs(v));
  }
  __syncthreads();                                                        // t_max has been zeroed
  // (one atomic a WAVE: 1024 LDS atomics on one word are served one after the other -- ~6 us of every launch, the `group tables` stamp)
  t_abs = bear_wave_max(t_abs);
  if ((tid & 63) == 0) atomicMax(t_max, (unsigned long long)__double_as_longlong(t_abs));      // non-negative doubles order like their bit patterns
  __syncthreads();
  const bool exp_tables = srt_uniform((uint32_t)(__longlong_as_double((long long)*t_max) * (double)G.ng < 600.0)) != 0u;
  for (int k = tid; k < n_tab; k += n_threads) T[k] = exp_tables ? exp(T[k]) : T[k] * LIN_EXP_UNIT;
  __syncthreads();
  return exp_tables;
}

// ---- fold the group tables into this block's d/d mat[l][a][b] partials: sum over the group's other letters; only b < 4 was
// accumulated (the softmax gradient of a context sums to zero over b: the last column is minus the sum of the others)
// (DET: the table words are fixed-point integers; the fold adds them as integers and the partial keeps the integer's bits)
template <bool DET = false>
__device__ __forceinline__ void lin_fold_tables(const double *GT, const lin_geom &G, int tid, int n_threads, double *__restrict__ dst) {
  using T = typename std::conditional<DET, lin_fx, double>::type;
  for (int k = tid; k < G.lag * 25; k += n_threads) {
    const int l = k / 25, r = k - l * 25, a = r / 5, b = r - a * 5;
    T s = T(0);
    auto add = [&](int row) {
      const double *gt = &GT[row];
      const T v0 = lin_unbits<T>(__double_as_longlong(gt[0])), v1 = lin_unbits<T>(__double_as_longlong(gt[LIN_GT_PLANE])),
              v2 = lin_unbits<T>(__double_as_longlong(gt[2 * LIN_GT_PLANE])), v3 = lin_unbits<T>(__double_as_longlong(gt[3 * LIN_GT_PLANE]));
      s += b == 0 ? v0 : b == 1 ? v1 : b == 2 ? v2 : b == 3 ? v3 : -((v0 + v1) + (v2 + v3));
    };
    if (l >= G.tri) {
      const int pos = l - G.tri, base = G.npair * LIN_PAIR_COMBOS;
      for (int p = 0; p < 36; ++p) {
        const int p0 = p / 6, p1 = p % 6;
        add(base + (pos == 0 ? (a * 6 + p0) * 6 + p1 : pos == 1 ? (p0 * 6 + a) * 6 + p1 : (p0 * 6 + p1) * 6 + a));
      }
    } else {
      const int g = l >> 1;
      for (int p = 0; p < 6; ++p) add(g * LIN_PAIR_COMBOS + ((l & 1) ? p * 6 + a : a * 6 + p));
    }
    bear_store_agent(&dst[k], __longlong_as_double(lin_bits<T>(s)));
  }
}

// ---- the same fold ADDED into the launch's accumulator [n_grad] (device memory, zero between launches): one hardware atomic per
// entry and block (global_atomic_add_f64; DET: 64-bit integer adds, exact in any order) instead of a partial the last block has
// to fetch -- its fixed-order sum over 256 x 325 partials (665 KB through one CU) was ~10 us of EVERY launch, a twentieth of
// configs[2]'s step.  The regular build's d/d mat was reproducible to rounding only before (LDS float atomics); it still is.
// The sums are dealt to ALL threads of the block: a letter of the triple sums 36 table rows, a letter of a pair 6 -- left to one thread
// per entry, a launch's last 75 entries kept six of the sixteen waves busy for ~8000 clocks while the others idled at the
// barrier.  Items: one per pair-letter entry (6 rows), LIN_FOLD_SLICES per triple-letter entry (4 rows each); the slices meet in
// `part` (lin_fold_items(G) doubles of LDS: the tile loop's row buffer, idle by now).  A thread reads the plane of ITS b only.
#define LIN_FOLD_SLICES 9
__host__ __device__ inline int lin_fold_items(const lin_geom &G) { return G.tri * 25 + 75 * LIN_FOLD_SLICES; }
template <bool DET = false>
__device__ __forceinline__ void lin_fold_tables_add(const double *GT, const lin_geom &G, int tid, int n_threads, double *__restrict__ accum,
                                                    double *part) {
  using T = typename std::conditional<DET, lin_fx, double>::type;
  const int n_pair_out = G.tri * 25, n_items = lin_fold_items(G);
  for (int it = tid; it < n_items; it += n_threads) {
    const int k = it < n_pair_out ? it : n_pair_out + (it - n_pair_out) / LIN_FOLD_SLICES;
    const int slice = it < n_pair_out ? 0 : (it - n_pair_out) % LIN_FOLD_SLICES;
    const int l = k / 25, r = k - l * 25, a = r / 5, b = r - a * 5;
    T s = T(0);
    auto add = [&](int row) {
      const double *gt = &GT[row];
      if (b < 4) {
        s += lin_unbits<T>(__double_as_longlong(gt[b * LIN_GT_PLANE]));
      } else {
        const T v0 = lin_unbits<T>(__double_as_longlong(gt[0])), v1 = lin_unbits<T>(__double_as_longlong(gt[LIN_GT_PLANE])),
                v2 = lin_unbits<T>(__double_as_longlong(gt[2 * LIN_GT_PLANE])), v3 = lin_unbits<T>(__double_as_longlong(gt[3 * LIN_GT_PLANE]));
        s -= (v0 + v1) + (v2 + v3);
      }
    };
    if (l >= G.tri) {
      const int pos = l - G.tri, base = G.npair * LIN_PAIR_COMBOS;
      for (int p = slice * (36 / LIN_FOLD_SLICES); p < (slice + 1) * (36 / LIN_FOLD_SLICES); ++p) {
        const int p0 = p / 6, p1 = p % 6;
        add(base + (pos == 0 ? (a * 6 + p0) * 6 + p1 : pos == 1 ? (p0 * 6 + a) * 6 + p1 : (p0 * 6 + p1) * 6 + a));
      }
    } else {
      const int g = l >> 1;
      for (int p = 0; p < 6; ++p) add(g * LIN_PAIR_COMBOS + ((l & 1) ? p * 6 + a : a * 6 + p));
    }
    part[it] = __longlong_as_double(lin_bits<T>(s));
  }
  __syncthreads();
  for (int k = tid; k < G.lag * 25; k += n_threads) {
    T s;
    if (k < n_pair_out) {
      s = lin_unbits<T>(__double_as_longlong(part[k]));
    } else {
      const double *q = &part[n_pair_out + (k - n_pair_out) * LIN_FOLD_SLICES];
      s = T(0);
#pragma unroll
      for (int j = 0; j < LIN_FOLD_SLICES; ++j) s += lin_unbits<T>(__double_as_longlong(q[j]));
    }
    // AGENT scope: the blocks of a launch sit on eight XCDs with an L2 each -- the add has to happen where all of them see it
    if (DET) {
      if (s != T(0)) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&accum[k]), (unsigned long long)lin_bits<T>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (s != T(0)) __hip_atomic_fetch_add(&accum[k], (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// ... and the last block: the accumulator's sums -> grad_out, the accumulator back to zero (the next launch starts from it)
template <bool DET = false>
__device__ __forceinline__ void lin_take_accum(double *__restrict__ accum, int n_grad, int tid, int n_threads, double *__restrict__ grad_out,
                                               bool accumulate, double inv_scale) {
  using T = typename std::conditional<DET, lin_fx, double>::type;
  for (int k = tid; k < n_grad; k += n_threads) {
    const double raw = __hip_atomic_load(&accum[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const T t = lin_unbits<T>(__double_as_longlong(raw));
    const double v = DET ? (double)t * inv_scale : (double)t;     // (DET: the launch's exact integer sum becomes a double here: one rounding)
    grad_out[k] = accumulate ? grad_out[k] + v : v;
    __hip_atomic_store(&accum[k], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- the last block to finish (bear_arrive_last): d/d mat[k] = sum over the blocks' partials [gridDim.x][LIN_MAX_GRAD] in a fixed
// order -- three threads per entry take a third of the blocks each (independent loads, consecutive threads on consecutive
// entries), their sums meet in `part` (3 * LIN_MAX_GRAD doubles of LDS).
template <bool DET = false>
__device__ __forceinline__ void lin_sum_block_partials(const double *__restrict__ grad_partials, int n_grad, double *part, int tid,
                                                       int n_threads, double *__restrict__ grad_out, bool accumulate = false,
                                                       double inv_scale = 1.0) {
  using T = typename std::conditional<DET, lin_fx, double>::type;
  const int nb = (int)gridDim.x, third = (nb + 2) / 3;
  for (int t = tid; t < 3 * n_grad; t += n_threads) {
    const int k = t % n_grad, c = t / n_grad;
    const int b0 = c * third, b1 = b0 + third < nb ? b0 + third : nb;
    // 32 independent loads in flight per thread (the lines come from memory: the L2 was just invalidated), the ragged end of a
    // thread's range as predicated loads of the same batch -- a scalar loop over it waited for every load in turn: a launch of
    // any size spent ~15 us here on one CU (scripts/dev/lin_fixed_cost.py); fixed order
    const double *src = grad_partials + k;
    T s = T(0);
    for (int b = b0; b < b1; b += 32) {
      double v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = b + j < b1 ? src[(size_t)(b + j) * LIN_MAX_GRAD] : __longlong_as_double(lin_bits<T>(T(0)));
#pragma unroll
      for (int j = 0; j < 32; ++j) s += lin_unbits<T>(__double_as_longlong(v[j]));
    }
    part[c * LIN_MAX_GRAD + k] = __longlong_as_double(lin_bits<T>(s));
  }
  __syncthreads();
  for (int k = tid; k < n_grad; k += n_threads) {
    const T t = (lin_unbits<T>(__double_as_longlong(part[k])) + lin_unbits<T>(__double_as_longlong(part[LIN_MAX_GRAD + k]))) +
                lin_unbits<T>(__double_as_longlong(part[2 * LIN_MAX_GRAD + k]));
    const double v = DET ? (double)t * inv_scale : (double)t;     // (DET: the launch's exact integer sum becomes a double here: one rounding)
    grad_out[k] = accumulate ? grad_out[k] + v : v;
  }
}

// compile-time group count from the run-time one (a build with a smaller LIN_MAX_LAG never sees the larger counts)
#define LIN_NG_CLAMP(k) ((k) <= LIN_MAX_GROUPS ? (k) : LIN_MAX_GROUPS)
#define LIN_FOR_NG(ng, CALL)                   \
  switch (ng) {                                \
    case 1: { constexpr int NG = LIN_NG_CLAMP(1); CALL; } break;   \
    case 2: { constexpr int NG = LIN_NG_CLAMP(2); CALL; } break;   \
    case 3: { constexpr int NG = LIN_NG_CLAMP(3); CALL; } break;   \
    case 4: { constexpr int NG = LIN_NG_CLAMP(4); CALL; } break;   \
    case 5: { constexpr int NG = LIN_NG_CLAMP(5); CALL; } break;   \
    case 6: { constexpr int NG = LIN_NG_CLAMP(6); CALL; } break;   \
    case 7: { constexpr int NG = LIN_NG_CLAMP(7); CALL; } break;   \
    case 8: { constexpr int NG = LIN_NG_CLAMP(8); CALL; } break;   \
    case 9: { constexpr int NG = LIN_NG_CLAMP(9); CALL; } break;   \
    default: { constexpr int NG = LIN_MAX_GROUPS; CALL; } break; \
  }
static_assert(LIN_MAX_GROUPS <= 10, "LIN_FOR_NG lists the group counts");

#ifdef LIN_STAMPS   // developer build: clocks per section of the tile loop, summed over the waves (scripts/dev/lin_stamps.py)
__device__ unsigned long long lin_stamp_sums[8];
#define LIN_STAMP(k)                                              \
  {                                                               \
    const unsigned long long now = __builtin_amdgcn_s_memtime();  \
    tph[k] += now - t_prev;                                       \
    t_prev = now;                                                 \
  }
// ... and of a launch's prologue / epilogue, thread 0 of block 0 (the sections after the arrival: of the launch's LAST block)
__device__ unsigned long long lin_pe_stamps[12];
#define LIN_PE(k)                                                 \
  if (tid == 0 && (blockIdx.x == 0 || (k) >= 9)) {                \
    const unsigned long long now = __builtin_amdgcn_s_memtime();  \
    lin_pe_stamps[k] = now - pe_prev;                             \
    pe_prev = now;                                                \
  }
#else
#define LIN_STAMP(k)
#define LIN_PE(k)
#endif
// DET (BEAR_AMD_DETERMINISTIC): the gradient tables hold fixed-point integers, gt_scale = 2^50 / bound (see lin_fx above)
// NGK: the number of letter groups as a compile-time constant (0: taken from `lag` at run time through LIN_FOR_NG).  With it a
// launch's kernel holds ONE form of phases A and C instead of ten behind a switch inside the tile loop: a tenth of the code (the
// instruction cache), no dispatch, and the register allocator sees one variant (round 6).
#define LIN_FOR_NGK(ng, CALL)                    \
  if (NGK != 0) {                                \
    constexpr int NG = NGK ? NGK : 1;            \
    CALL;                                        \
  } else {                                       \
    LIN_FOR_NG(ng, CALL)                         \
  }
template <bool AR, bool PAIRED, bool DET = false, int NGK = 0>
__global__ __launch_bounds__(PLN_THREADS, 4) void dm_linear_plan_kernel(
    const unsigned long long *__restrict__ kmer_code, const double *__restrict__ mat, int lag, bear_params prm_arg, pln_view pv,
    const double2 *__restrict__ logtab_g, double *__restrict__ partials, double *__restrict__ grad_partials,
    const bear_step_io io, double *__restrict__ grad_out, int accumulate,    // 0: the step's only launch; 2: the first of two (d/d mat stays in the accumulator); 1: the second (adds to io.out, takes d/d mat)
    const lin_fx_bound gt_bound, const bear_apply_io apply) {   // apply.theta != NULL: the Adam update by the last block (grad_out == io.out + 2)
  extern __shared__ __attribute__((aligned(16))) unsigned char srt_smem[];
  pln_lds_lin &S = *reinterpret_cast<pln_lds_lin *>(srt_smem);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = srt_uniform(tid >> 6);
#ifdef LIN_STAMPS
  unsigned long long pe_prev = __builtin_amdgcn_s_memtime();
#endif
  const bear_params prm = bear_params_of(prm_arg, io);   // device-resident parameters: constants derived in the prologue
  const double u = prm.inv_h, eps = prm.eps, eps5 = 5.0 * prm.eps;
  double eps_v = eps;       // eps in a vector register pair: an instruction takes ONE scalar operand, and x = f u + eps of the item units has two
  asm volatile("" : "+v"(eps_v));
  const lin_geom G = lin_make_geom(lag);
  const int ng = G.ng;
  double acc[2] = {0.0, 0.0};
  const double gt_scale = DET ? lin_fx_scale(gt_bound, u, AR) : 0.0;
  // every item's argument x = f u + eps has f in [0, 1] (a softmax row formed by this kernel): inside the product path's domain
  // (0, SRT_XMAX] whenever eps > 0 and u + eps <= SRT_XMAX -- the item units then skip their domain test (srt_light)
  const bool x_in_domain = srt_uniform((uint32_t)(eps > 0.0 && u > 0.0 && u + eps <= SRT_XMAX)) != 0u;

  if (tid < BEAR_LOGTAB_N) S.logtab[tid] = logtab_g[tid];
  if (tid == 0) {
    S.pri[PLN_SENTINEL] = 1.0;
    S.ticket[0] = PLN_TICKET_START(PLN_WAVES);
    S.ticket[1] = PLN_TICKET_START(PLN_WAVES);
    S.c_done = 0;
  }
  if (tid < BEAR_EXPTAB_N) S.exptab[tid] = exp2((double)tid * (1.0 / BEAR_EXPTAB_N));
  for (int k = tid; k < LIN_TAB_DOUBLES; k += PLN_THREADS) S.GT[k] = 0.0;
  LIN_PE(0)      // parameters, log / exp tables, zeroed gradient tables

  // LIN_DMA_WAVES waves issue the tile DMA (the last ones of the block); measured in round 2, when they also had rows: 16 / 8 / 4 / 2 waves: 1.92 / 1.89 / 1.87 / 1.86 ms
  // Tile descriptors reach the waves through LDS: a scalar load inside the tile loop costs every wave a full memory latency per
  // tile (~1800 clocks with HBM busy: s_load shares lgkmcnt with the LDS, so the wave's next LDS result waits for it -- the
  // `staging` stamp of round 3, 9 % of the kernel).  The first DMA wave lands descriptor `tile` (a zeroed one behind the last
  // tile) in ring slot `slot`; it is read two barriers later.
  auto stage_desc = [&](uint64_t tile, uint32_t slot_d) {
    if (wave != PLN_WAVES - LIN_DMA_WAVES) return;
    uint32_t lane = tid & 63u;
    asm volatile("" : "+v"(lane));
    pln_dma_piece(&S.desc[slot_d], pv.tiles + (tile < pv.n_tiles ? tile : pv.n_tiles), (uint32_t)sizeof(pln_tile), 0u, lane);
  };
  auto read_desc = [&](uint32_t slot_d) {
    const uint32_t *d = reinterpret_cast<const uint32_t *>(&S.desc[slot_d]);
    pln_tile ti;
    const uint32_t lo = srt_uniform(d[0]), hi = srt_uniform(d[1]);
    ti.row0 = ((uint64_t)hi << 32) | lo;
    ti.rows_items = srt_uniform(d[2]);
    ti.off16 = srt_uniform(d[3]);
    ti.hc_hr = srt_uniform(d[4]);
    ti.blk16 = srt_uniform(d[5]);
    ti.pad = ((uint64_t)srt_uniform(d[7]) << 32) | srt_uniform(d[6]);
    return ti;
  };
  auto stage = [&](const pln_tile &ti, uint64_t tile, uint32_t b) {
    const uint32_t rows = ti.rows_items >> 16;
    if (rows == 0) return;
    if (wave < PLN_WAVES - LIN_DMA_WAVES) return;
    uint32_t lane = tid & 63u;
    asm volatile("" : "+v"(lane));   // no lane-derived addresses kept (and spilled) across the tile loop
    // the three slabs of a tile as one sequence of 1 KiB pieces dealt round-robin to the issuing waves
    const uint32_t dw = wave - (PLN_WAVES - LIN_DMA_WAVES);
    // (the paired list: its length sits in the descriptor's spare word, plan_pair_kernel)
    // a launch over a SUBSET of the tiles (pv.subset): the descriptor's spare word holds the tile's number in the plan (its lists
    // are indexed by it) and, for the paired form, the length of its paired list
    const uint64_t ltile = pv.subset ? ti.pad >> 32 : tile;
    const uint32_t cb = (rows * 8u) & ~15u, bb = ti.blk16 * 16u,
                   lb = PAIRED ? ((((uint32_t)ti.pad & 0xffffu) + 2u + lin_lev_len((uint32_t)ti.pad & 0xffffu)) * 2u + 15u) & ~15u   // entries + level words
                               : ((rows + 1u) * 2u + 15u) & ~15u;
    const uint32_t pc = (cb + 1023u) >> 10, pb = (bb + 1023u) >> 10, pl = (lb + 1023u) >> 10;
    for (uint32_t q = dw; q < pc + pb + pl; q += LIN_DMA_WAVES) {
      if (q < pc) pln_dma_piece(S.buf[b].codes, kmer_code + ti.row0, cb, q, lane);
      else if (q < pc + pb) pln_dma_piece(S.buf[b].blk, pv.stream + (size_t)ti.off16 * 16, bb, q - pc, lane);
      else if (PAIRED) pln_dma_piece(S.buf[b].live, pv.live2 + ltile * LIN_LIVE2_STRIDE, lb, q - pc - pb, lane);
      else pln_dma_piece(S.buf[b].live, pv.live + ltile * PLN_LIVE_STRIDE, lb, q - pc - pb, lane);
    }
    if ((rows * 8u) & 15u) {  // odd row count: trailing word through the scalar path (see dm_prior_plan_kernel)
      const __attribute__((address_space(4))) unsigned long long *tail =
          (const __attribute__((address_space(4))) unsigned long long *)(uintptr_t)(kmer_code + ti.row0 + rows - 1);
      const unsigned long long v = *tail;
      if (tid == PLN_THREADS - 64) S.buf[b].codes[rows - 1] = v;
    }
  };

  // the block's first tile is on its way while the group tables are built (its DMA was ~2 us of every launch behind them)
  const uint64_t GR = gridDim.x;
  const pln_tile cur0 = pln_load_tile(pv, blockIdx.x), nxt0 = pln_load_tile(pv, blockIdx.x + GR);    // (prologue: scalar loads)
  LIN_PE(2)      // first descriptors (scalar loads)
  stage(cur0, blockIdx.x, 0);
  const bool exp_tables = lin_build_tables(S.T, &S.t_max, mat, G, tid, PLN_THREADS, S.pri);     // (S.pri: scratch until the first phase A)
  LIN_PE(1)      // group tables
  if (tid < SRT_NKEY) {      // (read in the epilogue; from the LDS copy of the log table, which the table build's barriers have published)
    const bear_dp o = srt_general_fast(u + eps5, (double)(tid + 1), S.logtab);
    S.tabD[tid] = o.D;
    S.tabP[tid] = o.P;
  }

  // the tile loop, compiled once per table form (one loop with both forms of phase A in it ran out of registers)
  auto tile_loop = [&](auto exp_tag) {
  constexpr bool EXP = decltype(exp_tag)::value;
  double fA[LIN_RPT][5];
  unsigned long long cA[LIN_RPT] = {0ull, 0ull};
  uint32_t n_live = 0;   // of the tile whose phase A ran last
  uint32_t rowA = 0xffffffffu;
  uint32_t levA = 0u;    // (paired form) the level word of this lane's pair: read with the entries, used by phase C
  auto phase_a = [&](const lin_buf &B, const pln_tile &ti) {
    const uint32_t rows = ti.rows_items >> 16;
    n_live = rows ? srt_uniform((uint32_t)B.live[0]) : 0u;
    rowA = 0xffffffffu;
    if (PAIRED && tid < LIN_ROW_THREADS && 2u * (tid & ~63u) < n_live) levA = B.live[2u + n_live + tid];     // (whole units: padded by the builder)
    if (PAIRED) {
      LIN_FOR_NGK(ng, rowA = (lin_phase_a_paired<NG, EXP>(S, B, n_live, tid, fA, cA)))
    } else {
      LIN_FOR_NGK(ng, rowA = (lin_phase_a<NG, EXP>(S, B, n_live, tid, fA, cA)))
    }
  };
  auto phase_c = [&]() {
    if (PAIRED) {
      LIN_FOR_NGK(ng, (lin_phase_c_paired<NG, DET>(S, n_live, tid, lane, fA, cA, rowA, levA, gt_scale)))
    } else {
      LIN_FOR_NGK(ng, (lin_phase_c<NG, DET>(S, n_live, tid, lane, fA, cA, rowA, acc, gt_scale)))
    }
  };

  pln_tile cur = cur0, nxt = nxt0;
  srt_wait_dma();
  srt_sync();
  LIN_PE(3)      // first tile landed
  phase_a(S.buf[0], cur);
  lin_phase_a_store(S, fA, rowA);
  srt_sync();
  LIN_PE(4)      // its phase A
  stage(nxt, blockIdx.x + GR, 1);
  stage_desc(blockIdx.x + 2 * GR, 2);     // descriptor j of this block's tiles lives in ring slot j & 3
  uint32_t slot = 0, c_target = 0, iter = 0;
#ifdef LIN_STAMPS
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#endif
  for (uint64_t t = blockIdx.x; t < pv.n_tiles; t += GR) {
    const lin_buf &B = S.buf[slot];
    const uint32_t rows = cur.rows_items >> 16, n_light = cur.rows_items & 0xffffu;
    const uint32_t hc = cur.hc_hr >> 16, hr = cur.hc_hr & 0xffffu;
    const pln_layout L = pln_block_layout(rows, n_light, hc, hr);
    const uint16_t *E = reinterpret_cast<const uint16_t *>(B.blk);
    const uint16_t *items = reinterpret_cast<const uint16_t *>(B.blk + L.items);
    if (tid == 0) S.ticket[slot ^ 1u] = PLN_TICKET_START(PLN_WAVES);
    // ---- B: items (tickets, dearest first): ELBO / d/dh, and -w = -f q into the item's own cell
    auto item = [&](uint32_t off, double D, double P, double x, double cnt) {
      const double fb = S.pri[off];
      double q;
      if (AR) {
        const double pp = fb + eps;
        acc[0] = __builtin_fma(cnt, bear_log_tab(pp, S.logtab), acc[0]);
        q = cnt * bear_rcp(pp);
      } else {
        acc[0] += D;
        acc[1] = __builtin_fma(eps - x, P, acc[1]);
        q = u * P;
      }
      if (cnt != 0.0) S.pri[off] = -(fb * q);
    };
    const uint32_t n_hcu = (hc + 63u) >> 6, n_hru = AR ? 0u : (hr + 63u) >> 6, n_units = (n_light + 63u) >> 6;
    const uint32_t n_work = n_hcu + n_hru + n_units;
    PLN_FOR_UNITS_F(w, &S.ticket[slot], n_work, wave, PLN_WAVES) {
      if (w < n_hcu) {
        const uint32_t i = w * 64u + lane;
        if (i < hc) {
          const uint32_t off = reinterpret_cast<const uint16_t *>(B.blk + L.hoff)[i];
          const double cnt = (double)reinterpret_cast<const uint32_t *>(B.blk + L.hcnt)[i];
          const double x = __builtin_fma(S.pri[off], u, eps);
          bear_dp o = {0.0, 0.0};
          if (!AR) o = srt_general_fast(x, cnt, S.logtab);
          item(off, o.D, o.P, x, cnt);
        }
        continue;
      }
      if (w < n_hcu + n_hru) {  // contexts with a large total: context terms only (no gradient: the base cancels)
        // (totals up to PLN_NBIG are in the plan's histogram -- pln_big_totals below: on a k-mer table the whole unit is skipped)
        const uint32_t i = (w - n_hcu) * 64u + lane;
        const double n = i < hr ? reinterpret_cast<const double *>(B.blk + L.hn)[i] : 0.0;
        const bool mine = i < hr && !pln_in_big_hist(pv, n);
        if (__builtin_amdgcn_ballot_w64(mine)) {
          if (mine) {
            const bear_dp o = srt_general_fast(u + eps5, n, S.logtab);
            acc[0] -= o.D;
            acc[1] = __builtin_fma(u, o.P, acc[1]);
          }
        }
        continue;
      }
      const uint32_t un = n_work - 1u - w;
      uint32_t cmin, cmax;
      const uint32_t ci[1] = {pln_unit_counts(E, n_light, un, lane, &cmin, &cmax)};
      const uint32_t off = items[un * 64u + lane];
      const double x[1] = {__builtin_fma(S.pri[off], u, eps_v)};
      bear_dp o[1] = {{0.0, 0.0}};
      if (!AR) srt_light<1, true>(x, ci, cmin, cmax, S.logtab, o, x_in_domain);
      // (no test for the padding of a tile's last unit: its lanes read the neutral cell -- 1.0 -- with a count of zero, so D = P = 0
      // add nothing and nothing is written; the test cost a compare and, through the accumulators' two paths, four 64-bit moves)
      item(off, o[0].D, o[0].P, x[0], (double)ci[0]);
    }
    LIN_STAMP(0)     // B: items
    srt_wait_dma();  // the next tile's codes and plan block (issued a whole iteration ago)
    LIN_STAMP(1)     // wait for the DMA
    srt_sync();      // ... and every item of this tile has left its mark: nobody reads this tile's buffers any more
    LIN_STAMP(2)     // barrier after the items
    cur = nxt;
    nxt = read_desc((iter + 2u) & 3u);    // landed an iteration ago; this block's DMA wave has waited for it, the barrier published it
    stage(nxt, t + 2 * GR, slot);   // the tile after next lands while phases C, A and the next tile's B run
    stage_desc(t + 3 * GR, (iter + 3u) & 3u);
    ++iter;
    // ---- C of this tile (rows, index words and softmax rows from phase A's registers), A of the next
    LIN_STAMP(3)     // staging the tile after next
    phase_c();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my reads of this tile's LDS rows have returned
    LIN_STAMP(4)     // C
    if (lane == 0) atomicAdd(&S.c_done, 1u);
    c_target += PLN_WAVES;
    phase_a(S.buf[slot ^ 1u], cur);
    LIN_STAMP(5)     // A (compute)
    while (pln_peek(&S.c_done) < c_target) __builtin_amdgcn_s_sleep(1);   // every wave has: the rows may be overwritten
    LIN_STAMP(6)     // wait for the other waves' read-backs
    lin_phase_a_store(S, fA, rowA);
    srt_sync();      // the next tile's rows are in place for its items
    LIN_STAMP(7)     // row stores + barrier
    slot ^= 1u;
  }
#ifdef LIN_STAMPS
  if (lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&lin_stamp_sums[k], tph[k]);
#endif
  };
  if (exp_tables) tile_loop(std::true_type{});
  else tile_loop(std::false_type{});
  srt_wait_dma();
  __syncthreads();
  LIN_PE(5)      // tile loop
  // ---- items / contexts that overflowed to the plan's global lists (very dense tiles): self-contained
  const uint64_t gtid = (uint64_t)blockIdx.x * PLN_THREADS + tid, gsz = (uint64_t)gridDim.x * PLN_THREADS;
  for (uint64_t i = gtid; i < pv.n_heavy_col; i += gsz) {
    const pln_heavy_col h = pv.heavy_col[i];
    const uint64_t row = h.off / 5u;
    const uint32_t b = (uint32_t)(h.off - row * 5u);
    unsigned long long cv = 0ull;
    double f[5];
    cv = kmer_code[row];
    if (exp_tables) {
      LIN_FOR_NGK(ng, (lin_row<NG, true>(S.T, S.exptab, cv, f)))
    } else {
      LIN_FOR_NGK(ng, (lin_row<NG, false>(S.T, S.exptab, cv, f)))
    }
    double q;
    if (AR) {
      const double pp = f[b] + eps;
      acc[0] = __builtin_fma((double)h.c, bear_log_tab(pp, S.logtab), acc[0]);
      q = (double)h.c * bear_rcp(pp);
    } else {
      const double x = __builtin_fma(f[b], u, eps);
      const bear_dp o = srt_general_fast(x, (double)h.c, S.logtab);
      acc[0] += o.D;
      acc[1] = __builtin_fma(eps - x, o.P, acc[1]);
      q = u * o.P;
    }
    const double w = f[b] * q;
    for (int g = 0; g < ng; ++g) {
      double *gt = &S.GT[lin_off_any(cv, (uint32_t)g, (uint32_t)ng) >> 2];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        // (an overflow item is ONE cell of its context: the context's g is the sum of these per-item parts, each bounded by its count)
        const double gv = (bb == (int)b ? w : 0.0) - f[bb] * w;
        if (DET) lin_gt_add(&gt[bb * LIN_GT_PLANE], lin_to_fixed(gv, gt_scale));
        else lin_gt_add(&gt[bb * LIN_GT_PLANE], gv);
      }
    }
  }
  for (uint64_t i = gtid; !AR && i < pv.n_heavy_row; i += gsz) {
    const double n = pv.heavy_row[i].n;
    if (pln_in_big_hist(pv, n)) continue;
    const bear_dp o = srt_general_fast(u + eps5, n, S.logtab);
    acc[0] -= o.D;
    acc[1] = __builtin_fma(u, o.P, acc[1]);
  }
  if (!AR) pln_big_totals(pv, u + eps5, u, gtid, gsz, S.logtab, acc[0], acc[1]);      // totals in (SRT_CL, PLN_NBIG]: the plan's histogram
  if (!AR && pv.hist && blockIdx.x == 0 && tid < SRT_CL) {  // context terms of the small totals: the plan's histogram
    const double m = (double)pv.hist[tid];
    acc[0] -= m * S.tabD[tid];
    acc[1] = __builtin_fma(u * m, S.tabP[tid], acc[1]);
  }
  __syncthreads();
  LIN_PE(6)      // overflow lists, histograms
  lin_fold_tables_add<DET>(S.GT, G, (int)tid, PLN_THREADS, grad_partials, S.pri);     // (grad_partials: the launch's accumulator, bear_ws::lin_accum)
  LIN_PE(7)      // fold + adds into the accumulator
  block_store_partials<2, true>(acc, partials);      // (io.out is never NULL here: both entry points sum in this launch)
  LIN_PE(8)      // block sums
#ifdef LIN_STAMPS
  pe_prev = __builtin_amdgcn_s_memtime();
#endif
  if (!bear_arrive_last(io.arrive())) return;        // (its s_waitcnt vmcnt(0) covers the atomics: they are acknowledged before a block arrives)
  __syncthreads();
  LIN_PE(9)      // the last block's arrival
  // a step of two launches (paired tiles, then the tiles that kept their plain lists): the first leaves its d/d mat in the
  // accumulator, the second takes the sum of both -- in the deterministic mode ONE conversion of the exact integer total, whatever
  // the split of the tiles between the two forms
  // (waves 4.. take the accumulator while waves 0-3 fetch the block partials in bear_finalize_in_block: two round trips to memory side by side)
  if (accumulate != 2 && tid >= 256) lin_take_accum<DET>(grad_partials, lag * 25, (int)tid - 256, PLN_THREADS - 256, grad_out, false, DET ? 1.0 / gt_scale : 1.0);
  LIN_PE(10)     // d/d mat out of the accumulator
  bear_finalize_in_block(partials, 2, io.out, io.arrive(), accumulate == 1);
  bear_apply_in_block(apply, io.out);
  LIN_PE(11)     // sums of the partials, the update
}

// ---- the bear_net / linear optimizer step on the device (HIP-graph replay) ---------------------------------------
// theta = {h_signed, AR parameters...} contiguous (the kernels derive 1/h from it in their prologue); adam_vec_kernel is tf.keras Adam on the whole
// vector with gradients grad[k] * scale (k = 0: d/dh from out[1], skipped in AR mode; k >= 1: d/d mat).
// ONE block: every thread reads the step counter before the barrier and thread 0 advances it after it, so the update and the
// tick are a single launch (a step is launch-bound on small shards: scripts/dev/step_latency.py).
__global__ __launch_bounds__(1024) void adam_vec_kernel(const bear_apply_io A, const double *__restrict__ packed) {
  bear_adam_update(A, packed, (int)threadIdx.x, 1024);
}


// bear_levels.h -- internal: one prefix level of a k-mer-sorted batch (kernels_cnn.h, cnn_level_io), built in bear_count.hip
// (rocPRIM scan) and owned by a plan (bear_hip.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct bear_level_dev {
  uint64_t n;                  // rows of this level: the distinct prefixes of `letters` letters of the rows below, in their order
  int letters;
  unsigned long long *codes;   // [n] packed contexts: the prefix, every later letter "unknown" (5)
  uint32_t *parent_of_below;   // [n_below] row of this level that a row of the level below belongs to
  uint32_t *child_start;       // [n + 1] first row of the level below of each row (its rows are neighbours)
  double *rows;                // [n][16] the level's layer-1 sums (forward) / dT1 rows (backward)
  uint64_t bytes;              // what the owner of the level has accounted for it (bear_plan::bytes)
};

// A WINDOW TABLE of a position p of the convolutional AR function (kernels_cnn.h): what position p contributes to a context's
// layer-1 sums is a function of its window -- letters [p, p + fw) -- alone, and a batch holds at most (A + 2)^fw distinct windows
// (65 536 for 8 letters of ACGT) however many contexts it has.  One row of 16 layer-1 sums per DISTINCT window of the batch;
// contexts point at their window's row (forward: a gather) and are listed by window (backward: the window's dT1 row is the sum of
// its contexts' rows -- `perm`, sorted by window, children of a window contiguous in it).
struct bear_window_dev {
  int pos;                     // the position
  uint64_t n;                  // distinct windows
  unsigned long long *codes;   // [n] packed contexts: the window's letters at [pos, pos + fw), every other letter "unknown" (5)
  uint32_t *row_of_context;    // [n_rows] the window row of each context
  uint32_t *perm;              // [n_rows] the contexts sorted by window
  uint32_t *child_start;       // [n + 1] perm[child_start[w] .. child_start[w + 1]) are the contexts of window w
  double *rows;                // [n][16] the windows' layer-1 contributions (forward) / dT1 sums (backward)
  uint64_t bytes;
};
// Builds the table of position `pos` for the packed contexts `codes` (any order).  Allocates; synchronises `stream`.
int bear_window_build(const unsigned long long *codes, uint64_t n_rows, int pos, int fw, bear_window_dev *out, hipStream_t stream);
void bear_window_free(bear_window_dev *wt);

// Builds `out` from the packed contexts of the level below (equal prefixes must be neighbours: a k-mer-sorted batch; any other
// order is still correct, a run is then one row).  Allocates out's arrays (the caller frees them); synchronises `stream`.
int bear_level_build(const unsigned long long *codes_below, uint64_t n_below, int letters, bear_level_dev *out, hipStream_t stream);
void bear_level_free(bear_level_dev *lv);

// Deterministic build (-DBEAR_DET_BUILD): lists that were filled through atomic cursors, in whatever order the blocks got there,
// are put into a canonical order -- records of `width` bytes (4, 8 or 16), ascending as unsigned integers (16: first word, then
// second) -- so that the plan of a table is the same bits in every run.  Synchronises `stream`.
int bear_canonical_order(void *records, uint64_t n, int width, hipStream_t stream);

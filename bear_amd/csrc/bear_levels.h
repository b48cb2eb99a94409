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

// Builds `out` from the packed contexts of the level below (equal prefixes must be neighbours: a k-mer-sorted batch; any other
// order is still correct, a run is then one row).  Allocates out's arrays (the caller frees them); synchronises `stream`.
int bear_level_build(const unsigned long long *codes_below, uint64_t n_below, int letters, bear_level_dev *out, hipStream_t stream);
void bear_level_free(bear_level_dev *lv);

// Deterministic build (-DBEAR_DET_BUILD): lists that were filled through atomic cursors, in whatever order the blocks got there,
// are put into a canonical order -- records of `width` bytes (4, 8 or 16), ascending as unsigned integers (16: first word, then
// second) -- so that the plan of a table is the same bits in every run.  Synchronises `stream`.
int bear_canonical_order(void *records, uint64_t n, int width, hipStream_t stream);

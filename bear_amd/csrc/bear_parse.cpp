// bear_parse.cpp -- host-side reader for the summarize.py count-table format
// (bear_model/summarize.py:429-449).  Replaces the CsvDataset + JSON-decode path of
// bear_model/dataloader.py:35-46: one pass over the mapped file, integers parsed in place,
// output planar by dataset column as uint32 so each column can be uploaded as one [N,5] slab.
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/bear_hip.h"

namespace {
struct mapped_file {
  const char *data = nullptr;
  size_t size = 0;
  int fd = -1;
  int open_ro(const char *path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return BEAR_ERR_IO;
    struct stat st;
    if (fstat(fd, &st) != 0) return BEAR_ERR_IO;
    size = (size_t)st.st_size;
    if (size == 0) return BEAR_OK;
    void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) return BEAR_ERR_IO;
    data = static_cast<const char *>(p);
    return BEAR_OK;
  }
  ~mapped_file() {
    if (data) munmap(const_cast<char *>(data), size);
    if (fd >= 0) ::close(fd);
  }
};

inline bool blank_line(const char *b, const char *e) {
  for (; b < e; ++b)
    if (*b != ' ' && *b != '\r' && *b != '\t') return false;
  return true;
}
}  // namespace

extern "C" int bear_count_rows(const char *path, uint64_t *n_rows_out) {
  if (!path || !n_rows_out) return BEAR_ERR_INVALID_ARG;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  uint64_t n = 0;
  const char *p = f.data, *end = f.data + f.size;
  while (p < end) {
    const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    const char *le = nl ? nl : end;
    if (!blank_line(p, le)) ++n;
    p = le + 1;
  }
  *n_rows_out = n;
  return BEAR_OK;
}

extern "C" int bear_parse_counts_tsv(const char *path, int num_ds, int lag, uint64_t max_rows, char *kmers,
                                     uint32_t *counts, uint64_t *n_rows_out) {
  if (!path || !counts || !n_rows_out || num_ds < 1 || lag < 0) return BEAR_ERR_INVALID_ARG;
  *n_rows_out = 0;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  const char *p = f.data, *end = f.data + f.size;
  uint64_t row = 0;
  const int per_row = num_ds * BEAR_ROW_WIDTH;
  while (p < end && row < max_rows) {
    const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    const char *le = nl ? nl : end;
    if (blank_line(p, le)) {
      p = le + 1;
      continue;
    }
    const char *tab = static_cast<const char *>(memchr(p, '\t', (size_t)(le - p)));
    if (!tab || (tab - p) != lag) return BEAR_ERR_PARSE;
    if (kmers) memcpy(kmers + row * (uint64_t)lag, p, (size_t)lag);
    const char *q = tab + 1;
    int got = 0;
    while (q < le) {
      unsigned ch = (unsigned char)*q;
      if (ch >= '0' && ch <= '9') {
        uint64_t v = 0;
        while (q < le && (unsigned char)*q >= '0' && (unsigned char)*q <= '9') {
          v = v * 10 + (uint64_t)(*q - '0');
          if (v > 0xffffffffull) return BEAR_ERR_PARSE;  // beyond KMC's counter range
          ++q;
        }
        if (q < le && *q == '.') {  // tolerate "12.0"
          ++q;
          while (q < le && *q == '0') ++q;
        }
        if (got >= per_row) return BEAR_ERR_PARSE;
        const int ds = got / BEAR_ROW_WIDTH, b = got % BEAR_ROW_WIDTH;
        counts[((uint64_t)ds * max_rows + row) * BEAR_ROW_WIDTH + b] = (uint32_t)v;
        ++got;
      } else if (ch == '[' || ch == ']' || ch == ',' || ch == ' ' || ch == '\r') {
        ++q;
      } else {
        return BEAR_ERR_PARSE;
      }
    }
    if (got != per_row) return BEAR_ERR_PARSE;
    ++row;
    p = le + 1;
  }
  *n_rows_out = row;
  return BEAR_OK;
}

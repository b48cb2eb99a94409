// bear_parse.cpp -- host-side reader for the summarize.py count-table format
// (bear_model/summarize.py:429-449).  Replaces the CsvDataset + JSON-decode path of
// bear_model/dataloader.py:35-46: one pass over the mapped file, integers parsed in place,
// output planar by dataset column as uint32 so each column can be uploaded as one [N,5] slab.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>

#include <thread>
#include <vector>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/bear_hip.h"

// host threads of the parser / row counter / writer: one chunk of the mapped text each (BEAR_PARSE_THREADS overrides the count)
#ifndef BEAR_MAX_HOST_THREADS
#define BEAR_MAX_HOST_THREADS 64u   // measured on the 256-thread host of an MI355X box: 16 .. 64 threads parse a 5.2 GB table in 0.39 s, 128 / 256 in 0.55
#endif

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

// The page tables of a thread's chunk of the mapped file in one call (MADV_POPULATE_READ, Linux 5.14+) instead of one minor
// fault per 4 KiB page: a 5 GB table is 1.3 M page-cache pages, and faulting them in one by one from 64 threads was most of the
// parser's 0.39 s.  A kernel that does not know the advice returns EINVAL: the pages then come in by faults as before.
inline void populate_read(const char *b, const char *e) {
  if (e <= b) return;
  const uintptr_t a0 = reinterpret_cast<uintptr_t>(b) & ~(uintptr_t)4095;
  (void)madvise(reinterpret_cast<void *>(a0), (size_t)(reinterpret_cast<uintptr_t>(e) - a0), 22 /* MADV_POPULATE_READ */);
}

inline bool blank_line(const char *b, const char *e) {
  for (; b < e; ++b)
    if (*b != ' ' && *b != '\r' && *b != '\t') return false;
  return true;
}
}  // namespace

static uint64_t count_lines_mt(const char *base, size_t size);

extern "C" int bear_count_rows(const char *path, uint64_t *n_rows_out) {
  if (!path || !n_rows_out) return BEAR_ERR_INVALID_ARG;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  *n_rows_out = count_lines_mt(f.data, f.size);
  return BEAR_OK;
}

namespace {
// Parses the lines of [p, end) into rows row .. ; returns the status and the number of rows written.
// `keep(g)` (g = index of the line among the non-blank lines of [p, end), counted from g0) selects the lines that are decoded;
// the others are skipped at memchr speed.  Kept lines fill consecutive rows.
struct keep_all {
  bool operator()(uint64_t) const { return true; }
};
template <class Keep>
int parse_range(const char *p, const char *end, int num_ds, int lag, uint64_t max_rows, uint64_t row, uint64_t row_limit,
                char *kmers, uint32_t *counts, uint64_t *rows_done, uint64_t g0 = 0, Keep keep = Keep()) {
  const int per_row = num_ds * BEAR_ROW_WIDTH;
  const uint64_t row_begin = row;
  uint64_t g = g0;
  while (p < end && row < row_limit) {
    const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    const char *le = nl ? nl : end;
    if (blank_line(p, le)) {
      p = le + 1;
      continue;
    }
    if (!keep(g++)) {
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
  *rows_done = row - row_begin;
  return BEAR_OK;
}

uint64_t count_lines(const char *p, const char *end) {
  uint64_t n = 0;
  while (p < end) {
    const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    const char *le = nl ? nl : end;
    if (!blank_line(p, le)) ++n;
    p = le + 1;
  }
  return n;
}
}  // namespace

// Non-blank lines of a mapped text, one chunk (cut at a line start) per hardware thread: the row count of a 5 GB table is a
// pass at memory speed instead of half a second of one thread's memchr.
static uint64_t count_lines_mt(const char *base, size_t size) {
  const char *end = base + size;
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *env = getenv("BEAR_PARSE_THREADS")) nt = (unsigned)atoi(env);
  if (nt < 1) nt = 1;
  if (nt > BEAR_MAX_HOST_THREADS) nt = BEAR_MAX_HOST_THREADS;
  if (size < (size_t)(1u << 20)) nt = 1;
  if (nt == 1) return count_lines(base, end);
  std::vector<const char *> cut(nt + 1);
  cut[0] = base;
  cut[nt] = end;
  for (unsigned k = 1; k < nt; ++k) {
    const char *p = base + (size / nt) * k;
    const char *nl = p < end ? static_cast<const char *>(memchr(p, '\n', (size_t)(end - p))) : nullptr;
    cut[k] = nl ? nl + 1 : end;
    if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
  }
  std::vector<uint64_t> part(nt, 0);
  std::vector<std::thread> th;
  for (unsigned k = 0; k < nt; ++k) th.emplace_back([&, k] { part[k] = count_lines(cut[k], cut[k + 1]); });
  for (auto &t : th) t.join();
  uint64_t n = 0;
  for (unsigned k = 0; k < nt; ++k) n += part[k];
  return n;
}

// `wc -l` of a file (the reference's num_kmers, models/train_bear_net.py:52-55: newline bytes, blank lines and a sparse file's
// header included), on all host threads: the Python block loop of the driver took five times as long as parsing the table.
extern "C" int bear_count_newlines(const char *path, uint64_t *n_out) {
  if (!path || !n_out) return BEAR_ERR_INVALID_ARG;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *env = getenv("BEAR_PARSE_THREADS")) nt = (unsigned)atoi(env);
  if (nt < 1) nt = 1;
  if (nt > BEAR_MAX_HOST_THREADS) nt = BEAR_MAX_HOST_THREADS;
  if (f.size < (size_t)(1u << 20)) nt = 1;
  std::vector<uint64_t> part(nt, 0);
  auto count = [&](unsigned k) {
    const char *b = f.data + (f.size / nt) * k, *e = k + 1 == nt ? f.data + f.size : f.data + (f.size / nt) * (k + 1);
    populate_read(b, e);
    uint64_t n = 0;
    for (const char *p = b; p < e; ++p) n += *p == '\n';
    part[k] = n;
  };
  if (nt == 1) count(0);
  else {
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k) th.emplace_back(count, k);
    for (auto &t : th) t.join();
  }
  uint64_t n = 0;
  for (unsigned k = 0; k < nt; ++k) n += part[k];
  *n_out = n;
  return BEAR_OK;
}

// Text decoding is the first-epoch cost of a large table (~60-80 GB of text at 1e9 rows), so the file is
// cut at line boundaries into one chunk per hardware thread: pass 1 counts the rows of each chunk, a prefix sum gives
// every chunk its first row, pass 2 parses the chunks in place into the shared output arrays.
extern "C" int bear_parse_counts_tsv(const char *path, int num_ds, int lag, uint64_t max_rows, char *kmers,
                                     uint32_t *counts, uint64_t *n_rows_out) {
  if (!path || !counts || !n_rows_out || num_ds < 1 || lag < 0) return BEAR_ERR_INVALID_ARG;
  *n_rows_out = 0;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  const char *base = f.data, *end = f.data + f.size;
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *env = getenv("BEAR_PARSE_THREADS")) nt = (unsigned)atoi(env);
  if (nt < 1) nt = 1;
  if (nt > BEAR_MAX_HOST_THREADS) nt = BEAR_MAX_HOST_THREADS;
  if (f.size < (size_t)(1u << 20)) nt = 1;           // small files: not worth the threads
  // chunk boundaries at line starts
  std::vector<const char *> cut(nt + 1);
  cut[0] = base;
  cut[nt] = end;
  for (unsigned k = 1; k < nt; ++k) {
    const char *p = base + (f.size / nt) * k;
    const char *nl = p < end ? static_cast<const char *>(memchr(p, '\n', (size_t)(end - p))) : nullptr;
    cut[k] = nl ? nl + 1 : end;
    if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
  }
  std::vector<uint64_t> first(nt + 1, 0), done(nt, 0);
  std::vector<int> status(nt, BEAR_OK);
  if (nt == 1) {
    st = parse_range<keep_all>(base, end, num_ds, lag, max_rows, 0, max_rows, kmers, counts, &done[0]);
    if (st != BEAR_OK) return st;
    *n_rows_out = done[0];
    return BEAR_OK;
  }
  {
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k)
      th.emplace_back([&, k] {
        populate_read(cut[k], cut[k + 1]);
        first[k + 1] = count_lines(cut[k], cut[k + 1]);
      });
    for (auto &t : th) t.join();
  }
  for (unsigned k = 0; k < nt; ++k) first[k + 1] += first[k];
  {
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k)
      th.emplace_back([&, k] {
        const uint64_t lim = first[k + 1] < max_rows ? first[k + 1] : max_rows;
        if (first[k] >= lim) return;
        status[k] = parse_range<keep_all>(cut[k], cut[k + 1], num_ds, lag, max_rows, first[k], lim, kmers, counts, &done[k]);
      });
    for (auto &t : th) t.join();
  }
  uint64_t total = 0;
  for (unsigned k = 0; k < nt; ++k) {
    if (status[k] != BEAR_OK) return status[k];
    total += done[k];
  }
  *n_rows_out = total;
  return BEAR_OK;
}

// ------------------------------------------------------------------ the sparse row format
// `kmer; [[ds, col], ...]; [value, ...]` (bear_model/dataloader.py:52-109: CsvDataset with ';' + two decode_json calls +
// SparseTensor -> to_dense).  One pass over the mapped text; rows land in the same planar layout as the dense reader's.
extern "C" int bear_parse_sparse_counts(const char *path, int num_ds, int width, int lag, uint64_t skip_lines, uint64_t max_rows,
                                        char *kmers, uint32_t *counts, uint64_t *n_rows_out) {
  if (!path || !n_rows_out || num_ds < 1 || width < 1 || lag < 0 || (max_rows && (!counts || (lag && !kmers)))) return BEAR_ERR_INVALID_ARG;
  *n_rows_out = 0;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  if (max_rows) memset(counts, 0, sizeof(uint32_t) * (size_t)num_ds * max_rows * (size_t)width);
  const char *p = f.data, *end = f.data + f.size;
  uint64_t row = 0, line_no = 0;
  std::vector<long long> pos;
  while (p < end && row < max_rows) {
    const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    const char *le = nl ? nl : end;
    const char *q = p;
    p = le + 1;
    if (line_no++ < skip_lines || blank_line(q, le)) continue;
    const char *s1 = static_cast<const char *>(memchr(q, ';', (size_t)(le - q)));
    const char *s2 = s1 ? static_cast<const char *>(memchr(s1 + 1, ';', (size_t)(le - s1 - 1))) : nullptr;
    if (!s2) return BEAR_ERR_PARSE;
    const char *kb = q, *ke = s1;
    while (kb < ke && (*kb == ' ' || *kb == '\t')) ++kb;
    while (ke > kb && (ke[-1] == ' ' || ke[-1] == '\t' || ke[-1] == '\r')) --ke;
    if ((int)(ke - kb) != lag) return BEAR_ERR_PARSE;
    memcpy(kmers + row * (size_t)lag, kb, (size_t)lag);
    pos.clear();
    for (const char *c = s1 + 1; c < s2;) {      // the integers of the index list, in order
      if ((*c >= '0' && *c <= '9') || *c == '-') {
        char *stop = nullptr;
        pos.push_back(strtoll(c, &stop, 10));
        if (stop == c) return BEAR_ERR_PARSE;   // a '-' with no digit behind it: strtoll converts nothing and would never advance
        c = stop;
      } else {
        ++c;
      }
    }
    if (pos.size() & 1) return BEAR_ERR_PARSE;
    size_t k = 0;
    for (const char *c = s2 + 1; c < le;) {      // one value per index pair (the reference decodes them as floats)
      if ((*c >= '0' && *c <= '9') || *c == '-' || *c == '.') {
        char *stop = nullptr;
        const double v = strtod(c, &stop);
        if (stop == c) return BEAR_ERR_PARSE;
        c = stop;
        if (2 * k + 1 >= pos.size()) return BEAR_ERR_PARSE;
        const long long d = pos[2 * k], col = pos[2 * k + 1];
        if (d < 0 || d >= num_ds || col < 0 || col >= width || !(v >= 0.0) || v > 4294967295.0 || v != (double)(uint32_t)v) return BEAR_ERR_PARSE;
        counts[((size_t)d * max_rows + row) * (size_t)width + (size_t)col] = (uint32_t)v;
        ++k;
      } else {
        ++c;
      }
    }
    if (2 * k != pos.size()) return BEAR_ERR_PARSE;
    ++row;
  }
  *n_rows_out = row;
  return BEAR_OK;
}

// ------------------------------------------------------------------ one rank's rows of a row-sharded table
// Training shards every batch over the ranks (bear_net.py:273 `experimental_distribute_dataset`): of batch k = global rows
// [kB, min(kB + B, N)) with m rows, rank r owns the contiguous piece [lo, hi) with base = m / W, extra = m % W,
// lo = r base + min(r, extra), hi = lo + base + (r < extra)  (bear_amd.dist.shard_rows).  A rank decodes only its own lines;
// the other lines are stepped over at memchr speed, so W ranks decode 1/W of the text each.
namespace {
struct shard_map {
  uint64_t B, N, W, r;
  void piece(uint64_t m, uint64_t *lo, uint64_t *hi) const {
    const uint64_t base = m / W, extra = m % W;
    *lo = r * base + (r < extra ? r : extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
  }
  bool member(uint64_t g) const {
    if (g >= N) return false;
    const uint64_t a = (g / B) * B, m = (N - a < B) ? N - a : B;
    uint64_t lo, hi;
    piece(m, &lo, &hi);
    return g - a >= lo && g - a < hi;
  }
  uint64_t below(uint64_t G) const {   // number of member rows with global index < G
    if (G > N) G = N;
    uint64_t lo, hi;
    piece(B, &lo, &hi);
    const uint64_t k = G / B, a = k * B;
    uint64_t n = k * (hi - lo);
    if (a < N) {
      const uint64_t m = (N - a < B) ? N - a : B, off = G - a;
      piece(m, &lo, &hi);
      n += off <= lo ? 0 : (off >= hi ? hi - lo : off - lo);
    }
    return n;
  }
};
struct keep_shard {
  shard_map S;
  uint64_t row_base;
  bool operator()(uint64_t g) const { return S.member(row_base + g); }
};
}  // namespace

extern "C" int bear_shard_rows_count(uint64_t row_base, uint64_t file_rows, uint64_t total_rows, uint64_t batch_rows, int rank,
                                     int world, uint64_t *n_local_out) {
  if (!n_local_out || batch_rows == 0 || world < 1 || rank < 0 || rank >= world || row_base + file_rows > total_rows)
    return BEAR_ERR_INVALID_ARG;
  const shard_map S{batch_rows, total_rows, (uint64_t)world, (uint64_t)rank};
  *n_local_out = S.below(row_base + file_rows) - S.below(row_base);
  return BEAR_OK;
}

extern "C" int bear_parse_counts_tsv_shard(const char *path, int num_ds, int lag, uint64_t skip_lines, uint64_t row_base,
                                           uint64_t total_rows, uint64_t batch_rows, int rank, int world, uint64_t max_rows,
                                           char *kmers, uint32_t *counts, uint64_t *n_local_out, uint64_t *n_file_rows_out) {
  if (!path || !counts || !n_local_out || num_ds < 1 || lag < 0 || batch_rows == 0 || world < 1 || rank < 0 || rank >= world)
    return BEAR_ERR_INVALID_ARG;
  *n_local_out = 0;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  const char *base = f.data, *end = f.data + f.size;
  for (uint64_t k = 0; k < skip_lines && base < end; ++k) {   // header lines (dataloader.py:7 `header`)
    const char *nl = static_cast<const char *>(memchr(base, '\n', (size_t)(end - base)));
    base = nl ? nl + 1 : end;
  }
  const size_t body = (size_t)(end - base);
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *env = getenv("BEAR_PARSE_THREADS")) nt = (unsigned)atoi(env);
  if (nt < 1) nt = 1;
  if (nt > BEAR_MAX_HOST_THREADS) nt = BEAR_MAX_HOST_THREADS;
  if (body < (size_t)(1u << 20)) nt = 1;
  std::vector<const char *> cut(nt + 1);
  cut[0] = base;
  cut[nt] = end;
  for (unsigned k = 1; k < nt; ++k) {
    const char *p = base + (body / nt) * k;
    const char *nl = p < end ? static_cast<const char *>(memchr(p, '\n', (size_t)(end - p))) : nullptr;
    cut[k] = nl ? nl + 1 : end;
    if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
  }
  std::vector<uint64_t> first(nt + 1, 0), done(nt, 0);
  std::vector<int> status(nt, BEAR_OK);
  {
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k)
      th.emplace_back([&, k] {
        populate_read(cut[k], cut[k + 1]);
        first[k + 1] = count_lines(cut[k], cut[k + 1]);
      });
    for (auto &t : th) t.join();
  }
  for (unsigned k = 0; k < nt; ++k) first[k + 1] += first[k];
  const uint64_t file_rows = first[nt];
  if (n_file_rows_out) *n_file_rows_out = file_rows;
  if (row_base + file_rows > total_rows) return BEAR_ERR_INVALID_ARG;   // the caller's row count does not match the file
  const shard_map S{batch_rows, total_rows, (uint64_t)world, (uint64_t)rank};
  const uint64_t l0 = S.below(row_base);
  if (S.below(row_base + file_rows) - l0 > max_rows) return BEAR_ERR_INVALID_ARG;
  {
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; ++k)
      th.emplace_back([&, k] {
        const uint64_t out0 = S.below(row_base + first[k]) - l0, out1 = S.below(row_base + first[k + 1]) - l0;
        if (out0 >= out1) return;
        status[k] = parse_range<keep_shard>(cut[k], cut[k + 1], num_ds, lag, max_rows, out0, out1, kmers, counts, &done[k], first[k],
                                            keep_shard{S, row_base});
      });
    for (auto &t : th) t.join();
  }
  uint64_t total = 0;
  for (unsigned k = 0; k < nt; ++k) {
    if (status[k] != BEAR_OK) return status[k];
    total += done[k];
  }
  *n_local_out = total;
  return BEAR_OK;
}

// ------------------------------------------------------------------ binary cache of a parsed table
// The reference decodes the text of every file again in every process and keeps the tensors in host RAM
// (dataloader.py:47-48, `.cache()`); at 1e9 rows the text is 60-80 GB and decoding it is the first-epoch cost.
// The cache is the parsed table as it is uploaded: header, k-mer bytes, planar uint32 slabs.
namespace {
struct cache_header {
  char magic[8];          // "BEARCT01"
  uint64_t n_rows;
  uint32_t lag, num_ds;
  uint64_t src_size;      // size and mtime of the text file the cache was parsed from (staleness check)
  int64_t src_mtime_ns;
  uint64_t kmers_offset, counts_offset;
  uint64_t reserved[1];
};
static_assert(sizeof(cache_header) == 64, "cache header is 64 bytes");
const char CACHE_MAGIC[8] = {'B', 'E', 'A', 'R', 'C', 'T', '0', '1'};

bool write_all(int fd, const void *buf, size_t n) {
  const char *p = static_cast<const char *>(buf);
  while (n) {
    ssize_t w = ::write(fd, p, n > (1u << 30) ? (1u << 30) : n);
    if (w <= 0) return false;
    p += w;
    n -= (size_t)w;
  }
  return true;
}
bool read_all(int fd, void *buf, size_t n, uint64_t off) {
  char *p = static_cast<char *>(buf);
  while (n) {
    ssize_t r = ::pread(fd, p, n > (1u << 30) ? (1u << 30) : n, (off_t)off);
    if (r <= 0) return false;
    p += r;
    off += (uint64_t)r;
    n -= (size_t)r;
  }
  return true;
}
int read_header(int fd, cache_header *h) {
  if (!read_all(fd, h, sizeof(*h), 0)) return BEAR_ERR_PARSE;
  if (memcmp(h->magic, CACHE_MAGIC, 8) != 0) return BEAR_ERR_PARSE;
  struct stat st;
  if (fstat(fd, &st) != 0) return BEAR_ERR_IO;
  const uint64_t need = h->counts_offset + (uint64_t)h->num_ds * h->n_rows * BEAR_ROW_WIDTH * 4;
  if (h->kmers_offset < sizeof(*h) || h->counts_offset < h->kmers_offset + h->n_rows * h->lag || (uint64_t)st.st_size < need)
    return BEAR_ERR_PARSE;
  return BEAR_OK;
}
}  // namespace

extern "C" int bear_stat_source(const char *path, uint64_t *size_out, int64_t *mtime_ns_out) {
  if (!path || !size_out || !mtime_ns_out) return BEAR_ERR_INVALID_ARG;
  struct stat st;
  if (stat(path, &st) != 0) return BEAR_ERR_IO;
  *size_out = (uint64_t)st.st_size;
  *mtime_ns_out = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
  return BEAR_OK;
}

extern "C" int bear_cache_write(const char *path, const char *kmers, const uint32_t *counts, uint64_t n_rows, int lag,
                                int num_ds, uint64_t src_size, int64_t src_mtime_ns) {
  if (!path || !counts || (!kmers && lag > 0 && n_rows) || lag < 0 || num_ds < 1) return BEAR_ERR_INVALID_ARG;
  cache_header h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, CACHE_MAGIC, 8);
  h.n_rows = n_rows;
  h.lag = (uint32_t)lag;
  h.num_ds = (uint32_t)num_ds;
  h.src_size = src_size;
  h.src_mtime_ns = src_mtime_ns;
  h.kmers_offset = sizeof(h);
  h.counts_offset = (h.kmers_offset + n_rows * (uint64_t)lag + 63) & ~63ull;
  // write to a temporary name and rename: a reader never sees a half-written cache
  char tmp[4096];
  if (snprintf(tmp, sizeof(tmp), "%s.tmp.%d", path, (int)getpid()) >= (int)sizeof(tmp)) return BEAR_ERR_INVALID_ARG;
  int fd = ::open(tmp, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return BEAR_ERR_IO;
  static const char zeros[64] = {0};
  bool ok = write_all(fd, &h, sizeof(h)) && write_all(fd, kmers, n_rows * (uint64_t)lag) &&
            write_all(fd, zeros, h.counts_offset - (h.kmers_offset + n_rows * (uint64_t)lag)) &&
            write_all(fd, counts, (uint64_t)num_ds * n_rows * BEAR_ROW_WIDTH * 4);
  ok = (::close(fd) == 0) && ok;
  if (!ok || ::rename(tmp, path) != 0) {
    ::unlink(tmp);
    return BEAR_ERR_IO;
  }
  return BEAR_OK;
}

extern "C" int bear_cache_info(const char *path, uint64_t *n_rows, int *lag, int *num_ds, uint64_t *src_size,
                               int64_t *src_mtime_ns) {
  if (!path) return BEAR_ERR_INVALID_ARG;
  int fd = ::open(path, O_RDONLY);
  if (fd < 0) return BEAR_ERR_IO;
  cache_header h;
  int st = read_header(fd, &h);
  ::close(fd);
  if (st != BEAR_OK) return st;
  if (n_rows) *n_rows = h.n_rows;
  if (lag) *lag = (int)h.lag;
  if (num_ds) *num_ds = (int)h.num_ds;
  if (src_size) *src_size = h.src_size;
  if (src_mtime_ns) *src_mtime_ns = h.src_mtime_ns;
  return BEAR_OK;
}

extern "C" int bear_cache_read(const char *path, uint64_t row0, uint64_t n_rows, char *kmers, uint32_t *counts) {
  if (!path || !counts) return BEAR_ERR_INVALID_ARG;
  int fd = ::open(path, O_RDONLY);
  if (fd < 0) return BEAR_ERR_IO;
  cache_header h;
  int st = read_header(fd, &h);
  if (st == BEAR_OK && (row0 > h.n_rows || n_rows > h.n_rows - row0)) st = BEAR_ERR_INVALID_ARG;
  if (st == BEAR_OK && kmers && !read_all(fd, kmers, n_rows * h.lag, h.kmers_offset + row0 * h.lag)) st = BEAR_ERR_IO;
  for (uint32_t d = 0; st == BEAR_OK && d < h.num_ds; ++d)
    if (!read_all(fd, counts + (uint64_t)d * n_rows * BEAR_ROW_WIDTH, n_rows * BEAR_ROW_WIDTH * 4,
                  h.counts_offset + ((uint64_t)d * h.n_rows + row0) * BEAR_ROW_WIDTH * 4))
      st = BEAR_ERR_IO;
  ::close(fd);
  return st;
}

// ------------------------------------------------------------------ count-table writer (summarize.py:429-449 row format)
// Rows row_begin, row_begin + row_step, ... of a planar table:  kmer \t [[g0 A,C,G,T,$],[g1 ...],...] \n
namespace {
inline int dec_digits(uint32_t v) {
  return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7
         : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
// bytes of one row: kmer \t [ [a,b,c,d,e] , [..] ... ] \n
inline size_t row_bytes(const uint32_t *counts, uint64_t n_rows, uint64_t r, int lag, int num_ds) {
  size_t n = (size_t)lag + 1 + 1 + 1 + 1;                       // kmer, tab, outer [ ... ], newline
  for (int d = 0; d < num_ds; ++d) {
    const uint32_t *c = counts + ((uint64_t)d * n_rows + r) * BEAR_ROW_WIDTH;
    n += 2 + (BEAR_ROW_WIDTH - 1) + (d ? 1 : 0);                // [ ] , the commas inside, the comma in front of a later group
    for (int b = 0; b < BEAR_ROW_WIDTH; ++b) n += (size_t)dec_digits(c[b]);
  }
  return n;
}
inline char *format_row(char *o, const char *kmers, const uint32_t *counts, uint64_t n_rows, uint64_t r, int lag, int num_ds) {
  memcpy(o, kmers + r * (uint64_t)lag, (size_t)lag);
  o += lag;
  *o++ = '\t';
  *o++ = '[';
  for (int d = 0; d < num_ds; ++d) {
    if (d) *o++ = ',';
    *o++ = '[';
    const uint32_t *c = counts + ((uint64_t)d * n_rows + r) * BEAR_ROW_WIDTH;
    for (int b = 0; b < BEAR_ROW_WIDTH; ++b) {
      if (b) *o++ = ',';
      uint32_t v = c[b];
      const int k = dec_digits(v);
      for (int j = k - 1; j >= 0; --j) {
        o[j] = (char)('0' + v % 10);
        v /= 10;
      }
      o += k;
    }
    *o++ = ']';
  }
  *o++ = ']';
  *o++ = '\n';
  return o;
}
}  // namespace

// Two passes over the selected rows, both cut into one contiguous range per hardware thread: the byte length of every range
// (digit counting), a prefix sum, then each thread formats its range into 4 MiB pieces and writes them at its own file offset
// (pwrite) -- the text of a 1e8-row, three-column table (5 GB) in seconds instead of the half minute of one thread.
extern "C" int bear_write_counts_tsv(const char *path, const char *kmers, const uint32_t *counts, uint64_t n_rows, int lag,
                                     int num_ds, uint64_t row_begin, uint64_t row_step, int append) {
  if (!path || !counts || (!kmers && lag > 0 && n_rows) || lag < 0 || num_ds < 1 || row_step == 0) return BEAR_ERR_INVALID_ARG;
  const int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0644);
  if (fd < 0) return BEAR_ERR_IO;
  off_t base = 0;
  if (append) {
    base = lseek(fd, 0, SEEK_END);
    if (base < 0) {
      close(fd);
      return BEAR_ERR_IO;
    }
  }
  const uint64_t n_sel = row_begin < n_rows ? (n_rows - row_begin + row_step - 1) / row_step : 0;   // rows row_begin + i row_step
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > BEAR_MAX_HOST_THREADS) nt = BEAR_MAX_HOST_THREADS;
  if ((uint64_t)nt > (n_sel + 65535) / 65536) nt = (unsigned)((n_sel + 65535) / 65536);   // at least 64 Ki rows per thread
  if (nt == 0) nt = 1;
  std::vector<uint64_t> first(nt + 1), bytes(nt + 1, 0);
  for (unsigned t = 0; t <= nt; ++t) first[t] = n_sel * t / nt;
  auto measure = [&](unsigned t) {
    size_t n = 0;
    for (uint64_t i = first[t]; i < first[t + 1]; ++i) n += row_bytes(counts, n_rows, row_begin + i * row_step, lag, num_ds);
    bytes[t + 1] = n;
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(measure, t);
    measure(0);
    for (auto &x : th) x.join();
  }
  for (unsigned t = 0; t < nt; ++t) bytes[t + 1] += bytes[t];
  std::vector<int> status(nt, BEAR_OK);
  auto emit = [&](unsigned t) {
    static const size_t PIECE = 4u << 20;
    const size_t slack = 64 + (size_t)lag + (size_t)num_ds * 64;
    char *buf = static_cast<char *>(malloc(PIECE + slack));
    if (!buf) {
      status[t] = BEAR_ERR_NOMEM;
      return;
    }
    off_t at = base + (off_t)bytes[t];
    char *o = buf;
    auto flush = [&]() {
      size_t done = 0, n = (size_t)(o - buf);
      while (done < n) {
        const ssize_t w = pwrite(fd, buf + done, n - done, at + (off_t)done);
        if (w <= 0) {
          status[t] = BEAR_ERR_IO;
          return;
        }
        done += (size_t)w;
      }
      at += (off_t)n;
      o = buf;
    };
    for (uint64_t i = first[t]; i < first[t + 1] && status[t] == BEAR_OK; ++i) {
      o = format_row(o, kmers, counts, n_rows, row_begin + i * row_step, lag, num_ds);
      if ((size_t)(o - buf) >= PIECE) flush();
    }
    if (status[t] == BEAR_OK && o != buf) flush();
    if (status[t] == BEAR_OK && at != base + (off_t)bytes[t + 1]) status[t] = BEAR_ERR_IO;   // the two passes disagree: never silently
    free(buf);
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(emit, t);
    emit(0);
    for (auto &x : th) x.join();
  }
  int st = BEAR_OK;
  for (unsigned t = 0; t < nt; ++t)
    if (status[t] != BEAR_OK) st = status[t];
  if (close(fd) != 0 && st == BEAR_OK) st = BEAR_ERR_IO;
  return st;
}

// ------------------------------------------------------------------ FASTA / FASTQ -> device code text (summarize path)
// Replaces summarize.py's stage 1 readers (Biopython SimpleFastaParser / FastqGeneralIterator, summarize.py:96-100) for
// the device pipeline: per sequence  5 (start marker), letters A,C,G,T -> 0..3 (any other character 6), 4 (stop); with
// `reverse` every sequence is followed by its reverse complement (summarize.py:202-207).  Two calls: size, then fill.
namespace {
inline uint8_t letter_code(unsigned char ch) {
  switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 6;
  }
}

// Calls emit(begin, end) for every sequence line-segment and flush() at the end of each record.
template <class Seg, class End>
int walk_fastx(const mapped_file &f, int fastq, Seg seg, End end_record) {
  const char *p = f.data, *end = f.data + f.size;
  auto next_line = [&](const char *&b, const char *&e) {
    if (p >= end) return false;
    const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    b = p;
    e = nl ? nl : end;
    p = e + 1;
    if (e > b && e[-1] == '\r') --e;
    return true;
  };
  const char *b, *e;
  if (fastq) {
    while (next_line(b, e)) {
      if (b == e) continue;
      if (*b != '@') return BEAR_ERR_PARSE;
      if (!next_line(b, e)) return BEAR_ERR_PARSE;
      seg(b, e);
      end_record();
      const char *b2, *e2;
      if (!next_line(b2, e2) || b2 == e2 || *b2 != '+') return BEAR_ERR_PARSE;
      if (!next_line(b2, e2)) return BEAR_ERR_PARSE;
    }
  } else {
    bool open = false;
    while (next_line(b, e)) {
      if (b < e && *b == '>') {
        if (open) end_record();
        open = true;
      } else if (open) {
        while (b < e && (*b == ' ' || *b == '\t')) ++b;
        while (e > b && (e[-1] == ' ' || e[-1] == '\t')) --e;
        seg(b, e);
      }
    }
    if (open) end_record();
  }
  return BEAR_OK;
}
}  // namespace

extern "C" int bear_fastx_size(const char *path, int fastq, int reverse, uint64_t *n_pos_out, uint64_t *n_seqs_out) {
  if (!path || !n_pos_out) return BEAR_ERR_INVALID_ARG;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  uint64_t letters = 0, seqs = 0;
  st = walk_fastx(f, fastq, [&](const char *b, const char *e) { letters += (uint64_t)(e - b); }, [&] { ++seqs; });
  if (st != BEAR_OK) return st;
  const uint64_t mult = reverse ? 2 : 1;
  *n_pos_out = mult * (letters + 2 * seqs);
  if (n_seqs_out) *n_seqs_out = mult * seqs;
  return BEAR_OK;
}

extern "C" int bear_fastx_encode(const char *path, int fastq, int reverse, int group, uint64_t capacity, uint8_t *text,
                                 uint8_t *group_out, uint64_t *n_pos_out) {
  if (!path || !text || !n_pos_out || group < 0 || group > 254) return BEAR_ERR_INVALID_ARG;
  mapped_file f;
  int st = f.open_ro(path);
  if (st != BEAR_OK) return st;
  uint64_t pos = 0, start = 0;
  bool overflow = false, in_seq = false;
  auto put = [&](uint8_t v) {
    if (pos < capacity) text[pos] = v;
    else overflow = true;
    ++pos;
  };
  st = walk_fastx(
      f, fastq,
      [&](const char *b, const char *e) {
        if (!in_seq) {
          start = pos;
          put(5);
          in_seq = true;
        }
        for (const char *q = b; q < e; ++q) put(letter_code((unsigned char)*q));
      },
      [&] {
        if (!in_seq) {   // empty record: start + stop
          start = pos;
          put(5);
        }
        put(4);
        in_seq = false;
        if (reverse && !overflow) {
          const uint64_t first = start + 1, last = pos - 1;   // letters in [first, last)
          put(5);
          for (uint64_t q = last; q > first; --q) {
            const uint8_t c = text[q - 1];
            put(c < 4 ? (uint8_t)(3 - c) : c);
          }
          put(4);
        }
      });
  if (st != BEAR_OK) return st;
  if (overflow) return BEAR_ERR_INVALID_ARG;
  if (group_out) memset(group_out, group, (size_t)pos);
  *n_pos_out = pos;
  return BEAR_OK;
}

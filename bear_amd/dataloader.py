"""Count-table ingestion: host mirror of ``bear_model/dataloader.py``.

``dataloader(file, alphabet, batch_size, num_ds, cache, header, n_par, dtype)`` keeps the reference
signature (dataloader.py:6-7) and, iterated, yields the reference's batches
``(kmers [B] bytes, counts [B, num_ds, A+1])`` in file order with a short last batch
(dataloader.py:31-37).  Underneath, the file is parsed ONCE by the C++ reader in libbear_hip.so
(``bear_parse_counts_tsv``: mmap + in-place integer parse, replacing CsvDataset + JSON decoding,
dataloader.py:35-46) into k-mer bytes and planar ``uint32 [num_ds, N, 5]`` slabs, which
``bear_net.train`` / ``bear_ref.train`` upload once and keep resident.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, core, kernels


class CountDataset:
    """Parsed count table: ``kmers`` uint8 [rows, lag] (ASCII), ``counts`` uint32 [num_ds, rows, A+1].

    ``shard = (rank, world)``: the arrays hold only this rank's rows -- of every batch ``[a, b)`` of the table the contiguous
    piece ``dist.shard_rows(b - a, rank, world)``, pieces in table order -- while ``num_rows`` / ``batch_bounds`` keep describing
    the whole table (``total_rows`` rows).  That is the input sharding of ``strategy.experimental_distribute_dataset``
    (bear_net.py:273) done at load time, so a rank never decodes or holds another rank's rows."""

    def __init__(self, kmers, counts, alphabet, batch_size, dtype=torch.float64, repeats=1, shuffle_seed=None, shard=None,
                 total_rows=None):
        self.shuffle_seed = shuffle_seed   # None: file order; else rows are permuted on the device at upload
        self.kmers = kmers
        self.counts = counts
        self.alphabet = alphabet
        self.batch_size = int(batch_size)
        self.dtype = dtype
        self.repeats = int(repeats)
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        if shard is not None and total_rows is None:
            raise ValueError("a sharded CountDataset needs total_rows (the row count of the whole table)")
        if shard is not None and not (0 <= self.shard[0] < self.shard[1]):
            raise ValueError(f"shard = (rank, world) with 0 <= rank < world, got {self.shard}")
        self.total_rows = int(total_rows) if shard is not None else None
        if self.shard is not None and self.local_rows != sum(hi - lo for lo, hi in self._pieces()):
            raise ValueError("sharded CountDataset: the arrays do not hold exactly this rank's rows")
        self._device_cache = {}

    # ---- reference-shaped view -----------------------------------------------------------------
    @property
    def local_rows(self):
        return self.counts.shape[1]

    @property
    def num_rows(self):
        return self.total_rows if self.shard is not None else self.counts.shape[1]

    @property
    def num_ds(self):
        return self.counts.shape[0]

    @property
    def lag(self):
        return self.kmers.shape[1]

    def batch_bounds(self):
        """Row ranges of the batches of one epoch (last one short: no drop_remainder, dataloader.py:37)."""
        B, N = self.batch_size, self.num_rows
        return [(a, min(a + B, N)) for a in range(0, N, B)]

    def _pieces(self):
        from . import dist
        rank, world = self.shard
        out = []
        for a, b in self.batch_bounds():
            lo, hi = dist.shard_rows(b - a, rank, world)
            out.append((a + lo, a + hi))
        return out

    def rank_pieces(self, rank, world):
        """Per batch: (global row range [g0, g1) of `rank`'s piece, offset of g0 in this dataset's arrays)."""
        from . import dist
        if self.shard is not None and self.shard != (rank, world):
            raise ValueError(f"dataset was loaded for rank/world {self.shard}, asked for {(rank, world)}")
        out, off = [], 0
        for a, b in self.batch_bounds():
            lo, hi = dist.shard_rows(b - a, rank, world)
            out.append((a + lo, a + hi, off if self.shard is not None else a + lo))
            off += hi - lo
        return out

    def deal_by_kmer(self, rank, world):
        """This rank's rows of every batch when the batches are dealt to the ranks BY K-MER RANGE instead of by contiguous row
        pieces (``KmerDealtDataset``).  Needs the whole table (every rank parses it and keeps its range: no exchange); the
        cut points follow from the batch's own histogram of leading letters, so every rank computes the same ones."""
        if self.shard is not None:
            raise ValueError("deal_by_kmer starts from the whole table (load it unsharded)")
        if self.shuffle_seed is not None:
            raise ValueError("a table shuffled on the device cannot be dealt by k-mer")
        if world <= 1:
            return self
        keep, index, piece_rows = [], [], []
        for a, b in self.batch_bounds():
            key = kmer_deal_keys(self.kmers[a:b], self.alphabet)
            hist = np.bincount(key, minlength=6 ** min(self.lag, KMER_DEAL_LETTERS))
            cum = np.concatenate([[0], np.cumsum(hist)])
            # rank r takes the bins [cut[r], cut[r + 1]): the first bin whose rows reach past r / world of the batch opens rank r's range
            targets = [(b - a) * r // world for r in range(world + 1)]
            cut = [int(np.searchsorted(cum, t, side="left")) for t in targets]
            cut[0], cut[-1] = 0, len(hist)
            mine = np.nonzero((key >= cut[rank]) & (key < cut[rank + 1]))[0]
            keep.append(a + mine)
            index.append(mine)
            piece_rows.append(len(mine))
        rows = np.concatenate(keep) if keep else np.zeros(0, dtype=np.int64)
        return KmerDealtDataset(np.ascontiguousarray(self.kmers[rows]), np.ascontiguousarray(self.counts[:, rows]),
                                np.concatenate(index) if index else np.zeros(0, dtype=np.int32), piece_rows, self.alphabet,
                                self.batch_size, self.dtype, self.repeats, rank, world, self.num_rows)

    def __len__(self):
        return len(self.batch_bounds()) * self.repeats

    def __iter__(self):
        """The reference's batches ``(kmers [B], counts [B, num_ds, A+1])`` in file order; a sharded dataset yields this rank's
        piece of every batch."""
        np_dtype = np.float64 if self.dtype == torch.float64 else np.float32
        if self.shard is not None:
            bounds = [(off, off + g1 - g0) for g0, g1, off in self.rank_pieces(*self.shard)]
        else:
            bounds = self.batch_bounds()
        for _ in range(self.repeats):
            for a, b in bounds:
                km = np.array([bytes(r) for r in self.kmers[a:b]])
                yield km, torch.from_numpy(self.counts[:, a:b].transpose(1, 0, 2).astype(np_dtype))

    def map(self, fn):
        """tf.data ``.map(fn)`` as the reference's callers use it (``data.map(lambda kmers, counts: counts)`` in front of
        ``bmm_likelihood``, tests/test_dataloader.py:48): a lazily mapped view that remembers its source table."""
        return MappedDataset(self, fn)

    def _like(self, **kw):
        args = dict(alphabet=self.alphabet, batch_size=self.batch_size, dtype=self.dtype, repeats=self.repeats,
                    shuffle_seed=self.shuffle_seed, shard=self.shard, total_rows=self.total_rows)
        args.update(kw)
        return CountDataset(self.kmers, self.counts, **args)

    def repeat(self, epochs):
        """tf.data ``.repeat(epochs)`` (models/train_bear_net.py:88)."""
        return self._like(repeats=self.repeats * int(epochs))

    def shuffle(self, seed):
        """The `shuf` step of docs/usage.rst:191-200 without rewriting the file: training and evaluation see the rows
        in the order ``perm_seed`` (one gather pass on the device at upload, ``bear_shuffle_rows``).  Iterating the
        dataset on the host still yields file order."""
        if self.shard is not None:
            raise ValueError("the device shuffle permutes whole columns: load the table unsharded (or pre-shuffle the file, "
                             "docs/usage.rst:191-200) to combine it with row sharding")
        return self._like(shuffle_seed=int(seed))

    # ---- packed device view --------------------------------------------------------------------
    def codes(self):
        return core.encode_kmers(self.kmers, self.alphabet)

    def device_column(self, ds_loc, device, rows=None):
        """uint32 [n, 5] slab of one dataset column on `device` (int32 storage), cached."""
        key = (int(ds_loc), str(device), rows)
        t = self._device_cache.get(key)
        if t is None:
            a, b = rows if rows is not None else (0, self.local_rows)
            t = torch.from_numpy(np.ascontiguousarray(self.counts[ds_loc, a:b]).view(np.int32)).to(device)
            self._device_cache[key] = t
        return t


class KmerDealtDataset(CountDataset):
    """This rank's rows of every batch, dealt BY K-MER RANGE (``CountDataset.deal_by_kmer``): of every batch ``[a, b)`` the rows whose
    leading letters fall into this rank's range of the batch's k-mers -- ranges in k-mer order, cut so that the ranks hold the
    same number of rows to within one bin of leading letters.  ``row_index`` int32 [local rows]: each row's number inside its
    batch (the rows of a piece are not a contiguous row range of the table: the key of the evaluation's tie-breaking noise
    travels with them).  The sums of a step do not depend on WHICH rows a rank holds; what the fused AR-function kernels gain is
    density: a rank's piece of a k-mer-sorted batch keeps the whole batch's density of distinct prefixes and windows, where a
    contiguous row piece of a pre-shuffled table is a random 1 / world of its k-mers (kernels_cnn.h prefix levels, window tables;
    kernels_linear.h paired lists)."""

    def __init__(self, kmers, counts, row_index, piece_rows, alphabet, batch_size, dtype, repeats, rank, world, total_rows):
        self.row_index = np.ascontiguousarray(row_index, dtype=np.int32)
        self.piece_rows = [int(x) for x in piece_rows]          # rows of this rank in each batch
        super().__init__(kmers, counts, alphabet, batch_size, dtype, repeats=repeats, shard=(rank, world), total_rows=total_rows)

    def _pieces(self):
        return [(a, a + c) for (a, b), c in zip(self.batch_bounds(), self.piece_rows)]

    def rank_pieces(self, rank, world):
        """Per batch: (batch start a, a + rows of this rank's piece, offset of the piece in this dataset's arrays): the piece's
        rows are ``a + row_index[offset : offset + rows]``."""
        if self.shard != (rank, world):
            raise ValueError(f"dataset was dealt for rank/world {self.shard}, asked for {(rank, world)}")
        out, off = [], 0
        for (a, b), c in zip(self.batch_bounds(), self.piece_rows):
            out.append((a, a + c, off))
            off += c
        return out

    def _like(self, **kw):
        args = dict(alphabet=self.alphabet, batch_size=self.batch_size, dtype=self.dtype, repeats=self.repeats)
        if kw.get("shuffle_seed") is not None:
            raise ValueError("a dealt table cannot be shuffled on the device")
        args.update({k: v for k, v in kw.items() if k in args})
        return KmerDealtDataset(self.kmers, self.counts, self.row_index, self.piece_rows, args["alphabet"], args["batch_size"], args["dtype"],
                                args["repeats"], self.shard[0], self.shard[1], self.total_rows)


KMER_DEAL_LETTERS = 6      # leading letters that decide a row's rank (6^6 = 46 656 bins per batch)


def kmer_deal_keys(kmers, alphabet):
    """The dealing key of every row: its leading KMER_DEAL_LETTERS letters as a number in the k-mer order of the device sort
    (``bear_kmer_order_u64``: first letter most significant, letters in alphabet order, the start symbol behind them, anything
    else last)."""
    codes = core.encode_kmers(kmers[:, :KMER_DEAL_LETTERS], alphabet).astype(np.int64)
    codes[codes < 0] = 5
    key = np.zeros(kmers.shape[0], dtype=np.int64)
    for l in range(codes.shape[1]):
        key = key * 6 + codes[:, l]
    return key


class MappedDataset:
    """``CountDataset.map(fn)``: iterating yields ``fn(kmers, counts)`` per batch; ``source`` is the table."""

    def __init__(self, source, fn):
        self.source, self.fn = source, fn

    def __iter__(self):
        for kmers, counts in self.source:
            yield self.fn(kmers, counts)


class DeviceCountDataset(CountDataset):
    """A count table that already lives in HBM (e.g. built by ``bear_amd.summarize`` on the device): ``kmers_dev`` uint8
    [N, lag] and ``counts_dev`` int32-storage uint32 [num_ds, N, 5].  Training and evaluation use the device tensors
    directly (no upload); the host views ``kmers`` / ``counts`` are downloaded on first use."""

    def __init__(self, kmers_dev, counts_dev, alphabet, batch_size, dtype=torch.float64, repeats=1, shuffle_seed=None):
        self.kmers_dev, self.counts_dev = kmers_dev.contiguous(), counts_dev.contiguous()
        self._host = None
        self.shuffle_seed = shuffle_seed
        self.alphabet, self.batch_size, self.dtype, self.repeats = alphabet, int(batch_size), dtype, int(repeats)
        self.shard, self.total_rows = None, None      # a device-built table is whole; training slices this rank's rows from it
        self._device_cache = {}

    def _download(self):
        if self._host is None:
            self._host = (self.kmers_dev.cpu().numpy(), self.counts_dev.cpu().numpy().view(np.uint32))
        return self._host

    kmers = property(lambda self: self._download()[0])
    counts = property(lambda self: self._download()[1])
    num_rows = property(lambda self: self.counts_dev.shape[1])
    local_rows = property(lambda self: self.counts_dev.shape[1])
    num_ds = property(lambda self: self.counts_dev.shape[0])
    lag = property(lambda self: self.kmers_dev.shape[1])

    def repeat(self, epochs):
        return DeviceCountDataset(self.kmers_dev, self.counts_dev, self.alphabet, self.batch_size, self.dtype,
                                  self.repeats * int(epochs), self.shuffle_seed)

    def shuffle(self, seed):
        return DeviceCountDataset(self.kmers_dev, self.counts_dev, self.alphabet, self.batch_size, self.dtype, self.repeats, int(seed))

    def device_column(self, ds_loc, device, rows=None):
        a, b = rows if rows is not None else (0, self.num_rows)
        return self.counts_dev[ds_loc, a:b].to(device)


def concatenate(datasets):
    """Several count files of one table (models/train_bear_net.py:79-87 interleaves them; rows are
    independent and pre-shuffled, so concatenation is an equivalent batch stream).  Sharded parts must have been loaded
    with consecutive ``row_base`` and the same ``total_rows``: their local rows concatenate to this rank's rows of the table."""
    d0 = datasets[0]
    if any(d.shard != d0.shard or d.total_rows != d0.total_rows for d in datasets):
        raise ValueError("cannot concatenate count tables with different sharding")
    parts = [d for d in datasets if isinstance(d, _ShardPart)]
    if parts:
        if len(parts) != len(datasets):
            raise ValueError("cannot mix whole sharded tables and per-file parts")
        at = 0
        for d in parts:          # the parts must tile the table: each file starts where the previous one ended
            if d.row_base != at:
                raise ValueError(f"sharded parts are not consecutive: a part starts at global row {d.row_base}, expected {at}")
            at += d.file_rows
        if at != d0.total_rows:
            raise ValueError(f"sharded parts cover {at} rows of a table of {d0.total_rows}")
    return CountDataset(np.concatenate([d.kmers for d in datasets]), np.concatenate([d.counts for d in datasets], axis=1),
                        d0.alphabet, d0.batch_size, d0.dtype, shard=d0.shard, total_rows=d0.total_rows)


def _sniff_lag(file, header, delim):
    with open(file, "rb") as fh:
        if header:
            fh.readline()
        for line in fh:
            if line.strip():
                return line.split(delim)[0].__len__()
    return 0


def cache_path_for(file, binary_cache):
    """Where the binary cache of ``file`` lives: next to it (``True``), or inside a directory."""
    if binary_cache is True:
        return str(file) + ".bearcache"
    return os.path.join(str(binary_cache), os.path.basename(str(file)) + ".bearcache")


def _load_cache(path, file, num_ds, rows=None):
    """The parsed table from a valid, fresh binary cache, or None."""
    L = _lib.lib()
    n, lag, nds, ssz, smt = ctypes.c_uint64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64(), ctypes.c_int64()
    if not os.path.exists(path) or L.bear_cache_info(path.encode(), ctypes.byref(n), ctypes.byref(lag), ctypes.byref(nds),
                                                     ctypes.byref(ssz), ctypes.byref(smt)) != 0:
        return None
    fsz, fmt = ctypes.c_uint64(), ctypes.c_int64()
    if L.bear_stat_source(str(file).encode(), ctypes.byref(fsz), ctypes.byref(fmt)) != 0:
        return None
    if (fsz.value, fmt.value) != (ssz.value, smt.value) or nds.value != num_ds:
        return None      # stale (source rewritten) or parsed with another num_ds
    a, b = rows if rows is not None else (0, n.value)
    kmers = np.zeros((b - a, lag.value), dtype=np.uint8)
    counts = np.zeros((num_ds, b - a, 5), dtype=np.uint32)
    _lib.check(L.bear_cache_read(path.encode(), a, b - a, kmers.ctypes.data, counts.ctypes.data), "bear_cache_read")
    return kmers, counts


def _cache_meta(path, file, num_ds):
    """(n_rows, lag) of a valid, fresh binary cache of `file`, or None."""
    L = _lib.lib()
    n, lag, nds, ssz, smt = ctypes.c_uint64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64(), ctypes.c_int64()
    if not os.path.exists(path) or L.bear_cache_info(path.encode(), ctypes.byref(n), ctypes.byref(lag), ctypes.byref(nds),
                                                     ctypes.byref(ssz), ctypes.byref(smt)) != 0:
        return None
    fsz, fmt = ctypes.c_uint64(), ctypes.c_int64()
    if L.bear_stat_source(str(file).encode(), ctypes.byref(fsz), ctypes.byref(fmt)) != 0:
        return None
    if (fsz.value, fmt.value) != (ssz.value, smt.value) or nds.value != num_ds:
        return None
    return n.value, lag.value


def count_rows(file, header=False):
    """Number of table rows of a count file (non-empty lines, minus the header line)."""
    n = ctypes.c_uint64()
    _lib.check(_lib.lib().bear_count_rows(str(file).encode(), ctypes.byref(n)), "bear_count_rows")
    return n.value - (1 if header and n.value else 0)


def count_newlines(file):
    """``wc -l`` of a file on all host threads (the reference's ``num_kmers``, models/train_bear_net.py:52-55)."""
    n = ctypes.c_uint64()
    _lib.check(_lib.lib().bear_count_newlines(str(file).encode(), ctypes.byref(n)), "bear_count_newlines")
    return n.value


def dataloader(file, alphabet, batch_size, num_ds, cache=True, header=False, n_par=1, dtype=torch.float64,
               binary_cache=None, shard=None, row_base=0, total_rows=None):
    """dataloader.py:6-50.  ``cache`` / ``n_par`` are accepted for signature compatibility: the table
    is always parsed once and kept.  ``binary_cache`` (``True``: next to the file; or a directory; default: the
    ``BEAR_AMD_CACHE_DIR`` environment variable) keeps the parsed table on disk so later runs skip the text.

    ``shard=(rank, world)``: load only this rank's rows (see ``CountDataset``); ``shard="auto"`` takes them from the initialised
    ``torch.distributed`` group.  A sharded load READS a fresh binary cache (ranged reads) but never writes one -- the cache holds a
    whole table and no rank of a sharded run has it; build it once with an unsharded ``dataloader(..., binary_cache=...)`` call
    (``models/_driver.py`` does that on rank 0 when ``[data] binary_cache`` is set and the table fits its host memory).  ``row_base`` / ``total_rows`` place the file inside a table made of several files (the batches
    -- and so the pieces -- are cut on the whole table); by default the file is the table."""
    L = _lib.lib()
    A1 = len(core.alphabets_tf[alphabet])
    if A1 != 5:
        raise NotImplementedError("the HIP kernels are built for 4-letter alphabets (+ stop): dna / rna")
    deal_kmer = shard == "kmer"
    if shard in ("auto", "kmer"):
        from . import dist
        shard = dist.world() if dist.world()[1] > 1 else None
    if deal_kmer and shard is not None:
        # rows dealt to the ranks by k-mer range (KmerDealtDataset): every rank parses the whole table and keeps its range
        whole = dataloader(file, alphabet, batch_size, num_ds, cache=cache, header=header, n_par=n_par, dtype=dtype,
                           binary_cache=binary_cache, shard=None)
        if row_base or total_rows is not None:
            raise ValueError("shard='kmer' deals one whole table: concatenate the files first")
        return whole.deal_by_kmer(*shard)
    if binary_cache is None:
        binary_cache = os.environ.get("BEAR_AMD_CACHE_DIR") or None
    if shard is not None:
        return _load_shard(file, alphabet, int(batch_size), int(num_ds), header, dtype, binary_cache, shard, int(row_base), total_rows)
    if binary_cache:
        hit = _load_cache(cache_path_for(file, binary_cache), file, num_ds)
        if hit is not None:
            return CountDataset(hit[0], hit[1], alphabet, batch_size, dtype)
    n_rows = count_rows(file, header)
    lag = _sniff_lag(file, header, b"\t")
    kmers = np.zeros((n_rows, lag), dtype=np.uint8)
    counts = np.zeros((num_ds, n_rows, A1), dtype=np.uint32)
    got = ctypes.c_uint64()
    if header:      # the sharded reader with one rank is the plain reader with a header line
        _lib.check(L.bear_parse_counts_tsv_shard(str(file).encode(), int(num_ds), int(lag), 1, 0, n_rows, max(n_rows, 1), 0, 1, n_rows,
                                                 kmers.ctypes.data, counts.ctypes.data, ctypes.byref(got), None), "bear_parse_counts_tsv_shard")
    else:
        _lib.check(L.bear_parse_counts_tsv(str(file).encode(), int(num_ds), int(lag), n_rows, kmers.ctypes.data,
                                           counts.ctypes.data, ctypes.byref(got)), "bear_parse_counts_tsv")
    if got.value != n_rows:
        raise RuntimeError(f"{file}: {got.value} rows parsed, {n_rows} counted (did the file change while it was read?)")
    if binary_cache:
        fsz, fmt = ctypes.c_uint64(), ctypes.c_int64()
        _lib.check(L.bear_stat_source(str(file).encode(), ctypes.byref(fsz), ctypes.byref(fmt)), "bear_stat_source")
        path = cache_path_for(file, binary_cache)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        _lib.check(L.bear_cache_write(path.encode(), kmers.ctypes.data, counts.ctypes.data, n_rows, int(lag), int(num_ds),
                                      fsz.value, fmt.value), "bear_cache_write")
    return CountDataset(kmers, counts, alphabet, batch_size, dtype)


def _load_shard(file, alphabet, batch_size, num_ds, header, dtype, binary_cache, shard, row_base, total_rows):
    """This rank's rows of `file` (global rows [row_base, row_base + file rows) of a table of total_rows rows): from a fresh
    binary cache when there is one (one ranged read per batch piece), else by the sharded text reader (every rank walks the
    file, decodes 1 / world of the lines)."""
    from . import dist
    L = _lib.lib()
    rank, world = int(shard[0]), int(shard[1])
    meta = _cache_meta(cache_path_for(file, binary_cache), file, num_ds) if binary_cache else None
    file_rows = meta[0] if meta else count_rows(file, header)
    if total_rows is None:
        total_rows = row_base + file_rows
    n_local = ctypes.c_uint64()
    _lib.check(L.bear_shard_rows_count(row_base, file_rows, int(total_rows), batch_size, rank, world, ctypes.byref(n_local)),
               "bear_shard_rows_count")
    lag = meta[1] if meta else _sniff_lag(file, header, b"\t")
    kmers = np.zeros((n_local.value, lag), dtype=np.uint8)
    counts = np.zeros((num_ds, n_local.value, 5), dtype=np.uint32)
    if meta:
        path, off = cache_path_for(file, binary_cache).encode(), 0
        for a in range((row_base // batch_size) * batch_size, row_base + file_rows, batch_size):
            b = min(a + batch_size, int(total_rows))
            lo, hi = dist.shard_rows(b - a, rank, world)
            g0, g1 = max(a + lo, row_base), min(a + hi, row_base + file_rows)      # the piece, clipped to this file
            if g1 <= g0:
                continue
            n = g1 - g0
            tmp = np.zeros((num_ds, n, 5), dtype=np.uint32)
            _lib.check(L.bear_cache_read(path, g0 - row_base, n, kmers[off:off + n].ctypes.data, tmp.ctypes.data), "bear_cache_read")
            counts[:, off:off + n] = tmp
            off += n
        if off != n_local.value:
            raise RuntimeError(f"{file}: the binary cache returned {off} rows for this rank, expected {n_local.value}")
    else:
        got, seen = ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(L.bear_parse_counts_tsv_shard(str(file).encode(), num_ds, int(lag), 1 if header else 0, row_base, int(total_rows),
                                                 batch_size, rank, world, n_local.value, kmers.ctypes.data, counts.ctypes.data,
                                                 ctypes.byref(got), ctypes.byref(seen)), "bear_parse_counts_tsv_shard")
        if got.value != n_local.value or seen.value != file_rows:
            raise RuntimeError(f"{file}: the sharded reader kept {got.value} of {seen.value} rows, expected {n_local.value} of {file_rows} "
                               "(did the file change while it was read?)")
    if row_base == 0 and file_rows == total_rows:
        return CountDataset(kmers, counts, alphabet, batch_size, dtype, shard=(rank, world), total_rows=total_rows)
    return _ShardPart(kmers, counts, alphabet, batch_size, dtype, (rank, world), int(total_rows), row_base, file_rows)


class _ShardPart:
    """One file's share of a sharded multi-file table: only ``concatenate`` makes a dataset of the parts."""

    def __init__(self, kmers, counts, alphabet, batch_size, dtype, shard, total_rows, row_base, file_rows):
        self.kmers, self.counts, self.alphabet, self.batch_size, self.dtype = kmers, counts, alphabet, batch_size, dtype
        self.shard, self.total_rows = shard, total_rows
        self.row_base, self.file_rows = int(row_base), int(file_rows)     # where the file sits in the table (checked by concatenate)


def sparse_dataloader(file, alphabet, batch_size, num_ds, cache=False, header=True, n_par=1, dtype=torch.float64):
    """dataloader.py:52-109: ``kmer; [[ds, col], ...]; [vals]`` rows with a header line."""
    A1 = len(core.alphabets_tf[alphabet])
    n_rows = count_rows(file, header)
    lag = 0
    with open(file, "rb") as fh:       # the k-mer length from the first data line
        if header:
            fh.readline()
        for line in fh:
            if line.strip():
                lag = len(line.split(b";", 1)[0].strip())
                break
    km = np.empty((n_rows, lag), dtype=np.uint8)
    counts = np.empty((num_ds, n_rows, A1), dtype=np.uint32)
    got = ctypes.c_uint64()
    st = _lib.lib().bear_parse_sparse_counts(os.fsencode(file), int(num_ds), A1, lag, 1 if header else 0, n_rows,
                                             km.ctypes.data, counts.ctypes.data, ctypes.byref(got))
    _lib.check(st, "bear_parse_sparse_counts")
    if got.value != n_rows:
        raise ValueError(f"{file}: {got.value} rows parsed, {n_rows} expected")
    return CountDataset(km, np.ascontiguousarray(counts), alphabet, batch_size, dtype)


def bmm_likelihood(data, alpha, dtype=torch.float64, device=None):
    """dataloader.py:120-147: BMM marginal ``sum_i lbeta(c_i + alpha) - lbeta(alpha)`` for every dataset
    column and every alpha -> [num_ds, len(alpha)].  One launch of ``bear_bmm_f64`` per column (all alphas
    in a single pass over the rows)."""
    if isinstance(data, MappedDataset):      # the reference passes data.map(lambda kmers, counts: counts)
        data = data.source
    if not isinstance(data, CountDataset):
        raise TypeError("bmm_likelihood expects the CountDataset returned by dataloader() (or its .map(...) view)")
    device = torch.device(device or "cuda")
    alpha = np.atleast_1d(np.asarray(alpha, dtype=np.float64))
    out = torch.zeros((data.num_ds, len(alpha)), dtype=torch.float64)
    for d in range(data.num_ds):
        col = data.device_column(d, device)
        for k in range(0, len(alpha), 64):
            out[d, k:k + 64] = kernels.bmm(col, alpha[k:k + 64]).cpu()
    return out.to(dtype)


def write_counts_tsv(path, kmers, counts):
    """Writes a dense count table in the summarize.py row format (summarize.py:429-449):
    ``kmer \\t [[g0 A,C,G,T,$],[g1 ...],...]``.  kmers: sequence of str/bytes or uint8 [N, lag];
    counts: integer array [num_ds, N, 5] (planar, as CountDataset.counts) or [N, num_ds, 5]."""
    counts = np.asarray(counts)
    if isinstance(kmers, np.ndarray) and kmers.dtype == np.uint8 and kmers.ndim == 2:
        km = np.ascontiguousarray(kmers)
    else:
        rows = [k if isinstance(k, bytes) else str(k).encode() for k in kmers]
        if len({len(r) for r in rows}) > 1:
            raise ValueError("write_counts_tsv: k-mers of one table have one length")
        km = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), len(rows[0]) if rows else 0).copy()
    n = km.shape[0]
    if counts.ndim != 3 or counts.shape[2] != 5:
        raise ValueError("write_counts_tsv: counts must be [num_ds, N, 5] or [N, num_ds, 5]")
    if counts.shape[0] == n:                          # [N, num_ds, 5] (takes precedence when both forms fit, as before) -> planar
        counts = counts.transpose(1, 0, 2)
    if counts.shape[1] != n:
        raise ValueError("write_counts_tsv: counts do not match the number of k-mers")
    if counts.size and (counts.min() < 0 or counts.max() > 0xffffffff):
        raise ValueError("write_counts_tsv: counts must fit uint32 (KMC's counter range, summarize.py:66-67)")
    planar = np.ascontiguousarray(counts, dtype=np.uint32)
    # the native writer of the summarize stage (csrc/bear_parse.cpp): formats in C++ instead of one Python call per number
    st = _lib.lib().bear_write_counts_tsv(os.fsencode(path), km.ctypes.data, planar.ctypes.data, n, km.shape[1], planar.shape[0], 0, 1, 0)
    _lib.check(st, "bear_write_counts_tsv")

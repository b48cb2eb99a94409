"""CPU tests of the ingestion add-ons: the binary cache of a parsed table and the shuffle
permutation (host evaluation through the C ABI against the oracle restatement)."""
import ctypes
import os
import shutil

import numpy as np

import bear_oracle as o
from conftest import YSD1


def test_binary_cache_roundtrip_and_staleness(tmp_path, ysd1):
    from bear_amd import _lib, dataloader
    src = tmp_path / "table.tsv"
    shutil.copy(YSD1, src)
    cdir = tmp_path / "cache"
    d0 = dataloader.dataloader(str(src), "dna", 500, 3, binary_cache=str(cdir))
    cpath = dataloader.cache_path_for(str(src), str(cdir))
    assert os.path.exists(cpath)
    kmers, counts = ysd1                                         # oracle parse of the same file
    assert np.array_equal(d0.counts, counts.transpose(1, 0, 2).astype(np.uint32))
    # second load: served from the cache (the text is not needed any more -> make it unparsable but same size/mtime)
    st = os.stat(src)
    raw = src.read_bytes()
    src.write_bytes(b"#" * len(raw))
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns))
    d1 = dataloader.dataloader(str(src), "dna", 500, 3, binary_cache=str(cdir))
    assert np.array_equal(d1.counts, d0.counts) and np.array_equal(d1.kmers, d0.kmers)
    assert [bytes(r).decode() for r in d1.kmers[:3]] == list(kmers[:3])
    # a rank reads only its shard
    L = _lib.lib()
    km = np.zeros((100, 5), dtype=np.uint8)
    cn = np.zeros((3, 100, 5), dtype=np.uint32)
    assert L.bear_cache_read(cpath.encode(), 200, 100, km.ctypes.data, cn.ctypes.data) == 0
    assert np.array_equal(cn, d0.counts[:, 200:300]) and np.array_equal(km, d0.kmers[200:300])
    assert L.bear_cache_read(cpath.encode(), 1300, 100, km.ctypes.data, cn.ctypes.data) == -1   # beyond the table
    # stale: source rewritten with different content -> cache ignored and rebuilt
    src.write_bytes(raw[: raw.index(b"\n", 2000) + 1])
    d2 = dataloader.dataloader(str(src), "dna", 500, 3, binary_cache=str(cdir))
    assert d2.num_rows < d0.num_rows and np.array_equal(d2.counts, d0.counts[:, :d2.num_rows])
    n = ctypes.c_uint64()
    assert L.bear_cache_info(cpath.encode(), ctypes.byref(n), None, None, None, None) == 0 and n.value == d2.num_rows
    # a corrupt cache is rejected, not trusted
    with open(cpath, "r+b") as fh:
        fh.write(b"XXXX")
    assert L.bear_cache_info(cpath.encode(), ctypes.byref(n), None, None, None, None) == -7
    d3 = dataloader.dataloader(str(src), "dna", 500, 3, binary_cache=str(cdir))
    assert np.array_equal(d3.counts, d2.counts)


def test_shuffle_permutation_is_a_bijection_and_matches_the_abi():
    from bear_amd import kernels
    for n in [1, 2, 3, 5, 16, 17, 1365, 4097, 100003]:
        p = o.shuffle_perm(n, 20211012)
        assert np.array_equal(np.sort(p), np.arange(n))
        for i in {0, n // 3, n - 1}:
            assert kernels.shuffle_source_row(i, n, 20211012) == p[i]
    # different seeds give different orders; fixed points are rare
    a, b = o.shuffle_perm(10000, 1), o.shuffle_perm(10000, 2)
    assert (a != b).mean() > 0.99 and (a == np.arange(10000)).mean() < 0.01


def test_threaded_parser_matches_single_thread(tmp_path):
    """A table above the 1 MiB threshold is cut at line boundaries and parsed by several threads: same result as one
    thread, including blank lines and an error in the middle of the file."""
    from bear_amd import _lib, dataloader
    rng = np.random.default_rng(1)
    n, lag = 60000, 7
    letters = np.frombuffer(b"ACGT[", dtype=np.uint8)
    kmers = letters[rng.integers(0, 5, size=(n, lag))]
    counts = rng.integers(0, 5, size=(2, n, 5)).astype(np.uint32)
    counts[0, ::97, 0] = 4000000000
    path = tmp_path / "big.tsv"
    L = _lib.lib()
    assert L.bear_write_counts_tsv(str(path).encode(), kmers.ctypes.data, counts.ctypes.data, n, lag, 2, 0, 1, 0) == 0
    text = path.read_bytes()
    assert len(text) > (1 << 20)
    lines = text.split(b"\n")
    lines.insert(20000, b"")                       # blank lines are skipped (dataloader.py: CsvDataset ignores them too)
    lines.insert(45000, b"   ")
    path.write_bytes(b"\n".join(lines))
    results = []
    for threads in ("1", "8", "3"):
        os.environ["BEAR_PARSE_THREADS"] = threads
        d = dataloader.dataloader(str(path), "dna", 1000, 2)
        results.append(d)
        assert d.num_rows == n and np.array_equal(d.counts, counts) and np.array_equal(d.kmers, kmers)
    # a malformed row in the middle fails loudly whichever thread meets it
    bad = lines[:]
    bad[30000] = bad[30000].replace(b"[[", b"[[x")
    path.write_bytes(b"\n".join(bad))
    import pytest
    for threads in ("1", "8"):
        os.environ["BEAR_PARSE_THREADS"] = threads
        with pytest.raises(_lib.BearError):
            dataloader.dataloader(str(path), "dna", 1000, 2)
    os.environ.pop("BEAR_PARSE_THREADS")


def test_parser_fuzz_against_oracle(tmp_path):
    """Seeded random tables in the summarize.py row format, with the separators the reference's json.loads accepts
    (spaces after commas): the C++ reader and the oracle's parser agree on every one."""
    from bear_amd import dataloader
    rng = np.random.default_rng(11)
    for trial in range(20):
        n, lag, num_ds = int(rng.integers(1, 40)), int(rng.integers(1, 9)), int(rng.integers(1, 4))
        kmers = ["".join(rng.choice(list("ACGT["), size=lag)) for _ in range(n)]
        counts = rng.integers(0, 10 ** rng.integers(1, 10), size=(n, num_ds, 5))
        sep = ", " if trial % 2 else ","
        path = tmp_path / f"t{trial}.tsv"
        with open(path, "w") as fh:
            for k, rows in zip(kmers, counts):
                fh.write(k + "\t[[" + "],[".join(sep.join(str(int(v)) for v in g) for g in rows) + "]]\n")
                if trial % 5 == 0:
                    fh.write("\n")
        want_k, want_c = o.parse_counts_tsv(str(path), num_ds)
        d = dataloader.dataloader(str(path), "dna", 7, num_ds)
        assert [bytes(r).decode() for r in d.kmers] == list(want_k)
        assert np.array_equal(d.counts.transpose(1, 0, 2), want_c.astype(np.uint32))
        batches = list(d)
        assert sum(len(b[0]) for b in batches) == n and batches[0][1].shape[1:] == (num_ds, 5)

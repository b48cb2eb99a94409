"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/bear_hip.h declares, fails loudly without a device, and its host-side count-table
parser reproduces the reference's golden first batch."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, YSD1


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "bear_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bear_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from bear_amd import _lib
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/bear_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names
    hdr = open(os.path.join(ROOT, "include", "bear_hip.h")).read()
    assert L.bear_abi_version() == int(re.search(r"#define BEAR_ABI_VERSION (\d+)", hdr).group(1))
    assert L.bear_strerror(0) == b"ok"


def test_deterministic_build_exports_the_same_abi():
    """libbear_hip_det.so (the library BEAR_AMD_DETERMINISTIC=1 selects at import) is the same ABI built with -DBEAR_DET_BUILD: every
    declared symbol, the same version; bear_deterministic_build tells the two apart."""
    from bear_amd import _lib
    reg = _lib.lib()
    det_path = os.path.join(ROOT, "bear_amd", "libbear_hip_det.so")
    assert os.path.exists(det_path), "make -C bear_amd/csrc builds both libraries"
    det = ctypes.CDLL(det_path)
    for n in _declared_symbols():
        assert hasattr(det, n), n
    assert det.bear_abi_version() == reg.bear_abi_version()
    if not os.environ.get("BEAR_AMD_DETERMINISTIC"):
        assert reg.bear_deterministic_build() == 0
    assert det.bear_deterministic_build() == 1


def test_no_device_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from bear_amd import _lib, kernels
    h = ctypes.c_void_p()
    st = _lib.lib().bear_ws_create(0, ctypes.byref(h))
    assert st == -2  # BEAR_ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        kernels.Workspace()


def test_parser_matches_reference_golden(ysd1):
    from bear_amd import _lib
    L = _lib.lib()
    n = ctypes.c_uint64()
    assert L.bear_count_rows(YSD1.encode(), ctypes.byref(n)) == 0
    assert n.value == 1365
    kmers = np.zeros((1365, 5), dtype=np.uint8)
    counts = np.zeros((3, 1365, 5), dtype=np.uint32)
    got = ctypes.c_uint64()
    st = L.bear_parse_counts_tsv(YSD1.encode(), 3, 5, 1365, kmers.ctypes.data, counts.ctypes.data, ctypes.byref(got))
    assert st == 0 and got.value == 1365
    ref_kmers, ref_counts = ysd1
    assert [bytes(k).decode() for k in kmers] == ref_kmers
    assert np.array_equal(counts.transpose(1, 0, 2).astype(np.float64), ref_counts)  # bit-exact counts
    # tests/test_dataloader.py:25-30
    assert bytes(kmers[0]) == b"TAATC" and counts[0, 0].tolist() == [14837, 15127, 22260, 16279, 446]
    assert counts[2, 1].tolist() == [69, 0, 45, 39, 0]


def test_parser_errors(tmp_path):
    from bear_amd import _lib
    L = _lib.lib()
    n = ctypes.c_uint64()
    assert L.bear_count_rows(b"/nonexistent/file.tsv", ctypes.byref(n)) == -6
    bad = tmp_path / "bad.tsv"
    bad.write_text("ACG\t[[1,2,3,4,5],[1,2,3,4]]\n")
    counts = np.zeros((2, 4, 5), dtype=np.uint32)
    st = L.bear_parse_counts_tsv(str(bad).encode(), 2, 3, 4, None, counts.ctypes.data, ctypes.byref(n))
    assert st == -7
    ok = tmp_path / "ok.tsv"
    ok.write_text("[[A\t[[1,2,3,4,5],[0,0,0,0,4000000000]]\n\nAC[\t[[0,0,0,0,0],[7,8,9,10,11]]\n")
    kmers = np.zeros((4, 3), dtype=np.uint8)
    st = L.bear_parse_counts_tsv(str(ok).encode(), 2, 3, 4, kmers.ctypes.data, counts.ctypes.data, ctypes.byref(n))
    assert st == 0 and n.value == 2
    assert bytes(kmers[0]) == b"[[A" and counts[1, 0, 4] == 4000000000 and counts[1, 1].tolist() == [7, 8, 9, 10, 11]
    # wrong lag
    st = L.bear_parse_counts_tsv(str(ok).encode(), 2, 4, 4, None, counts.ctypes.data, ctypes.byref(n))
    assert st == -7


def _write_table(path, n, rng, num_ds=3, lag=4, header=False):
    kmers = ["".join(rng.choice(list("ACGT["), lag)) for _ in range(n)]
    counts = rng.integers(0, 50, size=(n, num_ds, 5))
    with open(path, "w") as fh:
        if header:
            fh.write("kmer\tcounts\n")
        for k, rows in zip(kmers, counts):
            fh.write(k + "\t[[" + "],[".join(",".join(str(int(v)) for v in g) for g in rows) + "]]\n")
    return kmers, counts


@pytest.mark.parametrize("n,batch,world", [(1000, 1000, 2), (1001, 250, 3), (37, 5, 8), (64, 100, 4), (5, 2, 8)])
def test_sharded_reader_partitions_the_table(tmp_path, n, batch, world):
    """bear_parse_counts_tsv_shard / dataloader(shard=...): the ranks' pieces are disjoint, cover the table, follow
    dist.shard_rows per batch, and hold exactly the rows the plain reader returns (SURVEY 8e input sharding)."""
    from bear_amd import dataloader, dist
    rng = np.random.default_rng(n * 31 + world)
    path = tmp_path / "t.tsv"
    kmers, counts = _write_table(path, n, rng)
    full = dataloader.dataloader(str(path), "dna", batch, 3)
    assert full.num_rows == n and full.shard is None
    assert np.array_equal(full.counts.transpose(1, 0, 2), counts)
    seen = np.zeros(n, dtype=int)
    for r in range(world):
        part = dataloader.dataloader(str(path), "dna", batch, 3, shard=(r, world))
        assert part.num_rows == n and part.shard == (r, world) and part.batch_bounds() == full.batch_bounds()
        off = 0
        for (a, b), (g0, g1, o) in zip(full.batch_bounds(), part.rank_pieces(r, world)):
            lo, hi = dist.shard_rows(b - a, r, world)
            assert (g0, g1, o) == (a + lo, a + hi, off)
            assert np.array_equal(part.counts[:, o:o + g1 - g0], full.counts[:, g0:g1])
            assert np.array_equal(part.kmers[o:o + g1 - g0], full.kmers[g0:g1])
            seen[g0:g1] += 1
            off += g1 - g0
        assert off == part.local_rows
        # the unsharded dataset describes the same pieces for a rank that slices at upload
        assert [(g0, g1) for g0, g1, _ in full.rank_pieces(r, world)] == [(g0, g1) for g0, g1, _ in part.rank_pieces(r, world)]
        with pytest.raises(ValueError):
            part.rank_pieces((r + 1) % world, world)
        with pytest.raises(ValueError):
            part.shuffle(3)
    assert np.all(seen == 1)


def test_sharded_reader_multi_file_cache_and_header(tmp_path):
    from bear_amd import dataloader
    rng = np.random.default_rng(9)
    sizes, batch, world = [130, 77, 201], 100, 3
    tables = [_write_table(tmp_path / f"f{i}.tsv", n, rng) for i, n in enumerate(sizes)]
    whole = np.concatenate([c for _, c in tables])
    total = sum(sizes)
    for use_cache in (False, True):
        if use_cache:       # caches are written by an unsharded load, then served to the ranks by ranged reads
            for i in range(len(sizes)):
                dataloader.dataloader(str(tmp_path / f"f{i}.tsv"), "dna", batch, 3, binary_cache=str(tmp_path / "cache"))
        seen = np.zeros(total, dtype=int)
        for r in range(world):
            parts, base = [], 0
            for i, n in enumerate(sizes):
                parts.append(dataloader.dataloader(str(tmp_path / f"f{i}.tsv"), "dna", batch, 3, shard=(r, world), row_base=base,
                                                   total_rows=total, binary_cache=str(tmp_path / "cache") if use_cache else None))
                base += n
            data = dataloader.concatenate(parts)
            assert data.num_rows == total and data.shard == (r, world)
            for g0, g1, o in data.rank_pieces(r, world):
                assert np.array_equal(data.counts[:, o:o + g1 - g0].transpose(1, 0, 2), whole[g0:g1])
                seen[g0:g1] += 1
        assert np.all(seen == 1)
    # header=True (dataloader.py:7): the first line is skipped, sharded or not
    kmers, counts = _write_table(tmp_path / "h.tsv", 50, rng, header=True)
    d = dataloader.dataloader(str(tmp_path / "h.tsv"), "dna", 20, 3, header=True)
    assert d.num_rows == 50 and np.array_equal(d.counts.transpose(1, 0, 2), counts) and bytes(d.kmers[0]).decode() == kmers[0]
    p = dataloader.dataloader(str(tmp_path / "h.tsv"), "dna", 20, 3, header=True, shard=(1, 2))
    g = p.rank_pieces(1, 2)
    assert np.array_equal(p.counts[:, :g[0][1] - g[0][0]].transpose(1, 0, 2), counts[g[0][0]:g[0][1]])


def test_sharded_reader_large_file_threads(tmp_path):
    """> 1 MiB of text: the chunked, threaded path of the sharded reader agrees with the plain reader."""
    from bear_amd import dataloader
    rng = np.random.default_rng(4)
    n = 40000
    path = tmp_path / "big.tsv"
    _write_table(path, n, rng, num_ds=2, lag=9)
    assert os.path.getsize(path) > (1 << 20)
    full = dataloader.dataloader(str(path), "dna", 7001, 2)
    os.environ["BEAR_PARSE_THREADS"] = "7"
    try:
        for r in (0, 3, 4):
            part = dataloader.dataloader(str(path), "dna", 7001, 2, shard=(r, 5))
            for g0, g1, o in part.rank_pieces(r, 5):
                assert np.array_equal(part.counts[:, o:o + g1 - g0], full.counts[:, g0:g1])
                assert np.array_equal(part.kmers[o:o + g1 - g0], full.kmers[g0:g1])
    finally:
        os.environ.pop("BEAR_PARSE_THREADS")


def test_release_build_refuses_developer_switches():
    """The kernel sources carried "timing only, results meaningless" branches behind macros until round 6 (one stray -D and the
    shipped library computed garbage).  They are gone from the sources (scripts/dev/patches/timing_switches.patch re-adds them),
    and `bear_release_guard.h` -- included by every kernel source through bear_common.h -- stops a build that defines one of
    their names, or a stamp / probe switch, without -DBEAR_DEV_BUILD."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "bear_amd", "csrc")
    guard = os.path.join(csrc, "bear_release_guard.h")

    def preprocess(*defs):
        return subprocess.run(["gcc", "-E", "-x", "c++", *defs, guard], capture_output=True, text=True)
    assert preprocess().returncode == 0
    for name in ("LIN_SKIP_B", "LIN_FAKE_TRIPLE_ROWS=1", "LIN_MIX", "LIN_NOSYNC", "LIN_DBG=2", "PLN_NOWORK", "EVP_DEBUG_SWITCHES", "LIN_STAMPS",
                 "PLN_STAMPS", "EVP_STAMPS", "CNN_STAMPS"):
        p = preprocess("-D" + name)
        assert p.returncode != 0 and "release builds refuse it" in p.stderr, name
        assert preprocess("-D" + name, "-DBEAR_DEV_BUILD").returncode == 0, name
    assert '#include "bear_release_guard.h"' in open(os.path.join(csrc, "bear_common.h")).read()
    # no shipped source still tests one of the timing-only names, and the Makefile defines none of them
    names = re.findall(r"defined\((\w+)\)", open(guard).read().split("#error")[0])
    assert "LIN_SKIP_B" in names and "PLN_NOWORK" in names
    for fn in os.listdir(csrc):
        if fn == "bear_release_guard.h" or not fn.endswith((".h", ".hip", ".cpp", "Makefile")):
            continue
        text = open(os.path.join(csrc, fn)).read()
        for n in names:
            assert not re.search(r"\b%s\b" % n, text), (fn, n)

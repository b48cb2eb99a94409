"""CPU tests of the summarize path: the oracle's in-memory count (bear_model/tests/test_summarize.py:88-115) against the
reference pipeline's own output files for the ex_seqs sequences (bear_model/data/kmaps/ex_seqs_lag_*_file_0.tsv), and
the host pieces of bear_amd.summarize (sequence readers, text encoding, bin sizing, the C++ table writer)."""
import io
import os

import numpy as np

import bear_oracle as o
from conftest import GOLDEN

SUM = os.path.join(GOLDEN, "summarize")


def test_oracle_count_matches_reference_pipeline_files():
    want_seqs = ["TTTAT", "TTCTT", "TTTTT", "TTTTT"]              # tests/test_var_prob.py:13
    got = o.count_transitions(want_seqs, [0] * 4, 3)
    for lag in (1, 2, 3):
        kmers, counts = o.parse_counts_tsv(os.path.join(SUM, f"ex_seqs_lag_{lag}_file_0.tsv"), 1)
        assert set(kmers) == set(got[lag - 1])
        for k, c in zip(kmers, counts):
            assert np.array_equal(c.astype(np.int64), got[lag - 1][k]), (lag, k)


def test_sequence_readers_and_encoding():
    from bear_amd import summarize
    fa = io.StringIO(">a desc\nACG\nTTA\n>b\nGG\n")
    assert list(summarize.load_input(fa, "fa")) == [("a desc", "ACGTTA"), ("b", "GG")]
    fq = io.StringIO("@r1\nACGT\n+\nFFFF\n@r2\nTT\n+\nFF\n")
    assert list(summarize.load_input(fq, "fq")) == [("r1", "ACGT"), ("r2", "TT")]
    with open(os.path.join(SUM, "infile_1.fq")) as fh:
        assert [s for _, s in summarize.load_input(fh, "fq")][0] == "AATCCGTAGCCGTTT"
    text, grp = summarize.encode_sequences(["ACGT", "nA"], [0, 3])
    assert text.tolist() == [5, 0, 1, 2, 3, 4, 5, 6, 0, 4] and grp.tolist() == [0] * 6 + [3] * 4
    text, _ = summarize.encode_sequences(["AACG"], [1], reverse=True)           # reverse complement CGTT follows
    assert text.tolist() == [5, 0, 0, 1, 2, 4, 5, 1, 2, 3, 3, 4]
    assert summarize.compute_n_bin_bits(3e9, 2, 0.1) == 6 and summarize.compute_n_bin_bits(10, 1, 0.1) == 0


def test_table_writer_roundtrip(tmp_path, ysd1):
    from bear_amd import dataloader
    kmers, counts = ysd1
    km = np.frombuffer("".join(kmers).encode(), dtype=np.uint8).reshape(len(kmers), 5).copy()
    d = dataloader.CountDataset(km, np.ascontiguousarray(counts.transpose(1, 0, 2).astype(np.uint32)), "dna", 500)
    # the bundled table (lag 5, 3 groups) split over 4 bins
    from bear_amd import _lib
    L = _lib.lib()
    seen = {}
    for b in range(4):
        path = str(tmp_path / f"t_lag_5_file_{b}.tsv")
        assert L.bear_write_counts_tsv(path.encode(), d.kmers.ctypes.data, d.counts.ctypes.data, d.num_rows, 5, 3, b, 4, 0) == 0
        k2, c2 = o.parse_counts_tsv(path, 3)
        assert len(k2) == len(range(b, d.num_rows, 4))
        seen.update({k: c for k, c in zip(k2, c2)})
    assert len(seen) == len(kmers)
    for k, c in zip(kmers, counts):
        assert np.array_equal(seen[k], c)
    # byte-identical to the reference's row format on the first row (summarize.py:429-449)
    first = open(tmp_path / "t_lag_5_file_0.tsv").readline()
    assert first == open(os.path.join(GOLDEN, "ysd1_lag_5_file_0_preshuf.tsv")).readline()


def test_cpp_sequence_reader_matches_python_encoding(tmp_path):
    """bear_fastx_encode against load_input + encode_sequences on the reference's example inputs, multi-line FASTA with
    CRLF / blank / lower-case / N, empty records, forward and with reverse complements."""
    from bear_amd import summarize
    fa = tmp_path / "m.fa"
    fa.write_bytes(b">a x\r\nACGT\r\nacgn\r\n>b\n\n>c\nTTTT\nGG\n")
    fq = tmp_path / "m.fq"
    fq.write_bytes(b"@r1\nACGTA\n+\nFFFFF\n@r2\nNNA\n+r2\nFFF\n")
    lst = tmp_path / "l.csv"
    rows = [(str(fa), 2, "fa"), (str(fq), 0, "fq")] + [(os.path.join(SUM, f"infile_{j}.{t}"), g, t)
                                                         for j, (g, t) in enumerate([(0, "fa"), (0, "fq"), (2, "fq"), (1, "fa"), (1, "fq")])]
    lst.write_text("".join(f"{p},{g},{t}\n" for p, g, t in rows))
    for reverse in (False, True):
        seqs, groups = summarize._load_sequences(str(lst))
        assert seqs[:3] == ["ACGTacgn", "", "TTTTGG"]
        want_text, want_grp = summarize.encode_sequences(seqs, groups, reverse)
        text, grp, n_groups = summarize.load_text(str(lst), reverse)
        assert n_groups == 3 and np.array_equal(text, want_text) and np.array_equal(grp, want_grp)


def test_sequence_reader_errors(tmp_path):
    import ctypes
    from bear_amd import _lib
    L = _lib.lib()
    n = ctypes.c_uint64()
    bad = tmp_path / "bad.fq"
    bad.write_text("@r1\nACGT\nFFFF\n")                      # separator line missing
    assert L.bear_fastx_size(str(bad).encode(), 1, 0, ctypes.byref(n), None) == -7
    assert L.bear_fastx_size(str(tmp_path / "missing.fa").encode(), 0, 0, ctypes.byref(n), None) == -6
    ok = tmp_path / "ok.fa"
    ok.write_text(">a\nACGT\n")
    assert L.bear_fastx_size(str(ok).encode(), 0, 1, ctypes.byref(n), None) == 0 and n.value == 12
    text = np.zeros(6, dtype=np.uint8)
    got = ctypes.c_uint64()
    assert L.bear_fastx_encode(str(ok).encode(), 0, 1, 0, 6, text.ctypes.data, None, ctypes.byref(got)) == -1   # capacity too small
    text = np.zeros(12, dtype=np.uint8)
    assert L.bear_fastx_encode(str(ok).encode(), 0, 1, 300, 12, text.ctypes.data, None, ctypes.byref(got)) == -1  # group id out of range
    assert L.bear_fastx_encode(str(ok).encode(), 0, 1, 3, 12, text.ctypes.data, None, ctypes.byref(got)) == 0
    assert text.tolist() == [5, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4]                                                   # ACGT is its own reverse complement

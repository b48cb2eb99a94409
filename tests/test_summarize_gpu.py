"""GPU tests of the device summarize path: bear_kmer_sort_* + the summarize host mirror against
(a) the reference pipeline's own output files for the ex_seqs sequences (bear_model/data/kmaps/ex_seqs_lag_*_file_0.tsv,
committed under tests/golden/summarize), (b) the in-memory count of bear_model/tests/test_summarize.py:88-115 on the
reference's example inputs (tests/exdata/infile_*), forward and reverse, and (c) seeded random sequences.  Bit exact."""
import os
import types

import numpy as np
import pytest

import bear_oracle as o
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
SUM = os.path.join(GOLDEN, "summarize")


def _table_dict(d):
    return {bytes(k).decode(): d.counts[:, i].astype(np.int64) for i, k in enumerate(d.kmers)}


def _assert_equal(tables, want):
    assert len(tables) == len(want)
    for d, w in zip(tables, want):
        got = _table_dict(d)
        assert set(got) == set(w)
        for k in w:
            assert np.array_equal(got[k], w[k]), k


def _read_tsv(path, num_ds):
    kmers, counts = o.parse_counts_tsv(path, num_ds)
    return {k: counts[i].astype(np.int64) for i, k in enumerate(kmers)}


def test_ex_seqs_matches_reference_pipeline_output(tmp_path):
    """The bundled kmaps were made by the reference from TTTAT, TTCTT, TTTTT, TTTTT (tests/test_var_prob.py:13)."""
    from bear_amd import summarize
    fa = tmp_path / "ex.fa"
    fa.write_text(">a\nTTTAT\n>b\nTTCTT\n>c\nTTTTT\n>d\nTTTTT\n")
    lst = tmp_path / "list.csv"
    lst.write_text(f"{fa},0,fa\n")
    args = types.SimpleNamespace(file=str(lst), out_prefix=str(tmp_path / "ex_seqs"), l=3, nf=False, r=False, mf=0.1)
    n_bins, n_bins_rev = summarize.main(args)
    assert (n_bins, n_bins_rev) == (1, None)
    for lag in (1, 2, 3):
        want = _read_tsv(os.path.join(SUM, f"ex_seqs_lag_{lag}_file_0.tsv"), 1)
        got = _read_tsv(str(tmp_path / f"ex_seqs_lag_{lag}_file_0.tsv"), 1)
        assert set(got) == set(want)
        for k in want:
            assert np.array_equal(got[k], want[k]), (lag, k)


def _exdata_list(tmp_path):
    groups = [0, 0, 2, 1, 1]                           # tests/test_summarize.py:53-54
    types_ = ["fa", "fq", "fq", "fa", "fq"]
    lst = tmp_path / "infiles.csv"
    lst.write_text("".join(f"{os.path.join(SUM, f'infile_{j}.{types_[j]}')},{groups[j]},{types_[j]}\n" for j in range(5)))
    return str(lst)


@pytest.mark.parametrize("reverse", [False, True])
def test_reference_example_inputs_match_in_memory_count(tmp_path, reverse):
    """tests/test_summarize.py:test_main on the reference's own example inputs, max lag 10, three groups."""
    from bear_amd import summarize
    lst = _exdata_list(tmp_path)
    seqs, groups = summarize._load_sequences(lst)
    assert len(seqs) == 13 and sorted(set(groups)) == [0, 1, 2]
    want = o.count_transitions(seqs, groups, 10, reverse=reverse)
    _assert_equal(summarize.count_tables(lst, 10, reverse=reverse), want)
    # through the files, as the reference test reads them back
    args = types.SimpleNamespace(file=lst, out_prefix=str(tmp_path / "out"), l=10, nf=False, r=reverse, mf=2)
    n_bins, n_bins_rev = summarize.main(args)
    prefix, nb = (str(tmp_path / "out_rev"), n_bins_rev) if reverse else (str(tmp_path / "out"), n_bins)
    for li in range(10):
        got = {}
        for b in range(nb):
            part = _read_tsv(f"{prefix}_lag_{li + 1}_file_{b}.tsv", 3)
            assert not set(part) & set(got)               # no k-mer twice (test_summarize.py:125-126)
            got.update(part)
        assert set(got) == set(want[li])
        for k in got:
            assert np.array_equal(got[k], want[li][k])


def test_random_sequences_many_bins_and_invalid_letters(tmp_path):
    from bear_amd import summarize
    rng = np.random.default_rng(0)
    seqs = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in rng.integers(1, 400, size=300)]
    seqs += ["A", "", "ACGTNACGT", "NNNN"]                # shorter than the lag, empty, letters outside the alphabet
    groups = [int(g) for g in rng.integers(0, 4, size=len(seqs))]
    text, grp = summarize.encode_sequences(seqs, groups)
    clean = [(s, g) for s, g in zip(seqs, groups) if "N" not in s]
    want = o.count_transitions([s for s, _ in clean], [g for _, g in clean], 21)
    letters = {"A": 0, "C": 1, "G": 2, "T": 3, "]": 4}
    for lag in (1, 5, 13, 21):
        kmers, counts = summarize.count_transitions(text, grp, lag, 4)
        got = {bytes(k).decode(): counts[:, i].astype(np.int64) for i, k in enumerate(kmers)}
        w = {k: np.pad(v, ((0, 4 - v.shape[0]), (0, 0))) for k, v in want[lag - 1].items()}
        # transitions whose window (context + next letter) holds an N are dropped, the others of that sequence count
        for s_, g_ in zip(seqs, groups):
            if "N" in s_:
                full = "[" * lag + s_ + "]"
                for j in range(lag, len(full)):
                    if "N" not in full[j - lag:j + 1]:
                        w.setdefault(full[j - lag:j], np.zeros((4, 5), dtype=np.int64))[g_, letters[full[j]]] += 1
        assert set(got) == set(w), lag
        for k in w:
            assert np.array_equal(got[k], w[k]), (lag, k)
    # totals: one transition per letter plus one stop per sequence
    kmers, counts = summarize.count_transitions(*summarize.encode_sequences([s for s, _ in clean], [g for _, g in clean]), 3, 4)
    assert counts.sum() == sum(len(s) + 1 for s, _ in clean)


def test_full_size_count_conservation():
    """1e7 positions at lag 13: every letter and every stop is exactly one transition, rows are distinct, and the lag-13
    table folds onto the lag-5 table (marginalising the 8 leading letters preserves each 5-mer's counts)."""
    import torch
    from bear_amd import summarize
    dev = torch.device("cuda", 0)
    reads, rl = 66_000, 150
    g = torch.Generator(device=dev).manual_seed(3)
    body = torch.randint(0, 4, (reads, rl), dtype=torch.uint8, device=dev, generator=g)
    text = torch.cat([torch.full((reads, 1), 5, dtype=torch.uint8, device=dev), body,
                      torch.full((reads, 1), 4, dtype=torch.uint8, device=dev)], 1).reshape(-1).contiguous()
    grp = (torch.arange(reads, device=dev) % 3).to(torch.uint8).repeat_interleave(rl + 2).contiguous()
    k13, c13 = summarize.count_transitions(text, grp, 13, 3)
    k5, c5 = summarize.count_transitions(text, grp, 5, 3)
    assert int(c13.sum()) == int(c5.sum()) == reads * (rl + 1)
    assert len({bytes(r) for r in k13[:200000]}) == 200000 and c13.shape == (3, k13.shape[0], 5)
    per_group = np.bincount(np.arange(reads) % 3, minlength=3) * (rl + 1)
    assert np.array_equal(c13.sum(axis=(1, 2)), per_group)
    # fold: suffix of length 5 of every 13-mer context
    fold = {}
    suf = k13[:, 8:]
    keys = (suf.astype(np.uint64) * (256 ** np.arange(5, dtype=np.uint64))).sum(1)
    order = np.argsort(keys, kind="stable")
    uk, start = np.unique(keys[order], return_index=True)
    folded = np.add.reduceat(c13[:, order].astype(np.int64), start, axis=1)
    k5keys = (k5.astype(np.uint64) * (256 ** np.arange(5, dtype=np.uint64))).sum(1)
    o5 = np.argsort(k5keys)
    assert np.array_equal(uk, k5keys[o5]) and np.array_equal(folded, c5[:, o5].astype(np.int64))


def test_device_resident_tables_train_like_host_tables(tmp_path):
    """count -> (shuffle) -> plan -> train without the table leaving HBM: same losses as the host-table path."""
    import torch
    from bear_amd import ar_funcs, bear_net, summarize
    rng = np.random.default_rng(5)
    fa = tmp_path / "s.fa"
    fa.write_text("".join(f">s{i}\n{''.join(rng.choice(list('ACGT'), size=int(n)))}\n" for i, n in enumerate(rng.integers(30, 200, size=200))))
    lst = tmp_path / "l.csv"
    lst.write_text(f"{fa},0,fa\n")
    host = summarize.count_tables(str(lst), 5)[4]
    dev = summarize.count_tables(str(lst), 5, on_device=True)[4]
    assert dev.num_rows == host.num_rows and np.array_equal(dev.counts, host.counts) and np.array_equal(dev.kmers, host.kmers)
    losses = []
    for data in (host, dev, host.shuffle(3), dev.shuffle(3)):
        torch.manual_seed(0)
        ls = []
        bear_net.train(data.repeat(2), data.num_rows, 2, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False, loss_save=ls)
        losses.append(ls)
    assert np.allclose(losses[0], losses[1], rtol=1e-13) and np.allclose(losses[2], losses[3], rtol=1e-13)
    assert np.isclose(losses[0][0], losses[2][0], rtol=1e-11)       # one batch = the whole table: order does not matter

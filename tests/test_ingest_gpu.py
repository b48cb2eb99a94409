"""GPU tests of the ingestion add-ons: bear_shuffle_rows against the oracle permutation (bit exact)
and training on a device-shuffled table against the oracle loop run on the permuted rows."""
import numpy as np
import pytest
import torch

import bear_oracle as o
from bear_amd import ar_funcs, bear_ref, dataloader, kernels
from conftest import YSD1

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 7, 1365, 100003])
def test_shuffle_rows_matches_oracle_permutation(n):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n)
    perm = o.shuffle_perm(n, 99)
    for shape, dtype in (((n, 5), np.int32), ((n, 13), np.int8), ((n,), np.int64), ((n, 5), np.float64)):
        a = rng.integers(-100, 100, size=shape).astype(dtype)
        got = kernels.shuffle_rows(torch.from_numpy(a).to(dev), 99).cpu().numpy()
        assert np.array_equal(got, a[perm]), (shape, dtype)


def test_training_on_shuffled_table_matches_oracle_on_permuted_rows(ysd1):
    _, counts = ysd1
    perm = o.shuffle_perm(1365, 5)
    data = dataloader.dataloader(YSD1, "dna", 500, 3).shuffle(5)
    loss_save = []
    bear_ref.train(data.repeat(1), 1365, 1, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False,
                   loss_save=loss_save)
    c = counts[perm]
    p = np.array([0.0, np.log(1 / 30), -np.log(100)])
    r = o.bear_ref_step(c[:500, 0], c[:500, 2], *p)
    assert np.isclose(loss_save[0], (1365 / 500) * r["ll"], rtol=1e-11)
    # the batches differ from file order (that is the point), the table total does not
    r_file = o.bear_ref_step(counts[:500, 0], counts[:500, 2], *p)
    assert not np.isclose(r["ll"], r_file["ll"], rtol=1e-6)
    full = dataloader.dataloader(YSD1, "dna", 1365, 3)
    l0, l1 = [], []
    bear_ref.train(full.repeat(1), 1365, 1, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False, loss_save=l0)
    bear_ref.train(full.shuffle(11).repeat(1), 1365, 1, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False,
                   loss_save=l1)
    assert np.isclose(l0[0], l1[0], rtol=1e-12)


def test_device_kmer_encoding_matches_host():
    from bear_amd import core
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    ascii_ = rng.integers(32, 127, size=(5000, 13)).astype(np.uint8)
    ascii_[:4000] = np.frombuffer(b"ACGTU[acgtN]", dtype=np.uint8)[rng.integers(0, 12, size=(4000, 13))]
    for alphabet in ("dna", "rna"):
        got = kernels.encode_kmers(torch.from_numpy(ascii_).to(dev), alphabet).cpu().numpy()
        assert np.array_equal(got, core.encode_kmers(ascii_, alphabet))


def test_kmer_order_and_gather_rows():
    """bear_kmer_order_u64 / bear_gather_rows (the k-mer order the fused AR-function kernels want, include/bear_hip.h): the
    permutation sorts the contexts lexicographically with the first letter most significant, is stable, and the gather applies it
    to slabs of any row width."""
    import torch
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    for n, lag in ((1, 1), (5000, 3), (100_003, 13), (4097, 21)):
        codes = rng.integers(-1, 6, size=(n, lag)).astype(np.int8)          # -1 / 5: unknown letters (sorted last), 4: the start symbol
        d_codes = torch.from_numpy(codes).to(dev)
        packed = kernels.pack_kmers(d_codes)
        perm = kernels.kmer_order(packed, lag)
        clean = np.where((codes >= 0) & (codes <= 4), codes, 5).astype(np.int64)
        key = np.zeros(n, dtype=object)
        for l in range(lag):
            key = key * 6 + clean[:, l]
        want = np.array(sorted(range(n), key=lambda i: (key[i], i)), dtype=np.int64)      # stable
        assert np.array_equal(perm.cpu().numpy().astype(np.int64), want), (n, lag)
        for width, dtype in ((5, torch.int32), (1, torch.int64), (lag, torch.int8), (7, torch.uint8)):
            src = torch.from_numpy(rng.integers(0, 100, size=(n, width))).to(dev).to(dtype).contiguous()
            if width == 1:
                src = src.reshape(n)
            got = kernels.gather_rows(src, perm)
            assert torch.equal(got, src[perm.long()]), (n, lag, width)
    with pytest.raises(Exception):
        kernels.kmer_order(packed, 22)

"""CPU tests of the host logic: plugin surface, data plane, optimizer, sharding helpers, config semantics."""
import configparser
import json
import os

import numpy as np
import pytest
import torch

import bear_oracle as o
from bear_amd import ar_funcs, core, dataloader, dist, _train
from conftest import GOLDEN, ROOT, YSD1


def test_ar_funcs_match_oracle():
    g = torch.Generator().manual_seed(0)
    kmers = ["ACGTA", "[[ACG", "TTTTN", "GATTA", "[[[[["]
    codes = torch.as_tensor(core.encode_kmers(kmers))
    oh = core.tf_one_hot(kmers, "dna")
    assert np.array_equal(oh.numpy(), o.one_hot(kmers))
    f, p = ar_funcs.make_ar_func_linear(5, 4, generator=g)
    assert p[0].shape == (5, 5, 5) and p[0].requires_grad
    assert np.allclose((p[0] ** 2).sum(1).detach().numpy(), 0.05 ** 2)      # 0.05 * l2_normalize(axis=1), ar_funcs.py:41-42
    want = o.ar_func_linear(o.one_hot(kmers), p[0].detach().numpy())
    assert np.allclose(f(oh).detach().numpy(), want, atol=1e-15) and np.allclose(f(codes).detach().numpy(), want, atol=1e-15)
    f, p = ar_funcs.make_ar_func_cnn(5, 4, filter_width=3, num_filters=6, generator=g)
    assert [tuple(x.shape) for x in p] == [(3, 5, 6), (3, 6), (3, 6, 16), (16,), (16, 5), (5,), (3, 6), (16,)]
    want = o.ar_func_cnn(o.one_hot(kmers), [x.detach().numpy() for x in p])
    assert np.allclose(f(oh).detach().numpy(), want, atol=1e-14) and np.allclose(f(codes).detach().numpy(), want, atol=1e-14)
    f, p = ar_funcs.make_ar_func_stop(5, 4)
    assert p == [] and f(oh).tolist() == [0, 0, 0, 0, 1]
    # gradients flow through the code path
    f, p = ar_funcs.make_ar_func_linear(5, 4, generator=g)
    f(codes).log().sum().backward()
    assert p[0].grad is not None and torch.isfinite(p[0].grad).all()


def test_plugin_lookup_by_name():
    # models/train_bear_net.py:103
    for name in ("linear", "cnn", "stop"):
        assert callable(getattr(ar_funcs, "make_ar_func_" + name))


def test_dataloader_reference_golden_batches():
    # bear_model/tests/test_dataloader.py:20-32
    data = dataloader.dataloader(YSD1, "dna", 3, 3)
    kmers, counts = next(iter(data))
    assert np.all(kmers == np.array([b"TAATC", b"CGGTC", b"ACGCT"]))
    counts_real = [[[14837, 15127, 22260, 16279, 446], [5029, 5095, 7408, 5487, 134], [16, 16, 23, 17, 0]],
                   [[61890, 729, 39733, 35956, 1017], [20524, 239, 13199, 12046, 309], [69, 0, 45, 39, 0]],
                   [[13965, 23135, 73870, 37045, 1035], [4705, 7591, 24532, 12305, 385], [14, 25, 81, 39, 0]]]
    assert np.all(counts.numpy() == np.array(counts_real))
    assert counts.dtype == torch.float64
    assert len(list(data)) == 1365 / 3
    big = dataloader.dataloader(YSD1, "dna", 2000, 3)
    k2, c2 = next(iter(big))
    assert len(c2) == 1365 < 2000 and len(big) == 1            # one short batch (no drop_remainder)
    assert len(big.repeat(4)) == 4
    assert big.codes().shape == (1365, 5) and (big.codes() == 4).any()   # '[' padded prefixes


def test_sparse_dataloader():
    s = dataloader.sparse_dataloader(os.path.join(GOLDEN, "ex_seqs_kmap_for_var_pred.csv"), "dna", 4, 1)
    assert s.num_rows == 9 and s.lag == 3 and s.num_ds == 1
    k, c = next(iter(s))
    assert k[0] == b"CTT" and c[0, 0].tolist() == [0, 0, 0, 0, 1] and c[2, 0].tolist() == [0, 0, 0, 1, 0]


def test_sparse_dataloader_against_oracle_and_malformed_rows(tmp_path):
    """Generated three-dataset sparse file: the native reader == the oracle's restatement of dataloader.py:52-109; rows it
    cannot represent are refused, not guessed at."""
    from oracle import bear_oracle as o
    rng = np.random.default_rng(3)
    path = tmp_path / "sparse.csv"
    with open(path, "w") as fh:
        fh.write("kmer; count_mat_indices; count_mat_values\n")
        for i in range(500):
            k = "".join(rng.choice(list("ACGT["), size=4))
            cells = sorted({(int(rng.integers(0, 3)), int(rng.integers(0, 5))) for _ in range(int(rng.integers(0, 6)))})
            vals = [int(rng.integers(1, 2 ** 32 - 1)) if j == 0 and i % 50 == 0 else int(rng.integers(1, 40)) for j in range(len(cells))]
            fh.write(f"{k}; {json.dumps([list(c) for c in cells])}; {json.dumps(vals)}\n" if i % 7 else
                     f" {k} ;[{','.join('[%d, %d]' % c for c in cells)}];[{', '.join(str(float(v)) if v < 100 else str(v) for v in vals)}]\n")
        fh.write("\n")
    d = dataloader.sparse_dataloader(str(path), "dna", 64, 3)
    want_k, want_c = o.parse_sparse_rows(str(path), 3, 5)
    assert d.num_rows == 500 and d.lag == 4 and [bytes(r) for r in d.kmers] == want_k
    assert np.array_equal(d.counts, want_c.transpose(1, 0, 2))
    for bad in ("ACGT; [[0,1]]\n", "ACGT; [[0,1],[1]]; [1,2]\n", "ACGT; [[3,1]]; [1]\n", "ACGT; [[0,5]]; [1]\n", "ACGT; [[0,1]]; [1.5]\n",
                "ACGT; [[0,1]]; [-1]\n", "ACGT; [[0,1]]; [1,2]\n", "ACG; [[0,1]]; [1]\n",
                "ACGT; [[-, 1]]; [3]\n", "ACGT; [[0, 1-]]; [3]\n", "ACGT; [[0,1]]; [-]\n"):     # a sign without digits (once: an endless loop)
        with open(path, "w") as fh:
            fh.write("h\nACGT; [[0,1]]; [1]\n" + bad)
        with pytest.raises(Exception):
            dataloader.sparse_dataloader(str(path), "dna", 64, 3)


def test_write_counts_tsv_native_round_trip(tmp_path):
    """write_counts_tsv goes through the native writer; both count layouts; what it writes the dense reader reads back."""
    rng = np.random.default_rng(5)
    n, lag, nds = 300, 6, 3
    km = np.frombuffer(b"ACGT[", dtype=np.uint8)[rng.integers(0, 5, size=(n, lag))]
    counts = rng.integers(0, 2 ** 32, size=(nds, n, 5), dtype=np.uint64).astype(np.uint32)
    counts[:, ::3] = 0
    for form, kmers in ((counts, km), (counts.transpose(1, 0, 2), [bytes(r) for r in km])):
        path = tmp_path / "t.tsv"
        dataloader.write_counts_tsv(str(path), kmers, form)
        back = dataloader.dataloader(str(path), "dna", 100, nds)
        assert np.array_equal(back.kmers, km) and np.array_equal(back.counts, counts)
    with pytest.raises(ValueError):
        dataloader.write_counts_tsv(str(tmp_path / "x.tsv"), km, counts[:, :10])


def test_keras_adam_update_rule():
    p = [torch.tensor(1.0, dtype=torch.float64), torch.tensor([0.5, -2.0], dtype=torch.float64)]
    opt = _train.KerasAdam(p, 0.01)
    g = [torch.tensor(3.0, dtype=torch.float64), torch.tensor([1e-3, -4.0], dtype=torch.float64)]
    opt.apply_gradients(g)
    # first step: m = 0.1 g, v = 0.001 g^2, lr_t = lr sqrt(1-b2)/(1-b1): theta -= lr_t m / (sqrt(v) + 1e-7)
    lr_t = 0.01 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want0 = 1.0 - lr_t * 0.3 / (np.sqrt(0.009) + 1e-7)
    assert abs(p[0].item() - want0) < 1e-15
    opt.apply_gradients([None, g[1]])
    assert abs(p[0].item() - want0) < 1e-15                    # None gradient leaves the variable alone (AR mode h_signed)


def test_shard_rows_partition():
    for n in (0, 1, 7, 1365, 10 ** 9 + 3):
        for w in (1, 2, 3, 8):
            r = [dist.shard_rows(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    flat, unpack = dist.pack([torch.ones(2), torch.arange(6.0).reshape(2, 3)])
    a, b = unpack(flat * 2)
    assert a.tolist() == [2, 2] and b.shape == (2, 3) and b[1, 2] == 10


def test_config_files_have_reference_keys():
    d = os.path.join(ROOT, "bear_amd", "models", "config_files")
    want = {"general": {"out_folder", "seed", "precision"},
            "data": {"files_path", "start_token", "sparse", "num_ds", "alphabet", "train_column", "test_column", "reference_column"},
            "hyperp": {"lag"},
            "train": {"train", "epochs", "batch_size", "optimizer_name", "learning_rate", "train_ar", "accumulation_steps", "cache",
                      "restart", "restart_path"},
            "test": {"test", "train_test", "van_reg"}, "model": {"ar_func_name", "af_kwargs"}, "results": set()}
    names = sorted(os.listdir(d))
    assert names == ["bear_cnn_ar.cfg", "bear_cnn_bear.cfg", "bear_lin_ar.cfg", "bear_lin_bear.cfg", "bear_stop_ar.cfg",
                     "bear_stop_bear.cfg", "bear_test.cfg"]
    for n in names:
        c = configparser.ConfigParser()
        c.read(os.path.join(d, n))
        assert {s: set(c[s].keys()) for s in c.sections()} == want, n


def test_training_needs_a_device():
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    with pytest.raises(RuntimeError):
        _train.require_device()


def test_count_table_writer_round_trip(tmp_path):
    """summarize.py row format (summarize.py:429-449): the writer's output is byte-identical to the bundled
    table and parses back to the same arrays."""
    data = dataloader.dataloader(YSD1, "dna", 100, 3)
    out = tmp_path / "copy.tsv"
    dataloader.write_counts_tsv(out, data.kmers, data.counts)
    assert open(out, "rb").read() == open(YSD1, "rb").read()
    back = dataloader.dataloader(str(out), "dna", 100, 3)
    assert np.array_equal(back.counts, data.counts) and np.array_equal(back.kmers, data.kmers)
    rng = np.random.default_rng(0)
    c = rng.integers(0, 4_000_000_000, size=(2, 7, 5), dtype=np.uint64).astype(np.uint32)
    dataloader.write_counts_tsv(tmp_path / "r.tsv", ["[[ACG"] * 7, c)
    assert np.array_equal(dataloader.dataloader(str(tmp_path / "r.tsv"), "dna", 3, 2).counts, c)


def test_bench_flop_models():
    """The useful-flop counts behind the fp64 rooflines of the bench line (bench.py:flops_*) on tables small
    enough to count by hand."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fwd, bwd = bench.flops_cnn(13, 8)
    per_pos = 8 * 30 + 5 * 30 + 9 + 30 * 20 + 2 * 30 * 16
    head = 5 * 16 + 9 + 16 * 20 + 2 * 16 * 5 + 5 * 20 + 9 + 5
    assert fwd == 6 * per_pos + head == 12437 and bwd == 3 * fwd
    train = torch.tensor([[2, 0, 0, 1, 0], [0, 0, 0, 0, 0], [30, 0, 1, 0, 3]], dtype=torch.int32)
    ref = torch.tensor([[1, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]], dtype=torch.int32)
    # mode R: records exist only for the items of contexts WITH reference counts: row 0 -> items c = 2, 1
    assert bench.flops_ref_items(train, ref) == 2 * 53 + 4 * (2 + 1)
    # linear head at lag 13 (six groups): 2 live contexts, 5 items with min(c, 24) = 2, 1, 24, 1, 3
    assert bench.flops_linear(train, 13) == 2 * 65 + 5 * 39 + 4 * (2 + 1 + 24 + 1 + 3)
    # evaluation: 2 rows with held-out counts (totals 3 and min(34, 24)), 5 cells
    assert bench.flops_eval(train) == 2 * 67 + 4 * (3 + 24) + 5 * 51 + 4 * (2 + 1 + 24 + 1 + 3)
    r = bench.fp64_roofline(78.6e12 * 1e-3, 1.0)          # 78.6 GFLOP in 1 ms = the peak
    assert abs(r["frac_of_fp64_peak"] - 1.0) < 1e-12 and r["peak_TFLOPs"] == 78.6


def test_count_newlines_is_wc_l(tmp_path):
    """The driver's num_kmers is `wc -l` (models/train_bear_net.py:52-55): newline bytes -- blank lines and a header count, a last
    line without a newline does not; small files on one thread, large ones cut by byte offset over the host threads."""
    rng = np.random.default_rng(1)
    cases = {"empty": b"", "one": b"ACGTA\t[[1,2,3,4,5]]\n", "no_final_newline": b"a\nb\nc", "blank_lines": b"a\n\n \nb\n\n",
             "large": b"".join(rng.choice([b"ACGT\t[[1,0,0,0,0]]\n", b"\n", b"x" * 37 + b"\n"], size=120_000).tolist()) + b"tail"}
    assert len(cases["large"]) > (1 << 20)
    for name, blob in cases.items():
        p = tmp_path / (name + ".tsv")
        p.write_bytes(blob)
        assert dataloader.count_newlines(p) == blob.count(b"\n"), name
    assert dataloader.count_newlines(YSD1) == 1365
    with pytest.raises(Exception):
        dataloader.count_newlines(tmp_path / "missing.tsv")


def test_synthetic_kmers_are_a_bijection_of_the_row_index():
    """the synthetic tables' contexts are the row index mapped through a FIXED BIJECTION to k-mers -- distinct, as the
    contexts of any count table are (summarize.py:429-449 writes one row per k-mer), scrambled, and the same whatever the shard."""
    import torch
    from bear_amd import kernels
    for lag in (1, 2, 5, 8):
        ids = kernels.synth_kmer_ids(20211012, 0, 4 ** lag, lag, "cpu")
        assert ids.dtype == torch.int64 and int(ids.min()) == 0 and int(ids.max()) == 4 ** lag - 1
        assert ids.unique().numel() == 4 ** lag                       # every k-mer exactly once
    a = kernels.synth_kmer_ids(7, 12345, 1000, 13, "cpu")
    assert torch.equal(a, kernels.synth_kmer_ids(7, 0, 20000, 13, "cpu")[12345:13345])      # rows [a, b): the same k-mers in any shard
    assert not torch.equal(a, kernels.synth_kmer_ids(8, 12345, 1000, 13, "cpu"))
    big = kernels.synth_kmer_ids(20211012, 0, 1_000_000, 13, "cpu")
    assert big.unique().numel() == 1_000_000
    assert abs(float(big.double().mean()) / 4 ** 13 - 0.5) < 0.01       # scrambled over the whole range
    codes = kernels.synth_kmer_codes(20211012, 0, 4 ** 3, 3, "cpu", sort=True)
    assert codes.dtype == torch.int8 and codes.shape == (64, 3)
    assert codes.tolist() == [[a, b, c] for a in range(4) for b in range(4) for c in range(4)]   # k-mer order = ids ascending
    with pytest.raises(ValueError, match="only 4\\^13"):
        kernels.synth_kmer_ids(1, 0, 100_000_000, 13, "cpu")

"""GPU tests of the host drivers: a few optimizer steps of bear_ref.train / bear_net.train against an
oracle-driven replica of the same loop, held-out evaluation against the oracle, the reference's own
test_run / test_core / test_dataloader checks re-stated on the HIP path."""
import configparser
import math
import os

import numpy as np
import pytest
import torch
from scipy.special import loggamma

import bear_oracle as o
from bear_amd import _train, ar_funcs, bear_net, bear_ref, core, dataloader, kernels
from conftest import ROOT, YSD1

pytestmark = pytest.mark.gpu


def keras_adam_np(p, g, m, v, t, lr=0.01, b1=0.9, b2=0.999, eps=1e-7):
    m[...] = b1 * m + (1 - b1) * g
    v[...] = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    p[...] = p - lr_t * m / (np.sqrt(v) + eps)


@pytest.mark.parametrize("train_ar", [False, True])
def test_bear_ref_train_matches_oracle_loop(train_ar, ysd1):
    _, counts = ysd1
    data = dataloader.dataloader(YSD1, "dna", 500, 3)          # 3 batches per epoch, last one short (365 rows)
    steps_epochs = 2
    loss_save = []
    params, h_signed, ar_func = bear_ref.train(data.repeat(steps_epochs), 1365, steps_epochs, 0, 2, "dna", 5,
                                               ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", train_ar, loss_save=loss_save)
    p = np.array([0.0, np.log(1 / 30), -np.log(100)])
    m, v = np.zeros(3), np.zeros(3)
    want_loss, t = [], 0
    for _ in range(steps_epochs):
        for a in range(0, 1365, 500):
            b = min(a + 500, 1365)
            r = o.bear_ref_step(counts[a:b, 0], counts[a:b, 2], *p, train_ar=train_ar)
            scale = -(1365 / (b - a))
            want_loss.append(-scale * r["ll"])
            g = scale * np.array([r["d_h_signed"], r["d_tau_signed"], r["d_nu_signed"]])
            t += 1
            if train_ar:                                       # h_signed gets no gradient in AR mode
                keras_adam_np(p[1:], g[1:], m[1:], v[1:], t)
            else:
                keras_adam_np(p, g, m, v, t)
    got = np.array([x.item() for x in params])
    assert np.allclose(loss_save, want_loss, rtol=1e-10)
    assert np.allclose(got, p, rtol=1e-8, atol=1e-10)
    assert h_signed is params[0]
    # the returned ar_func reproduces bear_ref.py:63-68
    oh = core.tf_one_hot(["ACGTA"], "dna", device="cuda")
    ref_in = torch.tensor([[3 + 1e-7, 1e-7, 1 + 1e-7, 1e-7, 0.0]], dtype=torch.float64, device="cuda")
    want = o.ref_ar_func(o.ar_func_stop(None), ref_in.cpu().numpy(), got[1], got[2])
    assert np.allclose(ar_func(oh, ref_in).detach().cpu().numpy(), want, rtol=1e-12)


@pytest.mark.parametrize("train_ar", [False, True])
def test_bear_ref_linear_net_matches_oracle_loop(train_ar, ysd1):
    """bear_ref.train with a parametrised net function (bear_ref.py:63-68 with make_ar_func_linear): losses and
    parameters against a CPU replica (torch autograd through the same mixing formula, likelihood and gradient
    rows from the oracle)."""
    _, counts = ysd1
    data = dataloader.dataloader(YSD1, "dna", 700, 3)
    torch.manual_seed(5)
    _, init = ar_funcs.make_ar_func_linear(5, 4)
    mat0 = init[0].detach().numpy().copy()
    restart = [np.array(0.2), np.array(np.log(1 / 30)), np.array(-1.0), mat0]
    loss_save = []
    params, h_signed, ar_func = bear_ref.train(data.repeat(2), 1365, 2, 0, 2, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01,
                                               "Adam", train_ar, params_restart=restart, loss_save=loss_save)
    f_cpu, (mat,) = ar_funcs.make_ar_func_linear(5, 4)
    with torch.no_grad():
        mat.copy_(torch.as_tensor(mat0))
    tau_s = torch.tensor(np.log(1 / 30), dtype=torch.float64, requires_grad=True)
    nu_s = torch.tensor(-1.0, dtype=torch.float64, requires_grad=True)
    h = np.array(0.2)
    flat = [h, tau_s.detach().numpy(), nu_s.detach().numpy(), mat.detach().numpy()]
    ms, vs = [np.zeros_like(x) for x in flat], [np.zeros_like(x) for x in flat]
    codes = torch.as_tensor(data.codes())
    want_loss, t = [], 0
    for _ in range(2):
        for a in range(0, 1365, 700):
            b = min(a + 700, 1365)
            for q in (tau_s, nu_s, mat):
                q.grad = None
            ref_in = torch.as_tensor(o.ref_input(counts[a:b, 2]))
            nw, tau = torch.exp(nu_s), torch.exp(tau_s)
            prior = (nw * f_cpu(codes[a:b]) + bear_ref._counts_to_probs(ref_in, tau, 4)) / (nw + 1)
            r = o.bear_net_step(counts[a:b, 0], prior.detach().numpy(), float(h), train_ar=train_ar)
            scale = -(1365 / (b - a))
            prior.backward(torch.as_tensor(scale * r["d_prior"]))
            want_loss.append(-scale * r["ll"])
            t += 1
            if not train_ar:
                keras_adam_np(h, np.array(scale * r["d_h_signed"]), ms[0], vs[0], t)
            for i, q in enumerate((tau_s, nu_s, mat)):
                keras_adam_np(q.detach().numpy(), q.grad.numpy(), ms[i + 1], vs[i + 1], t)
    assert np.allclose(loss_save, want_loss, rtol=1e-9)
    assert np.isclose(params[0].item(), float(h), rtol=1e-7, atol=1e-9)
    for got, want in zip(params[1:], (tau_s, nu_s, mat)):
        assert np.allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-6, atol=1e-8)


CNN_CFG = {"num_filters": 30, "filter_width": 3, "kmer_layer1_width": 16}     # models/config_files/bear_cnn_bear.cfg:65 -> fused kernels


@pytest.mark.parametrize("name,kw,train_ar", [("linear", {}, False), ("linear", {}, True), ("cnn", {"filter_width": 3, "num_filters": 5}, False),
                                              ("cnn", CNN_CFG, False), ("cnn", CNN_CFG, True)])
def test_bear_net_train_matches_oracle_loop(name, kw, train_ar, ysd1):
    _, counts = ysd1
    data = dataloader.dataloader(YSD1, "dna", 700, 3)          # 2 batches (700 + 665)
    make = getattr(ar_funcs, "make_ar_func_" + name)
    torch.manual_seed(3)
    _, init = make(5, 4, **kw)
    init_np = [x.detach().numpy().copy() for x in init]
    restart = [np.array(0.1)] + init_np
    loss_save = []
    params, h_signed, ar_func = bear_net.train(data.repeat(2), 1365, 2, 0, "dna", 5, make, kw, 0.01, "Adam", train_ar,
                                               params_restart=restart, loss_save=loss_save)
    # replica on CPU: torch autograd through the same plugin, likelihood and row gradients from the oracle
    f_cpu, p_cpu = make(5, 4, **kw)
    with torch.no_grad():
        for a, b in zip(p_cpu, init_np):
            a.copy_(torch.as_tensor(b))
    h = np.array(0.1)
    flat = [h] + [x.detach().numpy() for x in p_cpu]
    ms = [np.zeros_like(x) for x in flat]
    vs = [np.zeros_like(x) for x in flat]
    codes = torch.as_tensor(data.codes())
    want_loss, t = [], 0
    for _ in range(2):
        for a in range(0, 1365, 700):
            b = min(a + 700, 1365)
            for q in p_cpu:
                q.grad = None
            prior = f_cpu(codes[a:b])
            r = o.bear_net_step(counts[a:b, 0], prior.detach().numpy(), float(h), train_ar=train_ar)
            scale = -(1365 / (b - a))
            prior.backward(torch.as_tensor(scale * r["d_prior"]))
            want_loss.append(-scale * r["ll"])
            t += 1
            if not train_ar:
                gh = np.array(scale * r["d_h_signed"])
                keras_adam_np(h, gh, ms[0], vs[0], t)
            for i, q in enumerate(p_cpu):
                arr = q.detach().numpy()
                keras_adam_np(arr, q.grad.numpy(), ms[i + 1], vs[i + 1], t)
    assert np.allclose(loss_save, want_loss, rtol=1e-9)
    assert np.isclose(params[0].item(), float(h), rtol=1e-7, atol=1e-9)
    for got, want in zip(params[1:], p_cpu):
        assert np.allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-6, atol=1e-8)


def test_evaluation_matches_oracle(ysd1):
    kmers, counts = ysd1
    data = dataloader.dataloader(YSD1, "dna", 400, 3)
    torch.manual_seed(1)
    f, p = ar_funcs.make_ar_func_linear(5, 4, device="cuda")
    van = np.array([0.1, 1.0, 10.0])
    prior = o.ar_func_linear(o.one_hot(kmers), p[0].detach().cpu().numpy())
    for use_train in (True, False):
        got = bear_net.evaluation(data, 0 if use_train else -1, 1, "dna", torch.tensor(0.37), f, van, seed=11)
        w = o.evaluation_step(counts[:, 1], prior, 0.37, van, counts[:, 0] if use_train else None,
                              rng=o.HashNoise(11, 0, len(counts)))
        total = w[6]
        assert np.isclose(got[0], w[0], rtol=1e-11) and np.isclose(got[1], w[1], rtol=1e-11)
        assert np.allclose(got[2], w[2], rtol=1e-11)
        assert np.isclose(got[3], np.exp(-w[0] / total), rtol=1e-10) and np.allclose(got[5], np.exp(-w[2] / total), rtol=1e-10)
        # accuracies: ties are broken by the hashed noise stream the oracle restates (core.py:69-71); the prior
        # rows come from torch here and from NumPy in the oracle, so allow a near-tie row to flip
        assert abs(got[6] - w[3] / total) < 1e-4 and abs(got[7] - w[4] / total) < 1e-4
        assert np.all(np.abs(got[8] - w[5] / total) < 1e-12)     # integer concentrations + the same noise: exact
    hs = np.array([0.05, 0.37, 2.0])
    ll, perp, acc = bear_net.h_scan(data, 0, 1, "dna", torch.tensor(hs), f)
    for i, hv in enumerate(hs):
        w = o.evaluation_step(counts[:, 1], prior, hv, np.ones(1), counts[:, 0])
        assert np.isclose(ll[i], w[0], rtol=1e-11)
    # bear_ref.evaluation (stop net function)
    _, _, arf = bear_ref._create_params(5, 4, ar_funcs.make_ar_func_stop, {}, device=torch.device("cuda"))
    got = bear_ref.evaluation(data, 0, 1, 2, "dna", torch.tensor(0.5), arf, van)
    pri = o.ref_ar_func(o.ar_func_stop(None), o.ref_input(counts[:, 2]), np.log(1 / 30), -np.log(100))
    w = o.evaluation_step(counts[:, 1], pri, 0.5, van, counts[:, 0])
    assert np.isclose(got[0], w[0], rtol=1e-11) and np.isclose(got[1], w[1], rtol=1e-11) and np.allclose(got[2], w[2], rtol=1e-11)


def test_evaluation_keeps_only_the_contexts_with_heldout_counts(tmp_path, monkeypatch):
    """evaluation / h_scan keep resident only the contexts that hold counts in the scored column (half of a sparse table; their
    table rows travel as row_ids for the tie noise): every one of the nine results equals the evaluation over all rows
    (BEAR_AMD_ALL_ROWS=1) -- accuracies exactly, sums to rounding -- and the oracle on the whole table."""
    from util import sparse_table
    n = 6000
    tr, te, rf = sparse_table(n, 21)
    assert 0.2 < (te.any(axis=1)).mean() < 0.8            # a column worth compacting
    rng = np.random.default_rng(4)
    km = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 5))]
    path = tmp_path / "sparse.tsv"
    dataloader.write_counts_tsv(str(path), km, np.stack([tr, te, rf]))
    data = dataloader.dataloader(str(path), "dna", 2500, 3)              # three batches, the last one short
    torch.manual_seed(2)
    f, p = ar_funcs.make_ar_func_linear(5, 4, device="cuda")
    _, _, arf = bear_ref._create_params(5, 4, ar_funcs.make_ar_func_stop, {}, device=torch.device("cuda"))
    van = np.array([0.1, 1.0, 10.0])

    def run():
        a = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, van, seed=11)
        b = bear_net.evaluation(data, -1, 1, "dna", torch.tensor(0.37), f, van, seed=11)
        c = bear_net.h_scan(data, 0, 1, "dna", torch.tensor([0.05, 0.37, 2.0]), f, seed=5)
        d = bear_ref.evaluation(data, 0, 1, 2, "dna", torch.tensor(0.5), arf, van, seed=7)
        return [np.asarray(v, dtype=np.float64) for r in (a, b, c, d) for v in r]
    seen = []
    orig = kernels.evaluate_planned
    monkeypatch.setattr(kernels, "evaluate_planned", lambda plan, *a, **kw: (seen.append((plan.test.shape[0], kw.get("row_ids") is not None)),
                                                                            orig(plan, *a, **kw))[1])
    compact = run()
    assert seen and all(ids for _, ids in seen) and sum(r for r, _ in seen[:3]) == int(te.any(axis=1).sum())
    seen.clear()
    monkeypatch.setenv("BEAR_AMD_ALL_ROWS", "1")
    whole = run()
    assert seen and not any(ids for _, ids in seen) and sum(r for r, _ in seen[:3]) == n
    names = ["ll", "ll", "ll", "perp", "perp", "perp", "acc", "acc", "acc"]
    for k, (a, b) in enumerate(zip(compact, whole)):
        kind = (names * 2 + ["ll", "perp", "acc"] + names)[k]
        if kind == "acc":
            assert np.array_equal(a, b), k
        else:
            assert np.allclose(a, b, rtol=1e-12), k
    prior = o.ar_func_linear(o.one_hot([bytes(r).decode() for r in km]), p[0].detach().cpu().numpy())
    w = o.evaluation_step(te, prior, float(torch.tensor(0.37)), van, tr, rng=o.HashNoise(11, 0, n))     # (the float32 the call passes)
    assert np.isclose(compact[0], w[0], rtol=1e-11) and np.allclose(compact[2], w[2], rtol=1e-11)
    assert np.all(np.abs(compact[8] - w[5] / w[6]) < 1e-12)


def test_evaluation_with_the_fused_cnn_keeps_batches_in_kmer_order(tmp_path, monkeypatch):
    """An evaluation with the fused convolutional AR function sorts every resident batch by k-mer (its forward kernel evaluates a
    window that neighbouring contexts share once); the tie noise is keyed by table row, so every result equals the oracle's on the
    table in file order -- accuracies exactly -- and bear_ref's evaluation with the cnn net function does the same."""
    from util import sparse_table
    n = 5000
    tr, te, rf = sparse_table(n, 33)
    rng = np.random.default_rng(9)
    km = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 5))]
    path = tmp_path / "cnn_eval.tsv"
    dataloader.write_counts_tsv(str(path), km, np.stack([tr, te, rf]))
    data = dataloader.dataloader(str(path), "dna", 2000, 3)
    torch.manual_seed(3)
    f, p = ar_funcs.make_ar_func_cnn(5, 4, device="cuda", **CNN_CFG)
    assert f.fused and ar_funcs.wants_kmer_order(f) and not ar_funcs.wants_kmer_order(ar_funcs.make_ar_func_linear(5, 4, device="cuda")[0])
    orders = []
    orig = _train.sort_by_kmer
    monkeypatch.setattr(_train, "sort_by_kmer", lambda codes, lag: (orders.append(codes.shape[0]), orig(codes, lag))[1])
    van = np.array([0.1, 1.0, 10.0])
    got = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, van, seed=11)
    assert len(orders) == 3                                   # three batches, each sorted once
    prior = o.ar_func_cnn(o.one_hot([bytes(r).decode() for r in km]), [q.detach().cpu().numpy() for q in p])
    w = o.evaluation_step(te, prior, float(torch.tensor(0.37)), van, tr, rng=o.HashNoise(11, 0, n))
    assert np.isclose(got[0], w[0], rtol=1e-11) and np.isclose(got[1], w[1], rtol=1e-11) and np.allclose(got[2], w[2], rtol=1e-11)
    # (BEAR / AR accuracies: the prior rows come from the HIP kernel here and from NumPy in the oracle: a near-tie row may flip)
    assert abs(float(got[6]) - w[3] / w[6]) < 1e-3 and abs(float(got[7]) - w[4] / w[6]) < 1e-3
    assert np.all(np.abs(np.asarray(got[8]) - w[5] / w[6]) < 1e-12)     # vanilla models: integer concentrations + the same noise: exact
    # the order is not the result: the same evaluation over batches left in file order
    monkeypatch.setattr(ar_funcs, "wants_kmer_order", lambda f: False)
    plain = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, van, seed=11)
    assert len(orders) == 3
    for a, b in zip(got, plain):
        assert np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-12, atol=0)
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got[6:], plain[6:]))       # accuracies exactly
    # a batch that drops NO row (every row kept) is permuted all the same: the permutation is then its row_ids, or the noise would
    # be keyed by the sorted position (round-3 advisor finding)
    monkeypatch.undo()
    monkeypatch.setenv("BEAR_AMD_ALL_ROWS", "1")
    seen = []
    orig_eval = kernels.evaluate_planned
    monkeypatch.setattr(kernels, "evaluate_planned", lambda plan, *a, **kw: (seen.append((plan.test.shape[0], kw.get("row_ids") is not None)),
                                                                            orig_eval(plan, *a, **kw))[1])
    whole = bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.37), f, van, seed=11)
    assert [r for r, _ in seen] == [2000, 2000, 1000] and all(ids for _, ids in seen)
    for a, b in zip(got, whole):
        assert np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-12, atol=0)
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got[6:], whole[6:]))


def test_uploader_and_hbm_budget(monkeypatch, ysd1):
    """The upload path: pinned staging ring on a side stream -- bytes arrive intact across piece boundaries, for several tensors
    in flight; a table that cannot stay resident is refused with a clear error before anything is uploaded."""
    from bear_amd import _train
    dev = torch.device("cuda", 0)
    monkeypatch.setattr(_train.Uploader, "PIECE", 4096)          # many pieces, ring wrap-around
    up = _train.Uploader(dev, n_buffers=2)
    rng = np.random.default_rng(0)
    arrays = [rng.integers(0, 2 ** 31, size=(n, 5)).astype(np.int32) for n in (1, 203, 5000, 0, 12345)]
    arrays.append(rng.integers(0, 255, size=(7001, 13)).astype(np.uint8))
    outs = [up.put(a, torch.int32 if a.dtype == np.int32 else torch.uint8) for a in arrays]
    up.wait()
    for a, t in zip(arrays, outs):
        assert t.shape == a.shape and np.array_equal(t.cpu().numpy(), a)
    assert up.bytes == sum(a.nbytes for a in arrays)
    with pytest.raises(ValueError):
        up.put(arrays[0], torch.float64)
    # training through the pipelined upload (3 batches: worker thread + staging ring of tiny pieces) == the oracle-checked result
    data = dataloader.dataloader(YSD1, "dna", 500, 3)
    ls_small = []
    bear_ref.train(data.repeat(3), 1365, 3, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False, loss_save=ls_small)
    monkeypatch.undo()
    ls = []
    bear_ref.train(data.repeat(3), 1365, 3, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False, loss_save=ls)
    assert len(ls) == 9 and np.allclose(ls, ls_small, rtol=1e-13, atol=0)     # (a plan's item order -- the order of the sums -- is not fixed)
    # the budget: 1365 contexts do not fit a card with 1 MiB free
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (1 << 20, 288 << 30))
    with pytest.raises(MemoryError, match="shard the rows over more GPUs"):
        bear_ref.train(data.repeat(3), 1365, 3, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False)
    with pytest.raises(MemoryError):
        bear_net.evaluation(data, 0, 1, "dna", torch.tensor(0.4), ar_funcs.make_ar_func_linear(5, 4, device="cuda")[0], np.array([1.0]))


@pytest.mark.parametrize("which", ["bear_ref + stop", "bear_net + linear (fused)", "bear_net + cnn (fused)", "bear_net + cnn plugin (autograd)",
                                   "bear_ref + linear net (autograd)"])
def test_streamed_epochs_equal_resident_epochs(which, monkeypatch):
    """An epoch that does not stay in HBM is STREAMED (the reference's tf.data pipeline with cache=False, dataloader.py:36-50): a
    window of batches on the device, every batch re-uploaded, compacted, sorted and planned each epoch, the next one crossing PCIe
    while a step runs.  Same kernels on the same batches in the same order as the resident loop: same losses and parameters (to
    the order of a plan's sums), for every driver; the held-out evaluation likewise (accuracies exactly); and the switch happens
    by itself when the budget check refuses the whole epoch but takes the window."""
    data = dataloader.dataloader(YSD1, "dna", 300, 3)           # 5 batches per epoch, the last one short
    data.counts[0, ::4] = 0
    n, epochs = data.num_rows, 3
    seen = {}
    budget = _train.hbm_budget_check

    def train_once():
        torch.manual_seed(11)
        ls = []
        if which.startswith("bear_ref + stop"):
            out = bear_ref.train(data.repeat(epochs), n, epochs, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False, loss_save=ls)
        elif which.startswith("bear_ref"):
            out = bear_ref.train(data.repeat(epochs), n, epochs, 0, 2, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False, loss_save=ls)
        elif "linear" in which:
            out = bear_net.train(data.repeat(epochs), n, epochs, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False, loss_save=ls)
        elif "plugin" in which:
            out = bear_net.train(data.repeat(epochs), n, epochs, 0, "dna", 5, ar_funcs.make_ar_func_cnn,
                                 {"num_filters": 20, "filter_width": 3, "kmer_layer1_width": 16}, 0.01, "Adam", False, loss_save=ls)
        else:
            out = bear_net.train(data.repeat(epochs), n, epochs, 0, "dna", 5, ar_funcs.make_ar_func_cnn, CNN_CFG, 0.01, "Adam", False, loss_save=ls)
        return ls, np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in out[0]]), dict(_train.LAST_RUN), out
    resident = train_once()
    assert len(resident[0]) == 5 * epochs
    monkeypatch.setenv("BEAR_AMD_STREAM", "1")
    loads = []
    real_load = _train.ResidentBatches.load

    def counting_load(self, k):
        e = real_load(self, k)
        loads.append((k, self.loads, sum(b.get("_loaded", True) for b in self.batches)))
        return e
    monkeypatch.setattr(_train.ResidentBatches, "load", counting_load)
    streamed = train_once()
    assert streamed[2]["graph"] is False
    assert loads and max(x[2] for x in loads) == 1 and loads[-1][1] == 5 * epochs       # one batch resident at a time; every batch, every epoch
    assert np.allclose(streamed[0], resident[0], rtol=1e-11, atol=0)
    assert np.allclose(streamed[1], resident[1], rtol=1e-8, atol=1e-10)
    # the held-out evaluation over streamed batches
    if which.startswith("bear_net"):
        ev = lambda: bear_net.evaluation(data, 0, 1, "dna", torch.exp(streamed[3][1]).detach(), streamed[3][2], np.array([0.1, 1.0]))
    else:
        ev = lambda: bear_ref.evaluation(data, 0, 1, 2, "dna", torch.exp(streamed[3][1]).detach(), streamed[3][2], np.array([0.1, 1.0]))
    loads.clear()
    got = ev()
    assert len(loads) == 5
    monkeypatch.delenv("BEAR_AMD_STREAM")
    want = ev()
    for a, b in zip(got, want):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert np.allclose(a, b, rtol=1e-11, atol=0)
    for k in (6, 7, 8):                      # accuracies: counts of exactly decided arg-maxes over the same rows
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k]))
    # ... and without being asked: the whole epoch is refused, the window is not
    def tight(data_, n_columns, want_codes, device, rows=None, per_row_extra=0):
        seen.setdefault("rows", []).append(rows)
        if rows is not None and rows > 3 * 300:
            raise MemoryError("this rank's contexts need more HBM than is free: shard the rows over more GPUs")
        return budget(data_, n_columns, want_codes, device, rows=rows, per_row_extra=per_row_extra)
    monkeypatch.setattr(_train, "hbm_budget_check", tight)
    loads.clear()
    with pytest.warns(UserWarning, match="streaming the epoch"):
        auto = train_once()
    assert seen["rows"][:2] == [n, 900] and loads and loads[-1][1] == 5 * epochs
    assert np.allclose(auto[0], resident[0], rtol=1e-11, atol=0)


@pytest.mark.parametrize("driver", ["bear_ref + stop", "bear_net + linear"])
@pytest.mark.parametrize("train_ar", [False, True])
def test_one_launch_step_matches_the_two_launch_step(driver, train_ar, monkeypatch):
    """With one rank, Adam and no accumulation nothing sits between a step's reduce and its update, and the step is ONE launch:
    the last block of the reduce kernel runs tf.keras Adam behind its sums (`bear_*_train_step_f64`, bear_apply_in_block; the
    update's source is the two-launch form's).  Same losses and parameters as reduce + `bear_train_apply_f64` (to the order of the
    sums of a launch; bit for bit in the deterministic build: tests/test_deterministic_gpu.py); accumulation keeps two launches."""
    data = dataloader.dataloader(YSD1, "dna", 500, 3)

    def run(acc_steps=1):
        torch.manual_seed(4)
        ls = []
        if driver.startswith("bear_ref"):
            p, _, _ = bear_ref.train(data.repeat(20), 1365, 20, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", train_ar,
                                     acc_steps=acc_steps, loss_save=ls)
        else:
            p, _, _ = bear_net.train(data.repeat(20), 1365, 20, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", train_ar,
                                     acc_steps=acc_steps, loss_save=ls)
        return np.array(ls), np.concatenate([x.detach().cpu().numpy().reshape(-1) for x in p]), dict(_train.LAST_RUN)
    one = run()
    assert one[2]["one_launch_steps"] is True and one[2]["graph"] is True and len(one[0]) == 60
    monkeypatch.setenv("BEAR_AMD_TWO_LAUNCH_STEP", "1")
    two = run()
    assert two[2]["one_launch_steps"] is False
    assert np.allclose(one[0], two[0], rtol=1e-12, atol=0)
    assert np.allclose(one[1], two[1], rtol=1e-9, atol=1e-12)
    monkeypatch.delenv("BEAR_AMD_TWO_LAUNCH_STEP")
    assert run(acc_steps=3)[2]["one_launch_steps"] is False
    monkeypatch.setenv("BEAR_AMD_NO_GRAPH", "1")                 # the eager loop takes the one-launch step too
    eager = run()
    assert eager[2]["one_launch_steps"] is True and eager[2]["graph"] is False
    assert np.allclose(one[0], eager[0], rtol=1e-12, atol=0)


@pytest.mark.parametrize("kind", ["net", "ref"])
def test_run_config_driver(kind, ysd1):
    """bear_model/tests/test_run.py:12-51 re-stated: the bear_test.cfg workflow returns 1 and the train-set BMM
    log-likelihoods / perplexities equal the closed-form BMM marginals with alpha + epsilon."""
    from bear_amd.models import train_bear_net, train_bear_ref
    config = configparser.ConfigParser()
    config.read(os.path.join(ROOT, "bear_amd", "models", "config_files", "bear_test.cfg"))
    mod = train_bear_net if kind == "net" else train_bear_ref
    if kind == "ref":
        config["model"]["ar_func_name"] = "stop"
    exit_code, ll_van, perp_van = mod.main(config)
    assert exit_code == 1
    _, counts = ysd1
    data = dataloader.dataloader(YSD1, "dna", 2000, 3)
    alpha = np.array([0.1, 1.0, 10.0]) + 1e-7
    calc = dataloader.bmm_likelihood(data, alpha)
    train_liks = calc[0].numpy()
    assert np.allclose(train_liks, ll_van)
    assert np.allclose(np.exp(-train_liks / counts[:, 0].sum()), perp_van)
    assert np.allclose(train_liks, [-152712571.34208858, -152709051.39618373, -152745386.28243095], rtol=1e-12)
    out = config["results"]["out_folder"]
    assert os.path.exists(os.path.join(out, "results.pickle")) and os.path.exists(os.path.join(out, "config.cfg"))
    assert "heldout_perplex_BEAR" in config["results"] and "h" in config["results"]


def test_bmm_likelihood_reference_test(ysd1):
    # bear_model/tests/test_dataloader.py:34-49
    _, counts = ysd1
    data = dataloader.dataloader(YSD1, "dna", 2000, 3)
    alpha = np.array([0.1, 1.0, 10.0])
    true_liks = np.sum((np.sum(loggamma(counts[:, :, None, :] + alpha[:, None]), axis=-1)
                        - loggamma(np.sum(counts[:, :, None, :] + alpha[:, None], axis=-1)))
                       - (np.sum(loggamma(0 * counts[:, :, None, :] + alpha[:, None]), axis=-1)
                          - loggamma(np.sum(0 * counts[:, :, None, :] + alpha[:, None], axis=-1))), axis=0)
    assert np.allclose(true_liks, dataloader.bmm_likelihood(data, alpha).numpy(), rtol=1e-12)
    # exactly as the reference test calls it (test_dataloader.py:48)
    calc_liks = dataloader.bmm_likelihood(data.map(lambda kmers, counts: counts), alpha)
    assert np.allclose(true_liks, calc_liks.numpy(), rtol=1e-12)
    first = next(iter(data.map(lambda kmers, counts: counts)))
    assert tuple(first.shape) == (1365, 3, 5)


def test_core_distributions_reference_tests():
    # bear_model/tests/test_core.py:7-26 and 42-60 on the HIP path
    rng = np.random.default_rng(5)
    shape = np.array([3, 5])
    trans = rng.poisson(size=np.r_[shape, 5]).astype(float)
    total = trans.sum(-1)
    conc = rng.exponential(size=np.r_[shape[1], 5])
    dist = core.tfpDirichletMultinomialPerm(torch.tensor(total, device="cuda"), torch.tensor(conc, device="cuda"))
    assert np.all(dist._sample_n(7).cpu().numpy() == np.zeros(np.r_[7, shape, 5]))
    assert np.all(dist.ml_output().cpu().numpy() == np.argmax(conc, axis=-1))
    want = (np.sum(loggamma(conc + trans) - loggamma(conc), axis=-1) - (loggamma(conc.sum(-1) + total) - loggamma(conc.sum(-1))))
    assert np.allclose(dist.counts_log_prob(torch.tensor(trans)).cpu().numpy(), want, rtol=1e-12)
    probs = conc / conc.sum(-1, keepdims=True)
    md = core.tfpMultinomialPerm(torch.tensor(total, device="cuda"), torch.tensor(probs, device="cuda"))
    assert np.allclose(md.counts_log_prob(torch.tensor(trans)).cpu().numpy(), np.sum(np.log(probs) * trans, axis=-1))
    assert np.all(md.ml_output().cpu().numpy() == np.argmax(probs, axis=-1))
    # tie breaking (test_core.py:29-39): only tied maxima are ever returned, both of them
    d2 = core.tfpDirichletMultinomialPerm(torch.tensor([1.0], device="cuda"), torch.tensor([1.0, 0.5, 1.0], device="cuda"))
    seen = {int(d2.ml_output().item()) for _ in range(200)}
    assert seen == {0, 2}


def test_driver_optional_cache_and_shuffle_keys(tmp_path, ysd1):
    """[data] binary_cache / shuffle_seed (beyond the reference's keys): the run goes through, a cache file appears, and the
    table-level BMM results do not depend on the row order."""
    from bear_amd.models import train_bear_ref
    import shutil
    src = tmp_path / "data"
    src.mkdir()
    shutil.copy(YSD1, src / "tab_lag_5_file_0.tsv")
    config = configparser.ConfigParser()
    config.read(os.path.join(ROOT, "bear_amd", "models", "config_files", "bear_test.cfg"))
    config["model"]["ar_func_name"] = "stop"
    config["general"]["out_folder"] = str(tmp_path / "out") + "*"
    config["data"]["files_path"] = str(src)
    config["data"]["start_token"] = "tab_lag_5"
    config["data"]["binary_cache"] = str(tmp_path / "cache")
    config["data"]["shuffle_seed"] = "7"
    exit_code, ll_van, _ = train_bear_ref.main(config)
    assert exit_code == 1 and os.path.exists(tmp_path / "cache" / "tab_lag_5_file_0.tsv.bearcache")
    assert np.allclose(ll_van, [-152712571.34208858, -152709051.39618373, -152745386.28243095], rtol=1e-12)
    exit_code, ll_van2, _ = train_bear_ref.main(config)          # second run: served from the cache
    assert np.allclose(ll_van2, ll_van, rtol=1e-13)


@pytest.mark.parametrize("train_ar", [False, True])
def test_bear_ref_graph_replay_matches_oracle_loop(train_ar, ysd1):
    """One resident batch: bear_ref.train replays the captured step (bear_ref_train_step_f64 in a HIP graph) -- losses and
    parameters against the oracle loop, and against the eager path of the same build."""
    _, counts = ysd1
    data = dataloader.dataloader(YSD1, "dna", 1500, 3)           # one batch of 1365 rows per epoch
    steps = 40
    loss_save = []
    params, h_signed, _ = bear_ref.train(data.repeat(steps), 1365, steps, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01,
                                         "Adam", train_ar, loss_save=loss_save)
    p = np.array([0.0, np.log(1 / 30), -np.log(100)])
    m, v = np.zeros(3), np.zeros(3)
    want_loss = []
    for t in range(1, steps + 1):
        r = o.bear_ref_step(counts[:, 0], counts[:, 2], *p, train_ar=train_ar)
        want_loss.append(r["ll"])
        g = -np.array([r["d_h_signed"], r["d_tau_signed"], r["d_nu_signed"]])
        if train_ar:
            keras_adam_np(p[1:], g[1:], m[1:], v[1:], t)
        else:
            keras_adam_np(p, g, m, v, t)
    got = np.array([x.item() for x in params])
    assert len(loss_save) == steps and np.allclose(loss_save, want_loss, rtol=1e-10)
    assert np.allclose(got, p, rtol=1e-8, atol=1e-10) and h_signed is params[0]
    os.environ["BEAR_AMD_NO_GRAPH"] = "1"
    try:
        loss_eager = []
        params_e, _, _ = bear_ref.train(data.repeat(steps), 1365, steps, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01,
                                        "Adam", train_ar, loss_save=loss_eager)
    finally:
        os.environ.pop("BEAR_AMD_NO_GRAPH")
    assert np.allclose(loss_save, loss_eager, rtol=1e-12)
    assert np.allclose(got, [x.item() for x in params_e], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("train_ar", [False, True])
def test_bear_net_linear_graph_replay_matches_eager(train_ar, ysd1):
    """One resident batch + linear AR function: the captured step (bear_net_linear_train_step_f64 in a HIP graph) against
    the eager path of the same build (itself held to the oracle loop above)."""
    data = dataloader.dataloader(YSD1, "dna", 1500, 3)
    torch.manual_seed(4)
    _, init = ar_funcs.make_ar_func_linear(5, 4)
    restart = [np.array(0.1)] + [x.detach().numpy().copy() for x in init]
    steps = 25
    runs = []
    for no_graph in (False, True):
        if no_graph:
            os.environ["BEAR_AMD_NO_GRAPH"] = "1"
        try:
            ls = []
            params, h_signed, _ = bear_net.train(data.repeat(steps), 1365, steps, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01,
                                                 "Adam", train_ar, params_restart=restart, loss_save=ls)
        finally:
            os.environ.pop("BEAR_AMD_NO_GRAPH", None)
        runs.append((ls, [p.detach().cpu().numpy() for p in params]))
    assert len(runs[0][0]) == steps and np.allclose(runs[0][0], runs[1][0], rtol=1e-10)
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.allclose(a, b, rtol=1e-7, atol=1e-9)
    if train_ar:
        assert runs[0][1][0] == 0.1            # h_signed untouched in AR mode


@pytest.mark.parametrize("train_ar", [False, True])
def test_bear_net_cnn_graph_replay_matches_eager(train_ar, ysd1):
    """One resident batch + the reference's cnn config: the captured step (bear_net_cnn_train_step_f64 in a HIP graph) against
    the eager path (fused kernels behind torch autograd + host Adam, itself held to the CPU replica above)."""
    data = dataloader.dataloader(YSD1, "dna", 1500, 3)
    torch.manual_seed(6)
    _, init = ar_funcs.make_ar_func_cnn(5, 4, **CNN_CFG)
    restart = [np.array(0.1)] + [x.detach().numpy().copy() for x in init]
    steps = 12
    runs = []
    for no_graph in (False, True):
        if no_graph:
            os.environ["BEAR_AMD_NO_GRAPH"] = "1"
        try:
            ls = []
            params, _, _ = bear_net.train(data.repeat(steps), 1365, steps, 0, "dna", 5, ar_funcs.make_ar_func_cnn, CNN_CFG, 0.01,
                                          "Adam", train_ar, params_restart=restart, loss_save=ls)
        finally:
            os.environ.pop("BEAR_AMD_NO_GRAPH", None)
        runs.append((ls, [p.detach().cpu().numpy() for p in params]))
    assert len(runs[0][0]) == steps and np.allclose(runs[0][0], runs[1][0], rtol=1e-8)
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("which,acc_steps,train_ar", [("net cnn 20 filters", 1, False), ("net cnn 20 filters", 3, False),
                                                      ("net cnn 20 filters", 2, True), ("ref linear", 1, False), ("ref cnn", 2, False)])
def test_generic_plugin_step_graph_replay_matches_eager(which, acc_steps, train_ar, monkeypatch):
    """An AR function made of torch ops (make_ar_func_cnn with a shape the fused kernels do not take; bear_ref with a parametrised
    net function): the optimizer loop captures one period of steps -- the function's ops, the planned DM kernel, autograd's
    backward, the packed reduce, Keras' Adam with its counter on the device -- and replays it; losses and parameters equal the
    eager loop's to 1e-12 (the same operations in the same order), with and without gradient accumulation, two batches per epoch."""
    data = dataloader.dataloader(YSD1, "dna", 700, 3)
    steps, runs = 12, []
    for no_graph in (False, True):
        if no_graph:
            monkeypatch.setenv("BEAR_AMD_NO_GRAPH", "1")
        torch.manual_seed(4)
        ls = []
        if which.startswith("net"):
            params, _, _ = bear_net.train(data.repeat(steps), 1365, steps, 0, "dna", 5, ar_funcs.make_ar_func_cnn,
                                          {"num_filters": 20, "filter_width": 3, "kmer_layer1_width": 16}, 0.01, "Adam", train_ar,
                                          acc_steps=acc_steps, loss_save=ls)
        else:
            make, kw = (ar_funcs.make_ar_func_linear, {}) if "linear" in which else (ar_funcs.make_ar_func_cnn, CNN_CFG)
            params, _, _ = bear_ref.train(data.repeat(steps), 1365, steps, 0, 2, "dna", 5, make, kw, 0.01, "Adam", train_ar,
                                          acc_steps=acc_steps, loss_save=ls)
        how = dict(_train.LAST_RUN)
        base = math.lcm(2, acc_steps)
        if no_graph:
            assert how["graph"] is False and how["eager_steps"] == 2 * steps
        else:   # the first period runs eagerly (the libraries settle, the normalized_rows promise is checked), the rest is replayed
            assert how["graph"] is True and how["replays"] >= 1 and how["period"] % base == 0
            assert how["eager_steps"] == 2 * steps - how["replays"] * how["period"] and how["eager_steps"] >= base
        runs.append((ls, [p.detach().cpu().numpy() for p in params]))
    assert len(runs[0][0]) == 2 * steps // acc_steps
    assert np.allclose(runs[0][0], runs[1][0], rtol=1e-12, atol=0)
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.allclose(a, b, rtol=1e-12, atol=1e-14)


def test_cnn_step_over_live_contexts_equals_the_step_over_all_rows(ysd1, monkeypatch):
    """The CNN training step walks the plan's lists of contexts that hold counts (forward and backward skip the others: their
    gradient rows are zero).  Forcing the all-rows kernels (BEAR_CNN_BACKWARD=1) must give the same losses and parameters; the
    table gets extra all-zero training rows so that the lists really skip something."""
    data = dataloader.dataloader(YSD1, "dna", 1500, 3)
    n = data.num_rows
    data.counts[0, ::3] = 0                     # a third of the contexts without training counts
    torch.manual_seed(6)
    _, init = ar_funcs.make_ar_func_cnn(5, 4, **CNN_CFG)
    restart = [np.array(0.1)] + [x.detach().numpy().copy() for x in init]
    steps, runs = 8, []
    for force in (None, "1"):
        if force:
            monkeypatch.setenv("BEAR_CNN_BACKWARD", force)
        ls = []
        params, _, _ = bear_net.train(data.repeat(steps), n, steps, 0, "dna", 5, ar_funcs.make_ar_func_cnn, CNN_CFG, 0.01, "Adam", False,
                                      params_restart=restart, loss_save=ls)
        runs.append((ls, [p.detach().cpu().numpy() for p in params]))
    assert np.allclose(runs[0][0], runs[1][0], rtol=1e-10)
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.allclose(a, b, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("which", ["bear_ref + stop", "bear_net + linear (fused)", "bear_net + cnn (fused)"])
def test_training_leaves_out_the_contexts_without_training_counts(which, monkeypatch):
    """A resident training batch holds only the contexts with counts in the fitted column (the others add exactly nothing to the
    ELBO or to a gradient; the loss scale keeps the full batch size): the same losses and parameters as with every row kept
    (BEAR_AMD_ALL_ROWS=1) on a table where a third of the contexts hold none, also with two batches per epoch."""
    data = dataloader.dataloader(YSD1, "dna", 700, 3)
    data.counts[0, ::3] = 0
    n = data.num_rows
    torch.manual_seed(4)
    runs = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("BEAR_AMD_ALL_ROWS", env)
        ls = []
        torch.manual_seed(4)
        if which.startswith("bear_ref"):
            params, _, _ = bear_ref.train(data.repeat(3), n, 3, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False, loss_save=ls)
        elif "linear" in which:
            params, _, _ = bear_net.train(data.repeat(3), n, 3, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "Adam", False, loss_save=ls)
        else:
            params, _, _ = bear_net.train(data.repeat(3), n, 3, 0, "dna", 5, ar_funcs.make_ar_func_cnn, CNN_CFG, 0.01, "Adam", False, loss_save=ls)
        runs.append((ls, np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in params])))
    assert len(runs[0][0]) == 6
    assert np.allclose(runs[0][0], runs[1][0], rtol=1e-11)
    assert np.allclose(runs[0][1], runs[1][1], rtol=1e-8, atol=1e-10)


def test_cnn_step_on_a_dense_sorted_table_equals_its_three_kernels_over_all_rows():
    """bear_net_cnn_train_reduce_f64 on 3e5 contexts in k-mer order, as dense in k-mer space as the 1e8-context benchmark: the step
    walks the plan's lists of contexts with counts, tiles of consecutive list entries share their leading windows (per-window
    path, carried column sums, prefix taps).  Its packed [sum LL, d/dh, d/d params] must equal the three kernels run one
    after the other over ALL rows in RANDOM order (forward, planned DM step with gradient rows, backward)."""
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    lag, fw, n = 13, 8, 300_000
    gen = torch.Generator(dev).manual_seed(5)
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=gen)
    codes[:, :4] = torch.tensor([0, 1, 2, 3], dtype=torch.int8, device=dev)          # 4^9 k-mers behind a fixed prefix
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for l in range(lag):
        key = key * 6 + codes[:, l].to(torch.int64)
    codes = codes[torch.argsort(key)].contiguous()
    counts = kernels.synth_counts(7, 0, n, dev, want=("train",))["train"]
    torch.manual_seed(9)
    _, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev)
    flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
    h_s = -0.3
    theta = torch.cat([torch.tensor([h_s], dtype=torch.float64, device=dev), flat]).contiguous()
    packed_codes = kernels.pack_kmers(codes)
    plan = kernels.Plan(counts, 5)
    pk = torch.zeros(2 + flat.numel(), dtype=torch.float64, device=dev)
    kernels.net_cnn_train_reduce(plan, packed_codes, lag, fw, theta, kernels.cnn_step_buffers(n, lag, fw, dev), pk)
    perm = torch.randperm(n, device=dev, generator=gen)
    counts_p, codes_p = counts[perm].contiguous(), kernels.pack_kmers(codes[perm].contiguous())
    prior, t1 = kernels.cnn_forward(codes_p, flat, lag, fw)
    out, g = kernels.dm_prior_planned(kernels.Plan(counts_p, 5), prior, h_s, want_grad=True)
    grad = kernels.cnn_backward(codes_p, flat, lag, fw, t1, prior, g)
    want = torch.cat([out[:2], grad])
    assert abs(float(pk[0] - want[0])) <= 1e-12 * abs(float(want[0]))
    assert abs(float(pk[1] - want[1])) <= 1e-10 * max(abs(float(want[1])), 1.0)
    assert (pk[2:] - want[2:]).abs().max().item() <= 1e-10 * want[2:].abs().max().item()


@pytest.mark.parametrize("lag,fw,n,fixed,want_levels", [(13, 8, 300_000, 4, 5), (13, 8, 40_000, 6, 5), (7, 3, 50_000, 0, 4), (9, 9, 20_000, 0, 0),
                                                       (13, 8, 5_000, 0, 0), (5, 3, 3_000, 0, 2), (13, 8, 200_000, 3, None), (21, 8, 60_000, 15, 0), (11, 6, 60_000, 4, None)])
def test_cnn_step_over_prefix_levels_equals_the_plain_kernels(lag, fw, n, fixed, want_levels, monkeypatch):
    """bear_plan_attach_cnn_levels: with prefix levels the convolutional step evaluates a position once per distinct prefix of the
    sorted batch (forward: the levels' rows of layer-1 sums down to the contexts; backward: dT1 rows summed up the levels).  Its
    packed [sum LL, d/dh, d/d params] must equal the three plain kernels over all rows in RANDOM order, and the step without levels
    (BEAR_AMD_CNN_NO_LEVELS=1) -- on tables as dense in k-mer space as the benchmark, with duplicates, start symbols and unknown
    letters, on shapes with one position (no level possible), on a table too sparse for any prefix to repeat (none attached), on a
    table whose prefixes of lag - 1 letters hardly repeat while shorter ones do (that length is skipped: the contexts evaluate two
    positions) and on a shape whose backward pass does not fit the part form of the kernel (lag 21: none attached either)."""
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    gen = torch.Generator(dev).manual_seed(lag * 100 + fw)
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=gen)
    if fixed:
        codes[:, :fixed] = torch.randint(0, 4, (fixed,), dtype=torch.int8, device=dev, generator=gen)     # 4^(lag - fixed) k-mers behind one prefix
    odd = torch.randperm(n, device=dev, generator=gen)[:max(2, n // 200)]
    codes[odd[::2], 0] = 4                                             # start symbols at the front, an unknown letter somewhere
    codes[odd[1::2], torch.randint(0, lag, (odd[1::2].numel(),), device=dev, generator=gen)] = -1
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for l in range(lag):
        c = codes[:, l].to(torch.int64)
        key = key * 6 + torch.where(c >= 0, c, torch.full_like(c, 5))
    codes = codes[torch.argsort(key)].contiguous()
    counts = kernels.synth_counts(7, 0, n, dev, want=("train",))["train"]
    counts[:, 4] += (counts == 0).all(dim=1).to(counts.dtype)          # every context holds a count: the step takes plain groups of rows
    torch.manual_seed(9)
    _, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev)
    flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
    h_s = -0.3
    theta = torch.cat([torch.tensor([h_s], dtype=torch.float64, device=dev), flat]).contiguous()
    packed_codes = kernels.pack_kmers(codes)
    plan = kernels.Plan(counts, 5)
    bufs = kernels.cnn_step_buffers(n, lag, fw, dev)
    got_levels = plan.attach_cnn_levels(packed_codes, lag, fw)
    if want_levels is not None:
        assert got_levels == want_levels
    else:
        assert 1 <= got_levels <= lag - fw
    rows, letters = plan.cnn_level_rows(with_letters=True)
    assert len(rows) == got_levels and rows == sorted(rows, reverse=True) and letters == sorted(letters, reverse=True)
    if (lag, fw, n, fixed) == (13, 8, 200_000, 3):
        assert letters[0] < lag - 1                      # 12-letter prefixes (4^9 of them for 2e5 contexts) do not pay: skipped
    pk = torch.zeros(2 + flat.numel(), dtype=torch.float64, device=dev)
    for ar in (False, True):
        kernels.net_cnn_train_reduce(plan, packed_codes, lag, fw, theta, bufs, pk, train_ar=ar)
        with_levels = pk.clone()
        monkeypatch.setenv("BEAR_AMD_CNN_NO_LEVELS", "1")
        kernels.net_cnn_train_reduce(plan, packed_codes, lag, fw, theta, bufs, pk, train_ar=ar)
        monkeypatch.delenv("BEAR_AMD_CNN_NO_LEVELS")
        perm = torch.randperm(n, device=dev, generator=gen)
        counts_p, codes_p = counts[perm].contiguous(), kernels.pack_kmers(codes[perm].contiguous())
        prior, t1 = kernels.cnn_forward(codes_p, flat, lag, fw)
        out, g = kernels.dm_prior_planned(kernels.Plan(counts_p, 5), prior, h_s, want_grad=True, train_ar=ar)
        grad = kernels.cnn_backward(codes_p, flat, lag, fw, t1, prior, g)
        want = torch.cat([out[:2], grad])
        for name, got in (("levels", with_levels), ("plain step", pk)):
            assert abs(float(got[0] - want[0])) <= 1e-12 * abs(float(want[0])), (name, ar)
            assert abs(float(got[1] - want[1])) <= 1e-10 * max(abs(float(want[1])), 1.0), (name, ar)
            assert (got[2:] - want[2:]).abs().max().item() <= 1e-10 * want[2:].abs().max().item(), (name, ar)
    # another buffer with the same contents, another filter width: the plain step
    other = packed_codes.clone()
    kernels.net_cnn_train_reduce(plan, other, lag, fw, theta, bufs, pk, train_ar=True)
    assert (pk[2:] - want[2:]).abs().max().item() <= 1e-10 * want[2:].abs().max().item()


@pytest.mark.parametrize("which", ["bear_ref + linear net", "bear_ref + cnn net", "bear_net + SGD (torch loop)"])
def test_torch_ar_functions_only_see_contexts_with_counts(which, monkeypatch):
    """The autograd loops hand an AR function only the contexts that hold training counts (gather, scatter back): the same losses
    and parameters as with all rows (BEAR_AMD_ALL_ROWS=1) on a table where a third of the contexts hold none."""
    data = dataloader.dataloader(YSD1, "dna", 700, 3)         # two batches
    data.counts[0, ::3] = 0
    n, steps, runs = data.num_rows, 6, []
    for all_rows in (False, True):
        if all_rows:
            monkeypatch.setenv("BEAR_AMD_ALL_ROWS", "1")
        torch.manual_seed(9)
        ls = []
        if which.startswith("bear_ref"):
            make, kw = (ar_funcs.make_ar_func_linear, {}) if "linear" in which else (ar_funcs.make_ar_func_cnn, CNN_CFG)
            params, _, _ = bear_ref.train(data.repeat(steps), n, steps, 0, 2, "dna", 5, make, kw, 0.01, "Adam", False, loss_save=ls)
        else:
            params, _, _ = bear_net.train(data.repeat(steps), n, steps, 0, "dna", 5, ar_funcs.make_ar_func_linear, {}, 0.01, "SGD", False,
                                          loss_save=ls)
        runs.append((ls, [p.detach().cpu().numpy() for p in params]))
    assert len(runs[0][0]) == 2 * steps and np.allclose(runs[0][0], runs[1][0], rtol=1e-11)
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.allclose(a, b, rtol=1e-8, atol=1e-10)


def test_wrong_normalized_rows_promise_is_refused():
    """A plugin that sets ``normalized_rows`` on rows that do not sum to one (bear_net.py takes any ar_funcs.make_ar_func_<name>):
    train() checks the promise once, on the first batch, and raises -- the normalised kernel would have returned wrong sums
    silently.  Without the attribute the same function trains through the general kernel."""
    data = dataloader.dataloader(YSD1, "dna", 700, 3)

    def make_unnormalised(promise):
        def make(lag, alphabet_size, dtype=torch.float64, device=None):
            w = torch.zeros(alphabet_size + 1, dtype=dtype, device=device, requires_grad=True)

            def ar_func(codes):
                return (torch.softmax(w, 0) * 1.001).expand(codes.shape[0], alphabet_size + 1)
            if promise:
                ar_func.normalized_rows = True
            return ar_func, [w]
        return make
    with pytest.raises(ValueError, match="normalized_rows"):
        bear_net.train(data.repeat(2), data.num_rows, 2, 0, "dna", 5, make_unnormalised(True), {}, 0.01, "Adam", False)
    ls = []
    bear_net.train(data.repeat(2), data.num_rows, 2, 0, "dna", 5, make_unnormalised(False), {}, 0.01, "Adam", False, loss_save=ls)
    assert len(ls) == 4 and np.all(np.isfinite(ls))


def test_graph_path_feeds_the_writer(ysd1):
    class W:
        def __init__(self):
            self.rows = []

        def add_scalar(self, name, value, step):
            self.rows.append((name, value, step))
    data = dataloader.dataloader(YSD1, "dna", 700, 3)
    w, ls = W(), []
    bear_ref.train(data.repeat(3), 1365, 3, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False, writer=w, loss_save=ls)
    assert [r[2] for r in w.rows] == [1, 2, 3, 4, 5, 6] and [r[1] for r in w.rows] == ls and w.rows[0][0] == "elbo"


def _oracle_ref_loop(counts, batch, epochs, acc_steps, optimizer, train_ar, lr):
    """bear_ref.train with the stop net function restated on the oracle: gradients SUMMED over acc_steps batches, the logged
    ELBO averaged (bear_ref.py:256-258, 369-381)."""
    p = np.array([0.0, np.log(1 / 30), -np.log(100)])
    m, v = np.zeros(3), np.zeros(3)
    acc, loss, step, t, logged = np.zeros(3), 0.0, 1, 0, []
    n = len(counts)
    for _ in range(epochs):
        for a in range(0, n, batch):
            b = min(a + batch, n)
            r = o.bear_ref_step(counts[a:b, 0], counts[a:b, 2], *p, train_ar=train_ar)
            scale = -(n / (b - a))
            loss += scale * r["ll"]
            acc += scale * np.array([r["d_h_signed"], r["d_tau_signed"], r["d_nu_signed"]])
            if step % acc_steps == 0:
                logged.append(-loss / acc_steps)
                t += 1
                k0 = 1 if train_ar else 0
                if optimizer == "Adam":
                    keras_adam_np(p[k0:], acc[k0:], m[k0:], v[k0:], t, lr=lr)
                else:
                    p[k0:] -= lr * acc[k0:]
                acc[:] = 0.0
                loss = 0.0
            step += 1
    return logged, p


@pytest.mark.parametrize("optimizer,acc_steps,train_ar", [("Adam", 3, False), ("Adam", 2, True), ("SGD", 1, False), ("SGD", 3, False)])
def test_accumulation_and_sgd_match_oracle_loop(optimizer, acc_steps, train_ar, ysd1):
    """acc_steps > 1 (gradients summed, logged ELBO averaged: bear_net.py:193-196, 303-309) and a non-Adam optimizer, on the
    device-resident loop of bear_ref.train; 3 batches per epoch x 3 epochs = 9 batch steps (a trailing partial accumulation
    is dropped, as in the reference)."""
    _, counts = ysd1
    lr = 0.01 if optimizer == "Adam" else 1e-9      # plain SGD on a loss of order 1e8
    data = dataloader.dataloader(YSD1, "dna", 500, 3)
    ls = []
    params, _, _ = bear_ref.train(data.repeat(3), 1365, 3, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, lr, optimizer, train_ar,
                                  acc_steps=acc_steps, loss_save=ls)
    want_loss, want_p = _oracle_ref_loop(counts, 500, 3, acc_steps, optimizer, train_ar, lr)
    assert len(ls) == 9 // acc_steps and np.allclose(ls, want_loss, rtol=1e-10)
    assert np.allclose([x.item() for x in params], want_p, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("name,kw", [("linear", {}), ("cnn", {"filter_width": 3, "num_filters": 5}), ("cnn", CNN_CFG)])
def test_accumulation_through_every_bear_net_step(name, kw, ysd1):
    """acc_steps = 2 through the three bear_net step implementations -- fused linear kernel and fused cnn kernels (device-resident
    theta, _train.run_device_steps) and a torch-op AR function with autograd (num_filters = 5 is outside the fused kernel,
    _train.run_autograd_steps) -- against a CPU replica: torch autograd through the plugin, likelihood and row gradients from the
    oracle, gradients summed over two batches, logged ELBO averaged (bear_net.py:193-196, 303-309)."""
    _, counts = ysd1
    data = dataloader.dataloader(YSD1, "dna", 400, 3)             # 4 batches per epoch
    make = getattr(ar_funcs, "make_ar_func_" + name)
    torch.manual_seed(8)
    _, init = make(5, 4, **kw)
    init_np = [x.detach().numpy().copy() for x in init]
    ls = []
    params, _, _ = bear_net.train(data.repeat(2), 1365, 2, 0, "dna", 5, make, kw, 0.01, "Adam", False, acc_steps=2,
                                  params_restart=[np.array(0.1)] + init_np, loss_save=ls)
    f_cpu, p_cpu = make(5, 4, **kw)
    with torch.no_grad():
        for a, b in zip(p_cpu, init_np):
            a.copy_(torch.as_tensor(b))
    h = np.array(0.1)
    arrs = [h] + [q.detach().numpy() for q in p_cpu]
    ms, vs, acc = ([np.zeros_like(x) for x in arrs] for _ in range(3))
    codes = torch.as_tensor(data.codes())
    want, loss, step, t = [], 0.0, 1, 0
    for _ in range(2):
        for a in range(0, 1365, 400):
            b = min(a + 400, 1365)
            for q in p_cpu:
                q.grad = None
            prior = f_cpu(codes[a:b])
            r = o.bear_net_step(counts[a:b, 0], prior.detach().numpy(), float(h))
            scale = -(1365 / (b - a))
            prior.backward(torch.as_tensor(scale * r["d_prior"]))
            loss += scale * r["ll"]
            acc[0] += scale * r["d_h_signed"]
            for i, q in enumerate(p_cpu):
                acc[i + 1] += q.grad.numpy()
            if step % 2 == 0:
                want.append(-loss / 2)
                t += 1
                for i in range(len(arrs)):
                    keras_adam_np(arrs[i], acc[i], ms[i], vs[i], t)
                    acc[i][...] = 0.0
                loss = 0.0
            step += 1
    assert len(ls) == 4 and np.allclose(ls, want, rtol=1e-9)
    assert np.isclose(params[0].item(), float(h), rtol=1e-7, atol=1e-9)
    for got, w in zip(params[1:], arrs[1:]):
        assert np.allclose(got.detach().cpu().numpy(), w, rtol=1e-6, atol=1e-8)


def test_graph_capture_failure_falls_back_to_the_eager_loop(monkeypatch, ysd1):
    """Stream capture unavailable -> the same kernels are enqueued eagerly (a warning, identical results); library errors
    raised inside a capture are NOT swallowed."""
    data = dataloader.dataloader(YSD1, "dna", 1500, 3)
    args = (data.repeat(5), 1365, 5, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False)
    ls_graph = []
    p_graph, _, _ = bear_ref.train(*args, loss_save=ls_graph)

    class Boom:
        def __init__(self, *a, **k):
            raise RuntimeError("capture refused (test)")
    monkeypatch.setattr(torch.cuda, "graph", Boom)
    ls = []
    with pytest.warns(UserWarning, match="capture"):
        p, _, _ = bear_ref.train(*args, loss_save=ls)
    # same kernels, same order of steps; the ticketed work distribution inside a launch reorders the fp64 sums by an ulp
    assert np.allclose(ls, ls_graph, rtol=1e-13) and np.allclose([x.item() for x in p], [x.item() for x in p_graph], rtol=1e-11)
    from bear_amd import _lib

    class Bad:
        def __init__(self, *a, **k):
            raise _lib.BearError(-1, "test")
    monkeypatch.setattr(torch.cuda, "graph", Bad)
    with pytest.raises(_lib.BearError):
        bear_ref.train(*args)


def test_unsupported_optimizer_is_a_clear_error():
    data = dataloader.dataloader(YSD1, "dna", 1500, 3)
    with pytest.raises(ValueError, match="Ftrl"):
        bear_ref.train(data.repeat(2), 1365, 2, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Ftrl", False)


def test_float32_precision_runs_in_float64():
    """`precision = float32` of the reference configs (models/train_bear_net.py:58): accepted, with a warning, and carried out
    in float64 -- the same losses and parameters as the float64 run."""
    data = dataloader.dataloader(YSD1, "dna", 1500, 3)
    runs = []
    for dt in (torch.float64, torch.float32):
        ls = []
        if dt == torch.float32:
            with pytest.warns(UserWarning, match="float64"):
                p, _, _ = bear_ref.train(data.repeat(3), 1365, 3, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False,
                                         loss_save=ls, dtype=dt)
        else:
            p, _, _ = bear_ref.train(data.repeat(3), 1365, 3, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False,
                                     loss_save=ls, dtype=dt)
        runs.append((ls, [float(x.detach()) for x in p[:3]]))
    assert np.allclose(runs[0][0], runs[1][0], rtol=1e-12) and np.allclose(runs[0][1], runs[1][1], rtol=1e-10)   # (summation order varies)
    with pytest.raises(NotImplementedError):
        bear_ref.train(data.repeat(1), 1365, 1, 0, 2, "dna", 5, ar_funcs.make_ar_func_stop, {}, 0.01, "Adam", False, dtype=torch.float16)


def test_c_example_compiles_and_runs(tmp_path):
    """examples/ref_step.c: a plain-C caller of the ABI (no Python, no torch) -- compiled with gcc against include/bear_hip.h and
    run; its sums are checked against the C oracle on the same synthetic table."""
    import subprocess
    import c_oracle as co
    from bear_amd import kernels
    exe = tmp_path / "ref_step"
    cmd = ["gcc", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "ref_step.c"), "-L" + os.path.join(ROOT, "bear_amd"), "-lbear_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + os.path.join(ROOT, "bear_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    n = 300_000
    p = subprocess.run([str(exe), str(n)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    import re
    vals = [float(x) for x in re.findall(r"= (-?\d\.\d+e[+-]\d+)", p.stdout)]
    assert len(vals) == 4, p.stdout
    t = kernels.synth_counts(20211012, 0, n, torch.device("cuda"), want=("train", "ref"))
    want = co.dm_ref(t["train"].cpu().numpy().view(np.uint32), t["ref"].cpu().numpy().view(np.uint32), 0.0, np.log(1 / 30), -np.log(100))
    assert np.isclose(vals[0], want[0], rtol=1e-11)            # printed with 12 digits
    assert np.allclose(vals[1:], want[1:], rtol=1e-5)          # printed with 6 digits


@pytest.mark.parametrize("lag,fw,n,fixed", [(10, 4, 300_000, 0), (13, 8, 2_400_000, 5), (7, 3, 50_000, 0), (8, 2, 100_000, 0)])
def test_cnn_step_over_window_tables_equals_the_plain_kernels(lag, fw, n, fixed, monkeypatch):
    """Window tables (round 5): a level's own positions evaluated once per distinct filter_width-letter window, rows gathering their
    window's row forward and summed by window backward.  The step with them == the step with prefix levels alone
    (BEAR_AMD_CNN_NO_WINDOWS at attach time) == the plain step (BEAR_AMD_CNN_NO_LEVELS) == the three kernels over all rows in random
    order -- on tables with duplicates, start symbols and unknown letters; shapes with tables at several levels, with several
    tables at level 0 (a table too sparse for prefix levels), with two-letter windows (36 rows per table)."""
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    gen = torch.Generator(dev).manual_seed(lag * 131 + fw)
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=gen)
    if fixed:
        codes[:, :fixed] = torch.randint(0, 4, (fixed,), dtype=torch.int8, device=dev, generator=gen)
    odd = torch.randperm(n, device=dev, generator=gen)[:max(2, n // 150)]
    codes[odd[::2], 0] = 4
    codes[odd[1::2], torch.randint(0, lag, (odd[1::2].numel(),), device=dev, generator=gen)] = -1
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for l in range(lag):
        c = codes[:, l].to(torch.int64)
        key = key * 6 + torch.where(c >= 0, c, torch.full_like(c, 5))
    codes = codes[torch.argsort(key)].contiguous()
    counts = kernels.synth_counts(11, 0, n, dev, want=("train",))["train"]
    counts[:, 4] += (counts == 0).all(dim=1).to(counts.dtype)
    torch.manual_seed(5)
    _, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev)
    flat = torch.cat([q.detach().reshape(-1) for q in params]).contiguous()
    h_s = 0.2
    theta = torch.cat([torch.tensor([h_s], dtype=torch.float64, device=dev), flat]).contiguous()
    packed_codes = kernels.pack_kmers(codes)
    plan = kernels.Plan(counts, 5)
    bufs = kernels.cnn_step_buffers(n, lag, fw, dev, ws=plan.ws)
    plan.attach_cnn_levels(packed_codes, lag, fw)
    tables = plan.cnn_window_rows()
    levels = plan.cnn_level_rows()
    assert tables, (levels, tables)
    assert all(0 <= lev <= len(levels) and 0 <= pos <= lag - fw and rows >= 1 for lev, pos, rows in tables)
    assert all(rows * 8 <= ([n] + levels)[lev] for lev, pos, rows in tables)       # the attach rule: >= 8 rows per distinct window
    assert len({(lev, pos) for lev, pos, _ in tables}) == len(tables)
    pk = torch.zeros(2 + flat.numel(), dtype=torch.float64, device=dev)
    perm = torch.randperm(n, device=dev, generator=gen)
    counts_p, codes_p = counts[perm].contiguous(), kernels.pack_kmers(codes[perm].contiguous())
    for ar in (False, True):
        kernels.net_cnn_train_reduce(plan, packed_codes, lag, fw, theta, bufs, pk, train_ar=ar)
        with_tables = pk.clone()
        monkeypatch.setenv("BEAR_AMD_CNN_KEEP_T1", "1")          # the contexts' layer-1 rows stored by the forward pass instead of put together again
        kernels.net_cnn_train_reduce(plan, packed_codes, lag, fw, theta, bufs, pk, train_ar=ar)
        monkeypatch.delenv("BEAR_AMD_CNN_KEEP_T1")
        kept_t1 = pk.clone()
        monkeypatch.setenv("BEAR_AMD_CNN_NO_HEAD_KERNEL", "1")   # ... and the part kernel's head instead of the head-only kernel
        kernels.net_cnn_train_reduce(plan, packed_codes, lag, fw, theta, bufs, pk, train_ar=ar)
        monkeypatch.delenv("BEAR_AMD_CNN_NO_HEAD_KERNEL")
        part_head = pk.clone()
        monkeypatch.setenv("BEAR_AMD_CNN_NO_LEVELS", "1")
        kernels.net_cnn_train_reduce(plan, packed_codes, lag, fw, theta, bufs, pk, train_ar=ar)
        monkeypatch.delenv("BEAR_AMD_CNN_NO_LEVELS")
        plain = pk.clone()
        prior, t1 = kernels.cnn_forward(codes_p, flat, lag, fw)
        out, g = kernels.dm_prior_planned(kernels.Plan(counts_p, 5), prior, h_s, want_grad=True, train_ar=ar)
        want = torch.cat([out[:2], kernels.cnn_backward(codes_p, flat, lag, fw, t1, prior, g)])
        for name, got in (("window tables", with_tables), ("stored layer-1 rows", kept_t1), ("part kernel's head", part_head), ("plain step", plain)):
            assert abs(float(got[0] - want[0])) <= 1e-12 * abs(float(want[0])), (name, ar)
            assert abs(float(got[1] - want[1])) <= 1e-10 * max(abs(float(want[1])), 1.0), (name, ar)
            assert (got[2:] - want[2:]).abs().max().item() <= 1e-10 * want[2:].abs().max().item(), (name, ar)
    # the forward pass alone over the same levels and tables (evaluation-style callers): the prior rows of the plain kernel
    prior_l, t1_l = kernels.cnn_forward(packed_codes, flat, lag, fw, plan=plan)
    prior_p, t1_p = kernels.cnn_forward(packed_codes, flat, lag, fw)
    assert torch.allclose(prior_l, prior_p, rtol=1e-11, atol=1e-300) and torch.allclose(t1_l, t1_p, rtol=1e-10, atol=1e-10)
    # prefix levels alone
    monkeypatch.setenv("BEAR_AMD_CNN_NO_WINDOWS", "1")
    plan.attach_cnn_levels(packed_codes, lag, fw)
    monkeypatch.delenv("BEAR_AMD_CNN_NO_WINDOWS")
    assert plan.cnn_window_rows() == []
    kernels.net_cnn_train_reduce(plan, packed_codes, lag, fw, theta, bufs, pk, train_ar=True)
    assert (pk[2:] - want[2:]).abs().max().item() <= 1e-10 * want[2:].abs().max().item()

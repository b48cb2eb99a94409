"""GPU parity tests of the fused convolutional AR function (bear_cnn_forward_f64 / bear_cnn_backward_f64, through the
C ABI): forward against the oracle's restatement of ar_funcs.py:91-97, backward against torch fp64 autograd of the
same formulas (the torch reference kept for this floating-point kernel).  Tolerances: rows 1e-12 relative; parameter
gradients 1e-10 of the gradient's largest entry per tensor."""
import numpy as np
import pytest
import torch

import bear_oracle as o
from bear_amd import ar_funcs, core, kernels

pytestmark = pytest.mark.gpu

ROW_RTOL = 1e-12
GRAD_RTOL = 1e-10


def _random_kmers(n, lag, rng, with_specials=True):
    letters = np.array(list("ACGT"))
    km = ["".join(rng.choice(letters, size=lag)) for _ in range(n)]
    if with_specials and n >= 8:
        km[0] = "[" * lag
        km[1] = "[" * (lag - 1) + "A"
        km[2] = "N" + km[2][1:]            # unknown letter: all-zero one-hot row (core.py:173)
        km[3] = km[3][:-1] + "x"
    return km


def _make(lag, fw, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    ar_func, params = ar_funcs.make_ar_func_cnn(lag, 4, filter_width=fw, device=dev, generator=g)
    with torch.no_grad():                 # move every parameter off its initial value (ones / zeros)
        for p in params:
            p.add_(0.3 * torch.randn(p.shape, dtype=p.dtype, device=dev, generator=g))
    return ar_func, params


@pytest.mark.parametrize("lag,fw,n", [(13, 8, 1000), (5, 3, 300), (21, 21, 130), (4, 1, 70), (1, 1, 5)])
def test_cnn_forward_matches_oracle(lag, fw, n):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(lag * 100 + fw)
    kmers = _random_kmers(n, lag, rng)
    ar_func, params = _make(lag, fw, dev, 3)
    codes = torch.from_numpy(core.encode_kmers(kmers, "dna")).to(dev)
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    assert flat.numel() == kernels.cnn_param_count(lag, fw)
    prior, t1 = kernels.cnn_forward(kernels.pack_kmers(codes), flat, lag, fw)
    want = o.ar_func_cnn(o.one_hot(kmers, "dna"), [p.detach().cpu().numpy() for p in params])
    assert np.allclose(prior.cpu().numpy(), want, rtol=ROW_RTOL, atol=1e-300)
    assert t1.shape == (n, 16)


def _drifting_prefix_kmers(n, lag, rng, block=200):
    """Contexts as a k-mer-sorted batch holds them: blocks share everything but their last letters; the prefix drifts slowly and
    may hold start symbols and letters outside the alphabet."""
    letters = np.array(list("ACGT[N"))
    prefix = letters[rng.choice(6, size=lag, p=[0.23, 0.23, 0.23, 0.23, 0.05, 0.03])]
    tail = max(1, min(3, lag - 1))
    kmers = []
    for k in range(n):
        if k % block == 0 and k:
            prefix = prefix.copy()
            prefix[rng.integers(0, lag - tail)] = letters[rng.integers(0, 6)]
        row = prefix.copy()
        row[lag - tail:] = letters[rng.integers(0, 4, size=tail)]
        kmers.append("".join(row))
    return kmers


@pytest.mark.parametrize("lag,fw,n", [(13, 8, 3000), (9, 3, 1000), (5, 5, 777), (13, 8, 40)])
def test_cnn_backward_shared_windows_in_sorted_contexts(lag, fw, n):
    """The backward kernel does a position whose window a whole tile of 32 contexts shares ONCE, from the column sums of dT1
    (everything it adds is linear in dT1): gradients against torch autograd of the one-hot formulation on contexts with long
    common prefixes (start symbols and unknown letters inside them), and == the gradients of the same contexts in random order."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(lag * 17 + fw + n)
    kmers = _drifting_prefix_kmers(n, lag, rng, block=70)
    ar_func, params = _make(lag, fw, dev, 11)
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    codes = torch.from_numpy(core.encode_kmers(kmers, "dna")).to(dev)
    packed = kernels.pack_kmers(codes)
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    grad_rows = torch.from_numpy(rng.standard_normal((n, 5)) * np.exp(rng.normal(size=(n, 1)))).to(dev)
    grad_rows[rng.random(n) < 0.3] = 0.0               # contexts without counts
    got = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, grad_rows)
    rows = ar_func(core.tf_one_hot(kmers, "dna", device=dev))
    rows.backward(grad_rows)
    want = torch.cat([p.grad.reshape(-1) for p in params])
    k = 0
    for p in params:
        m = p.numel()
        assert (got[k:k + m] - want[k:k + m]).abs().max().item() <= GRAD_RTOL * max(want[k:k + m].abs().max().item(), 1e-30), (p.shape,)
        k += m
    perm = torch.from_numpy(rng.permutation(n)).to(dev)
    got_p = kernels.cnn_backward(packed[perm].contiguous(), flat, lag, fw, t1[perm].contiguous(), prior[perm].contiguous(),
                                 grad_rows[perm].contiguous())
    assert (got_p - got).abs().max().item() <= 1e-11 * got.abs().max().item()


def test_cnn_dense_sorted_table_matches_autograd():
    """A table as dense in k-mer space as the 1e8-context benchmark (2e5 contexts over 4^9 k-mers behind a fixed prefix), rows in
    k-mer order: whole runs of tiles share the windows of the leading positions (their column sums are carried from tile to
    tile), the next position has a few windows per tile, the last ones take the per-context path with most taps inside the
    common prefix.  Prior rows against the torch formulation, gradients against its autograd, and == random order."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2024)
    lag, fw, n = 13, 8, 200_000
    letters = np.array(list("ACGT"))
    body = rng.integers(0, 4, size=(n, 9))
    key = (body * (4 ** np.arange(8, -1, -1))).sum(1)
    body = body[np.argsort(key, kind="stable")]
    kmers = ["ACGT" + "".join(letters[r]) for r in body]
    kmers[:40] = ["[[[[" + k[4:] for k in kmers[:40]]          # a few sequence starts
    ar_func, params = _make(lag, fw, dev, 3)
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    codes = torch.from_numpy(core.encode_kmers(kmers, "dna")).to(dev)
    packed = kernels.pack_kmers(codes)
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    grad_rows = torch.from_numpy(rng.standard_normal((n, 5)) * np.exp(rng.normal(size=(n, 1)))).to(dev)
    grad_rows[torch.from_numpy(rng.random(n) < 0.3).to(dev)] = 0.0
    got = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, grad_rows)
    rows = ar_func(core.tf_one_hot(kmers, "dna", device=dev))
    assert torch.allclose(rows, prior, rtol=1e-11, atol=0)
    rows.backward(grad_rows)
    k = 0
    for p in params:
        m = p.numel()
        want = p.grad.reshape(-1)
        assert (got[k:k + m] - want).abs().max().item() <= GRAD_RTOL * max(want.abs().max().item(), 1e-30), (p.shape,)
        k += m
    perm = torch.from_numpy(rng.permutation(n)).to(dev)
    got_p = kernels.cnn_backward(packed[perm].contiguous(), flat, lag, fw, t1[perm].contiguous(), prior[perm].contiguous(),
                                 grad_rows[perm].contiguous())
    assert (got_p - got).abs().max().item() <= 1e-11 * got.abs().max().item()


@pytest.mark.parametrize("lag,fw", [(13, 8), (9, 3), (5, 5)])
def test_cnn_forward_shared_windows_in_sorted_contexts(lag, fw):
    """k-mer-sorted contexts with long common prefixes (what bear_net.train uploads): whole waves share the windows inside the
    prefix and the forward kernel evaluates those once per wave; rows against the oracle, also with start symbols and unknown
    letters inside the shared part, and == the rows of the same contexts in random order."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(lag * 31 + fw)
    n = 3000
    letters = np.array(list("ACGT[N"))
    prefix = letters[rng.choice(6, size=lag, p=[0.23, 0.23, 0.23, 0.23, 0.05, 0.03])]
    tail = max(1, min(3, lag - 1))
    kmers = []
    for k in range(n):                     # blocks of ~200 contexts share everything but their last letters; the prefix drifts slowly
        if k % 200 == 0 and k:
            prefix = prefix.copy()
            prefix[rng.integers(0, lag - tail)] = letters[rng.integers(0, 6)]
        row = prefix.copy()
        row[lag - tail:] = letters[rng.integers(0, 4, size=tail)]
        kmers.append("".join(row))
    ar_func, params = _make(lag, fw, dev, 7)
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    codes = torch.from_numpy(core.encode_kmers(kmers, "dna")).to(dev)
    prior, t1 = kernels.cnn_forward(kernels.pack_kmers(codes), flat, lag, fw)
    want = o.ar_func_cnn(o.one_hot(kmers, "dna"), [p.detach().cpu().numpy() for p in params])
    assert np.allclose(prior.cpu().numpy(), want, rtol=ROW_RTOL, atol=1e-300)
    perm = torch.from_numpy(rng.permutation(n)).to(dev)
    prior_p, t1_p = kernels.cnn_forward(kernels.pack_kmers(codes[perm].contiguous()), flat, lag, fw)
    assert torch.allclose(prior_p, prior[perm], rtol=1e-12, atol=0) and torch.allclose(t1_p, t1[perm], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("form", ["32-context tiles, two waves per SIMD", "64-context tiles, one wave per SIMD"])
@pytest.mark.parametrize("lag,fw,n", [(13, 8, 3000), (5, 3, 1365), (7, 7, 64), (2, 1, 3), (21, 1, 500), (21, 16, 200), (13, 8, 31), (13, 8, 33)])
def test_cnn_backward_matches_torch_autograd(lag, fw, n, form, monkeypatch):
    """Both tile forms of the backward kernel (the library picks the first when its LDS fits: not for (21, 16); BEAR_CNN_BACKWARD=1
    forces the second)."""
    if form.startswith("64"):
        monkeypatch.setenv("BEAR_CNN_BACKWARD", "1")
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(lag * 7 + fw)
    kmers = _random_kmers(n, lag, rng)
    ar_func, params = _make(lag, fw, dev, 5)
    codes = torch.from_numpy(core.encode_kmers(kmers, "dna")).to(dev)
    packed = kernels.pack_kmers(codes)
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    grad_rows = torch.from_numpy(rng.standard_normal((n, 5)) * np.exp(rng.normal(size=(n, 1)))).to(dev)
    got = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, grad_rows)
    # torch reference: the one-hot formulation (conv1d / tensordot), autograd
    rows = ar_func(core.tf_one_hot(kmers, "dna", device=dev))
    assert torch.allclose(rows, prior, rtol=1e-11, atol=0)
    rows.backward(grad_rows)
    k = 0
    for p in params:
        want = p.grad.reshape(-1)
        g = got[k:k + want.numel()]
        k += want.numel()
        assert (g - want).abs().max().item() <= GRAD_RTOL * max(want.abs().max().item(), 1e-30), (p.shape,)
    assert k == got.numel()


def test_cnn_backward_is_linear_and_shard_additive():
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    lag, fw, n = 13, 8, 5000
    codes = torch.from_numpy(rng.integers(0, 4, size=(n, lag)).astype(np.int8)).to(dev)
    packed = kernels.pack_kmers(codes)
    _, params = _make(lag, fw, dev, 1)
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    g = torch.from_numpy(rng.standard_normal((n, 5))).to(dev)
    full = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, g)
    a = kernels.cnn_backward(packed[:2000].contiguous(), flat, lag, fw, t1[:2000].contiguous(), prior[:2000].contiguous(),
                             g[:2000].contiguous())
    b = kernels.cnn_backward(packed[2000:].contiguous(), flat, lag, fw, t1[2000:].contiguous(), prior[2000:].contiguous(),
                             g[2000:].contiguous())
    scale = full.abs().max().item()
    assert (full - (a + b)).abs().max().item() <= 1e-12 * scale
    twice = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, 2.0 * g)
    assert (twice - 2.0 * full).abs().max().item() <= 1e-12 * scale
    empty = kernels.cnn_backward(packed[:0].contiguous(), flat, lag, fw, t1[:0].contiguous(), prior[:0].contiguous(),
                                 g[:0].contiguous())
    assert torch.count_nonzero(empty).item() == 0


def test_cnn_full_size_properties():
    """BASELINE configs[2]/[4] size (1e7 contexts, lag 13): size-independent properties -- rows sum to one, the intercept2
    gradient sums to zero (softmax), shard additivity of the backward pass, forward independent of sharding."""
    dev = torch.device("cuda", 0)
    lag, fw, n = 13, 8, 10_000_000
    g = torch.Generator(device=dev).manual_seed(2)
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=g)
    codes[:1000, :3] = 4                                   # some start contexts
    packed = kernels.pack_kmers(codes)
    _, params = _make(lag, fw, dev, 9)
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    assert (prior.sum(1) - 1.0).abs().max().item() < 1e-14 and prior.min().item() > 0.0
    cut = 3_333_333
    pa, _ = kernels.cnn_forward(packed[:cut].contiguous(), flat, lag, fw)
    assert torch.equal(pa, prior[:cut])
    grad = torch.randn(n, 5, dtype=torch.float64, device=dev, generator=g)
    full = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, grad)
    parts = [kernels.cnn_backward(packed[a:b].contiguous(), flat, lag, fw, t1[a:b].contiguous(), prior[a:b].contiguous(),
                                  grad[a:b].contiguous()) for a, b in ((0, cut), (cut, n))]
    scale = full.abs().max().item()
    assert (full - (parts[0] + parts[1])).abs().max().item() <= 1e-11 * scale
    off = 8 * 5 * 30 + 6 * 30 + 6 * 30 * 16 + 16 + 16 * 5          # intercept2 sits after filters, intercept0, weights1, intercept1, weights2
    assert abs(full[off:off + 5].sum().item()) <= 1e-9 * full[off:off + 5].abs().sum().item()


def test_cnn_saturating_parameters():
    """Large weights: logits far apart (probabilities down to 1e-60), first-layer pre-activations far below zero (elu at
    its floor): rows still positive, normalised and equal to the oracle's; gradients against torch autograd."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    lag, fw, n = 13, 8, 600
    kmers = _random_kmers(n, lag, rng)
    ar_func, params = _make(lag, fw, dev, 6)
    with torch.no_grad():
        params[4].mul_(60.0)          # weights2
        params[1].sub_(6.0)           # intercept0: most first-layer units saturate at -1
        params[6].mul_(5.0)           # scale0
    codes = torch.from_numpy(core.encode_kmers(kmers, "dna")).to(dev)
    packed = kernels.pack_kmers(codes)
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    prior, t1 = kernels.cnn_forward(packed, flat, lag, fw)
    want = o.ar_func_cnn(o.one_hot(kmers, "dna"), [p.detach().cpu().numpy() for p in params])
    got = prior.cpu().numpy()
    assert got.min() > 0.0 and want.min() < 1e-20
    assert np.allclose(np.log(got), np.log(want), rtol=0, atol=1e-10) and np.allclose(got.sum(1), 1.0, rtol=1e-14)
    grad_rows = torch.from_numpy(rng.standard_normal((n, 5))).to(dev)
    g = kernels.cnn_backward(packed, flat, lag, fw, t1, prior, grad_rows)
    rows = ar_func(core.tf_one_hot(kmers, "dna", device=dev))
    rows.backward(grad_rows)
    k = 0
    for p in params:
        w = p.grad.reshape(-1)
        assert (g[k:k + w.numel()] - w).abs().max().item() <= 1e-9 * max(w.abs().max().item(), 1e-30), p.shape
        k += w.numel()

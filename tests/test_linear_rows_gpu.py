"""GPU parity tests of the linear AR function as rows (bear_linear_forward_f64 / bear_linear_backward_f64, through the C ABI):
forward against the oracle's restatement of ar_funcs.py:43-45, backward against the oracle chain (softmax backward + the
einsum's transpose in NumPy) and against torch fp64 autograd of the torch formulation.  Tolerances: rows 1e-13 relative;
d/d mat 1e-11 of the gradient's largest entry (LDS fp64 atomics reorder the sums)."""
import numpy as np
import pytest
import torch

import bear_oracle as o
from bear_amd import ar_funcs, bear_ref, core, kernels

pytestmark = pytest.mark.gpu

ROW_RTOL = 1e-13
GRAD_RTOL = 1e-11


def _codes(n, lag, rng, sorted_blocks=False):
    if sorted_blocks:       # as a k-mer-sorted batch holds them: long shared prefixes, the last letters vary fastest
        key = np.sort(rng.integers(0, 4 ** min(lag, 9), size=n))
        codes = np.zeros((n, lag), dtype=np.int8)
        codes[:, : lag - min(lag, 9)] = rng.integers(0, 4, size=lag - min(lag, 9))[None, :]
        for j in range(min(lag, 9)):
            codes[:, lag - 1 - j] = (key >> (2 * j)) & 3
    else:
        codes = rng.integers(0, 4, size=(n, lag)).astype(np.int8)
    codes[rng.random((n, lag)) < 0.03] = 4        # start symbol
    codes[rng.random((n, lag)) < 0.02] = -1       # unknown letter: all-zero one-hot row (core.py:173)
    return codes


def _onehot(codes):
    n, lag = codes.shape
    oh = np.zeros((n, lag, 5))
    for l in range(lag):
        ok = codes[:, l] >= 0
        oh[np.nonzero(ok)[0], l, codes[ok, l]] = 1.0
    return oh


@pytest.mark.parametrize("lag,n", [(13, 10_000), (5, 1365), (21, 4097), (1, 70), (2, 64), (3, 65), (4, 1), (12, 127), (20, 1023)])
@pytest.mark.parametrize("sorted_blocks", [False, True])
def test_linear_rows_match_oracle(lag, n, sorted_blocks):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(lag * 1000 + n)
    codes = _codes(n, lag, rng, sorted_blocks)
    mat = rng.normal(size=(lag, 5, 5)) * 0.4
    q = rng.normal(size=(n, 5)) * np.exp(rng.normal(size=(n, 1)))
    q[rng.random(n) < 0.3] = 0.0                      # contexts without counts: zero gradient rows
    oh = _onehot(codes)
    want = o.ar_func_linear(oh, mat)
    packed = kernels.pack_kmers(torch.from_numpy(codes).to(dev))
    d_mat = torch.from_numpy(mat).to(dev)
    prior = kernels.linear_forward(packed, d_mat, lag)
    assert np.allclose(prior.cpu().numpy(), want, rtol=ROW_RTOL, atol=1e-300)
    gz = want * (q - (want * q).sum(-1, keepdims=True))
    want_g = np.einsum("njk,nl->jkl", oh, gz)
    g = kernels.linear_backward(packed, lag, prior, torch.from_numpy(q).to(dev)).cpu().numpy()
    assert g.shape == (lag, 5, 5)
    assert np.allclose(g, want_g, rtol=0, atol=GRAD_RTOL * max(np.abs(want_g).max(), 1e-300)), np.abs(g - want_g).max()


def test_linear_rows_saturated_logits():
    """Logits of tens (tables of exponentials), hundreds (tables of logits) and thousands (max-shifted softmax)."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    n, lag = 3001, 5
    codes = _codes(n, lag, rng)
    packed = kernels.pack_kmers(torch.from_numpy(codes).to(dev))
    oh = _onehot(codes)
    for scale in (8.0, 25.0, 150.0, 400.0, 3000.0):
        mat = rng.normal(size=(lag, 5, 5)) * scale
        mat[0, 0] *= 0.001
        want = o.ar_func_linear(oh, mat)
        got = kernels.linear_forward(packed, torch.from_numpy(mat).to(dev), lag).cpu().numpy()
        assert np.all(np.isfinite(got))
        assert np.allclose(got, want, rtol=1e-11, atol=1e-300), scale


def test_linear_ar_func_autograd_matches_torch_formulation():
    """ar_func(codes) on the device runs the two kernels behind autograd; values and d/d mat equal the torch formulation
    (one-hot einsum + softmax, ar_funcs.py:41-45) on the same contexts, also inside bear_ref's mixing (bear_ref.py:63-68)."""
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    n, lag = 5000, 7
    codes = torch.from_numpy(_codes(n, lag, rng)).to(dev)
    onehot = torch.from_numpy(_onehot(codes.cpu().numpy())).to(dev)
    g = torch.Generator(device=dev).manual_seed(3)
    f, (mat,) = ar_funcs.make_ar_func_linear(lag, 4, device=dev, generator=g)
    assert f.fused
    w = torch.randn(n, 5, dtype=torch.float64, device=dev, generator=g)
    y = f(codes)
    y.backward(w)
    got_y, got_g = y.detach().clone(), mat.grad.clone()
    mat.grad = None
    y2 = f(onehot)
    y2.backward(w)
    assert torch.allclose(got_y, y2.detach(), rtol=1e-13, atol=0)
    assert float((got_g - mat.grad).abs().max()) <= GRAD_RTOL * float(mat.grad.abs().max())
    # leading dimensions and no_grad
    with torch.no_grad():
        y3 = f(codes.reshape(50, 100, lag))
    assert y3.shape == (50, 100, 5) and torch.equal(y3.reshape(n, 5), got_y)
    # bear_ref: (nw net(kmers) + jukes_cantor(ref, tau)) / (nw + 1), gradients of tau_signed, net_weight_signed and mat
    ref = torch.from_numpy(rng.poisson(0.3, size=(n, 5)).astype(np.float64)).to(dev) + 1e-7
    ref[:, -1] = 0
    af, params = bear_ref._make_ref_ar_func(lag, 4, ar_funcs.make_ar_func_linear, {}, device=dev)
    af(codes, ref).backward(w)
    got = [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    af(onehot, ref).backward(w)
    for a, p in zip(got, params):
        assert float((a - p.grad).abs().max()) <= 1e-10 * float(p.grad.abs().max()), (a, p.grad)


def test_linear_rows_full_size_properties():
    """1e7 contexts, lag 13 (BASELINE configs[2] at size): rows sum to one; d/d mat in k-mer order == in random order == the
    sum of two halves; every d/d mat[l][a][:] sums to zero over the output letter; repeated launches agree to rounding; a
    sampled chunk equals the oracle."""
    dev = torch.device("cuda", 0)
    N, lag = 10_000_019, 13
    gen = torch.Generator(dev).manual_seed(77)
    codes = torch.randint(0, 4, (N, lag), dtype=torch.int8, device=dev, generator=gen)
    codes[torch.rand(N, lag, device=dev, generator=gen) < 0.01] = 4
    codes[torch.rand(N, lag, device=dev, generator=gen) < 0.005] = -1
    mat = (0.3 * torch.randn(lag, 5, 5, dtype=torch.float64, device=dev, generator=gen)).contiguous()
    q = torch.randn(N, 5, dtype=torch.float64, device=dev, generator=gen)
    q[torch.rand(N, device=dev, generator=gen) < 0.3] = 0
    packed = kernels.pack_kmers(codes)
    prior = kernels.linear_forward(packed, mat, lag)
    assert float((prior.sum(1) - 1).abs().max()) < 1e-14 and float(prior.min()) > 0
    g_r = kernels.linear_backward(packed, lag, prior, q)
    scale = float(g_r.abs().max())
    assert float(g_r.sum(-1).abs().max()) <= 1e-9 * scale
    order = kernels.kmer_order(packed, lag).long()
    ps, fs, qs = packed[order].contiguous(), prior[order].contiguous(), q[order].contiguous()
    assert torch.equal(kernels.linear_forward(ps, mat, lag), fs)
    runs = [kernels.linear_backward(ps, lag, fs, qs) for _ in range(4)]
    assert float((runs[0] - g_r).abs().max()) <= 1e-10 * scale
    for r in runs[1:]:
        assert float((r - runs[0]).abs().max()) <= 1e-11 * scale
    cut = 5_000_007
    g_ab = kernels.linear_backward(ps[:cut].clone(), lag, fs[:cut].clone(), qs[:cut].clone()) + \
        kernels.linear_backward(ps[cut:].clone(), lag, fs[cut:].clone(), qs[cut:].clone())
    assert float((g_ab - runs[0]).abs().max()) <= 1e-10 * scale
    lo, hi = 3_000_000, 3_004_000
    cd = codes[lo:hi].cpu().numpy()
    want = o.ar_func_linear(_onehot(cd), mat.cpu().numpy())
    assert np.allclose(prior[lo:hi].cpu().numpy(), want, rtol=ROW_RTOL, atol=0)


def test_linear_rows_argument_errors():
    dev = torch.device("cuda", 0)
    packed = kernels.pack_kmers(torch.zeros((10, 5), dtype=torch.int8, device=dev))
    mat = torch.zeros((5, 5, 5), dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):
        kernels.linear_forward(packed, mat[:4], 5)
    with pytest.raises(ValueError):
        kernels.linear_backward(packed, 5, torch.zeros((9, 5), dtype=torch.float64, device=dev), torch.zeros((10, 5), dtype=torch.float64, device=dev))
    from bear_amd import _lib
    with pytest.raises(_lib.BearError):
        kernels.linear_forward(packed, torch.zeros((22, 5, 5), dtype=torch.float64, device=dev), 22)
    # an empty batch: no launch, zero gradient
    empty = torch.zeros((0,), dtype=torch.int64, device=dev)
    assert kernels.linear_forward(empty, mat, 5).shape == (0, 5)
    g = kernels.linear_backward(empty, 5, torch.zeros((0, 5), dtype=torch.float64, device=dev), torch.zeros((0, 5), dtype=torch.float64, device=dev))
    assert float(g.abs().max()) == 0.0

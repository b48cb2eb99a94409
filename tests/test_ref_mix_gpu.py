"""GPU parity tests of bear_ref's prior rows for a parametrised net function (bear_ref_mix_forward_f64 / bear_ref_mix_backward_f64,
through the C ABI) against the oracle's restatement of bear_ref.py:9-33, 63-68 and against torch fp64 autograd of the same formulas.
Tolerances: rows 1e-14 relative; gradient rows 1e-14 relative; the two scalar gradients 1e-12 of their own L1 mass."""
import numpy as np
import pytest
import torch

import bear_oracle as o
from bear_amd import bear_ref, kernels

pytestmark = pytest.mark.gpu


def _inputs(n, seed, dev):
    rng = np.random.default_rng(seed)
    g = rng.dirichlet(np.full(5, 0.4), size=n)
    counts = rng.poisson(0.2, size=(n, 5)).astype(np.float64)
    counts[rng.random(n) < 0.1] *= 1000.0
    ref = counts + 1e-7
    ref[:, -1] = 0.0
    q = rng.normal(size=(n, 5)) * np.exp(2 * rng.normal(size=(n, 1)))
    q[rng.random(n) < 0.3] = 0.0
    return [torch.from_numpy(a).to(dev) for a in (g, ref, q)]


def _torch_formula(g, ref, tau_s, nw_s):
    nw, tau = torch.exp(nw_s), torch.exp(tau_s)
    return (nw * g + bear_ref._counts_to_probs(ref, tau, 4)) / (nw + 1)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1365, 100_003])
@pytest.mark.parametrize("tau_s,nw_s", [(float(np.log(1 / 30)), float(-np.log(100))), (0.7, 1.3), (-6.0, -9.0)])
def test_ref_mix_matches_torch_autograd_and_oracle(n, tau_s, nw_s):
    dev = torch.device("cuda", 0)
    g, ref, q = _inputs(n, n + 17, dev)
    t = torch.tensor(tau_s, dtype=torch.float64, device=dev, requires_grad=True)
    w = torch.tensor(nw_s, dtype=torch.float64, device=dev, requires_grad=True)
    g_t = g.clone().requires_grad_(True)
    want = _torch_formula(g_t, ref, t, w)
    want.backward(q)
    got = kernels.ref_mix_forward(g, ref, t.detach(), w.detach())
    assert torch.allclose(got, want.detach(), rtol=1e-14, atol=0)
    want_o = o.ref_ar_func(g.cpu().numpy(), ref.cpu().numpy(), tau_s, nw_s)      # the oracle's restatement (NumPy)
    assert np.allclose(got.cpu().numpy(), want_o, rtol=1e-14, atol=0)
    rows, scalars = kernels.ref_mix_backward(g, ref, q, t.detach(), w.detach())
    assert torch.allclose(rows, g_t.grad, rtol=1e-14, atol=0)
    # scalar gradients against their own L1 mass (the sum of the absolute per-row terms)
    with torch.no_grad():
        nw, tau = np.exp(nw_s), np.exp(tau_s)
        d = ref / ref.abs().sum(-1, keepdim=True) - torch.tensor([0.25, 0.25, 0.25, 0.25, 0.0], dtype=torch.float64, device=dev)
        jc_t = torch.from_numpy(o.counts_to_probs(ref.cpu().numpy(), tau)).to(dev)
        mass_w = float((q * (g - jc_t)).sum(-1).abs().sum()) * nw / (nw + 1) ** 2
        mass_t = float((q * d).sum(-1).abs().sum()) * tau * np.exp(-tau) / (nw + 1)
    assert abs(float(scalars[0]) - float(t.grad)) <= 1e-12 * mass_t + 1e-300, (float(scalars[0]), float(t.grad))
    assert abs(float(scalars[1]) - float(w.grad)) <= 1e-12 * mass_w + 1e-300, (float(scalars[1]), float(w.grad))


def test_ref_ar_func_takes_the_fused_path_and_matches_the_torch_one():
    """bear_ref's ar_func on device tensors runs the kernels behind autograd; on the host the torch formulas: same values and
    the same gradients for tau_signed, net_weight_signed and the net function's parameters."""
    from bear_amd import ar_funcs
    dev = torch.device("cuda", 0)
    n, lag = 4001, 5
    rng = np.random.default_rng(3)
    codes = torch.from_numpy(rng.integers(0, 4, size=(n, lag)).astype(np.int8))
    _, ref, q = _inputs(n, 5, torch.device("cpu"))
    af_d, p_d = bear_ref._make_ref_ar_func(lag, 4, ar_funcs.make_ar_func_linear, {}, device=dev)
    af_h, p_h = bear_ref._make_ref_ar_func(lag, 4, ar_funcs.make_ar_func_linear, {})
    with torch.no_grad():
        p_d[0].fill_(0.3)
        p_d[1].fill_(-0.8)
        for a, b in zip(p_h, p_d):
            a.copy_(b.cpu())
    y_d = af_d(codes.to(dev), ref.to(dev))
    y_h = af_h(codes, ref)
    assert torch.allclose(y_d.cpu(), y_h.detach(), rtol=1e-13, atol=0)
    y_d.backward(q.to(dev))
    y_h.backward(q)
    for a, b in zip(p_d, p_h):
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 1e-11 * float(b.grad.abs().max())
    with torch.no_grad():
        assert torch.equal(af_d(codes.to(dev), ref.to(dev)), y_d.detach())
        # rows that start at an odd row of a larger tensor (8-byte aligned only) are copied, not refused
        c1, r1 = codes.to(dev)[1:], ref.to(dev)[1:]
        assert r1.data_ptr() % 16 == 8
        assert torch.equal(af_d(c1, r1), y_d.detach()[1:])


def test_ref_mix_argument_errors_and_empty_batch():
    dev = torch.device("cuda", 0)
    g, ref, q = _inputs(10, 1, dev)
    t = torch.tensor(0.1, dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):
        kernels.ref_mix_forward(g, ref[:9], t, t)
    with pytest.raises(ValueError):
        kernels.ref_mix_forward(g, ref, t.cpu(), t)
    with pytest.raises(ValueError):
        kernels.ref_mix_backward(g, ref, q.float(), t, t)
    e = torch.zeros((0, 5), dtype=torch.float64, device=dev)
    assert kernels.ref_mix_forward(e, e, t, t).shape == (0, 5)
    rows, scalars = kernels.ref_mix_backward(e, e, e, t, t)
    assert rows.shape == (0, 5) and float(scalars.abs().max()) == 0.0

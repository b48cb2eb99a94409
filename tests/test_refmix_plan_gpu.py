"""GPU parity tests of bear_dm_refmix_plan_grad_f64 (bear_ref's step for a net function with parameters, the reference mixing inside
the DM step; through the C ABI) against
  * the C oracle's mode R (bear_ref.py:207-259 with the stop net function): with g = the stop row the four sums must be its four,
  * the oracle chain for arbitrary normalised net rows: mixing (bear_ref.py:63-68, NumPy) -> C oracle's sum LL, d/dh and gradient
    rows -> the mixing's chain rule in NumPy,
  * the three unfused launches (bear_ref_mix_forward, gradient rows, bear_ref_mix_backward) and bear_ref.train run both ways.
Tolerances: sum LL 1e-11; scalar gradients 2e-13 of their own L1 mass; gradient rows 1e-9 of their largest entry."""
import numpy as np
import pytest
import torch

import bear_oracle as o
import c_oracle as co
from test_parity_gpu import CASES_REF, ELBO_RTOL, GRAD_RTOL, PARAMS, _close, _mass_close, _to_dev
from conftest import YSD1

pytestmark = pytest.mark.gpu


def _case(case, ysd1):
    if case == "ysd1":
        return ysd1[1][:, 0].astype(np.uint32), ysd1[1][:, 2].astype(np.uint32)
    return CASES_REF[case]()


def _scalars(dev, *vals):
    return [torch.tensor(float(v), dtype=torch.float64, device=dev) for v in vals]


@pytest.mark.parametrize("case", list(CASES_REF))
@pytest.mark.parametrize("train_ar", [False, True])
def test_refmix_plan_with_the_stop_row_is_mode_r(case, train_ar, ysd1):
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    tr, rf = _case(case, ysd1)
    n = len(tr)
    plan = kernels.Plan(_to_dev(tr, dev), 5)
    g = torch.zeros((n, 5), dtype=torch.float64, device=dev)
    g[:, 4] = 1.0
    ref_in = torch.from_numpy(o.ref_input(rf)).to(dev)
    for args in PARAMS:
        want = co.dm_ref(tr, rf, *args, train_ar=train_ar, nthreads=4)
        out, rows = kernels.dm_refmix_planned_dev(plan, g, ref_in, *_scalars(dev, *args), train_ar=train_ar)
        got = out.cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        mass = co.dm_ref_mass(tr, rf, *args, train_ar=train_ar, nthreads=4)
        for k in range(1, 4):
            _mass_close(got[k], want[k], mass[k - 1], (case, args, k))
        assert np.all(np.isfinite(rows.cpu().numpy()))
        if train_ar:
            assert got[1] == 0.0 and want[1] == 0.0      # no h in the multinomial


@pytest.mark.parametrize("case", list(CASES_REF))
@pytest.mark.parametrize("train_ar", [False, True])
def test_refmix_plan_matches_the_oracle_chain_and_the_unfused_launches(case, train_ar, ysd1):
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    tr, rf = _case(case, ysd1)
    n = len(tr)
    rng = np.random.default_rng(n + 3)
    d_tr = _to_dev(tr, dev)
    plan = kernels.Plan(d_tr, 5)
    g = rng.dirichlet(np.full(5, 0.5), size=n)
    ref_in = o.ref_input(rf)
    d_g, d_ref = torch.from_numpy(g).to(dev), torch.from_numpy(ref_in).to(dev)
    for h_s, tau_s, nu_s in PARAMS + [(0.3, 0.5, 2.0)]:
        nw, tau = np.exp(nu_s), np.exp(tau_s)
        V, E = 1.0 / (nw + 1.0), np.exp(-tau)
        f = o.ref_ar_func(g, ref_in, tau_s, nu_s)
        want, G = co.dm_prior(tr, f, h_s, train_ar=train_ar, want_grad=True, nthreads=4)
        mass_h = 0.0 if train_ar else co.dm_prior_mass(tr, f, h_s, nthreads=4)
        d = ref_in / np.abs(ref_in).sum(-1, keepdims=True) - np.r_[np.full(4, 0.25), 0.0]
        jc = o.counts_to_probs(ref_in, tau)
        t_terms, w_terms = (G * d).sum(-1), (G * (g - jc)).sum(-1)
        want_tau, mass_tau = -tau * E * V * t_terms.sum(), tau * E * V * np.abs(G * d).sum()
        want_nw, mass_nw = nw * V * V * w_terms.sum(), nw * V * V * np.abs(G * (g - jc)).sum()
        hp, tp, wp = _scalars(dev, h_s, tau_s, nu_s)
        out, rows = kernels.dm_refmix_planned_dev(plan, d_g, d_ref, hp, tp, wp, train_ar=train_ar)
        got, rows = out.cpu().numpy(), rows.cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        _mass_close(got[1], want[1], mass_h, (case, "h"))
        # the context term cancels in both parameter gradients analytically; in the oracle chain it cancels to rounding of terms
        # of its own size, which is what the mass (sum of |G d|, |G (g - jc)| over ALL cells) measures
        _mass_close(got[2], want_tau, mass_tau, (case, "tau"))
        _mass_close(got[3], want_nw, mass_nw, (case, "nw"))
        want_rows = G * (nw * V)
        assert np.allclose(rows, want_rows, rtol=GRAD_RTOL, atol=GRAD_RTOL * np.abs(want_rows).max()), (case, np.abs(rows - want_rows).max())
        # the three launches it replaces
        f_d = kernels.ref_mix_forward(d_g, d_ref, tp, wp)
        out2, q = kernels.dm_prior_planned_dev(plan, f_d, hp.reshape(1), want_grad=True, normalized=True, train_ar=train_ar)
        rows2, sc = kernels.ref_mix_backward(d_g, d_ref, q, tp, wp)
        out2, sc = out2.cpu().numpy(), sc.cpu().numpy()
        _close(got[0], out2[0], ELBO_RTOL)
        _mass_close(got[1], out2[1], mass_h, (case, "h, unfused"))
        _mass_close(got[2], sc[0], mass_tau, (case, "tau, unfused"))
        _mass_close(got[3], sc[1], mass_nw, (case, "nw, unfused"))
        assert np.allclose(rows, rows2.cpu().numpy(), rtol=GRAD_RTOL, atol=GRAD_RTOL * np.abs(want_rows).max())


def test_refmix_plan_rows_without_counts_and_ragged_tiles():
    """Contexts without training counts get exact zero rows even when their reference row is degenerate (all zero: the mixing is
    NaN there, and nobody reads it); a table whose last tile holds an odd number of rows."""
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    n = 5001
    tr = (rng.random((n, 5)) < 0.3).astype(np.uint32) * rng.integers(1, 9, size=(n, 5)).astype(np.uint32)
    tr[::7] = 0
    ref_in = o.ref_input(rng.poisson(0.3, size=(n, 5)))
    ref_in[::7] = 0.0                                   # degenerate rows, only where there are no counts
    g = rng.dirichlet(np.full(5, 0.5), size=n)
    plan = kernels.Plan(_to_dev(tr, dev), 5)
    out, rows = kernels.dm_refmix_planned_dev(plan, torch.from_numpy(g).to(dev), torch.from_numpy(ref_in).to(dev),
                                              *_scalars(dev, 0.1, -0.7, 0.4))
    rows, out = rows.cpu().numpy(), out.cpu().numpy()
    assert np.all(np.isfinite(out)) and np.all(rows[::7] == 0.0) and np.all(np.isfinite(rows))
    keep = np.ones(n, bool)
    keep[::7] = False
    f = o.ref_ar_func(g[keep], ref_in[keep], -0.7, 0.4)
    want, G = co.dm_prior(tr[keep], f, 0.1, want_grad=True, nthreads=4)
    _close(out[0], want[0], ELBO_RTOL)
    nw = np.exp(0.4)
    assert np.allclose(rows[keep], G * nw / (nw + 1), rtol=GRAD_RTOL, atol=GRAD_RTOL * np.abs(G).max())


@pytest.mark.parametrize("net,train_ar,acc_steps", [("linear", False, 1), ("cnn", False, 1), ("linear", True, 1), ("linear", False, 2)])
def test_bear_ref_train_fused_mixing_equals_the_three_launches(net, train_ar, acc_steps, monkeypatch):
    """bear_ref.train with a net function that has parameters: the loop with the mixing inside the DM kernel against the loop
    that mixes, takes gradient rows and goes back through the mixing in three launches (BEAR_AMD_UNFUSED_MIX=1)."""
    from bear_amd import _train, ar_funcs, bear_ref, dataloader
    data = dataloader.dataloader(YSD1, "dna", 700, 3)
    make = ar_funcs.make_ar_func_linear if net == "linear" else ar_funcs.make_ar_func_cnn
    kw = {} if net == "linear" else {"num_filters": 30, "filter_width": 3, "kmer_layer1_width": 16}
    runs = []
    for unfused in (False, True):
        if unfused:
            monkeypatch.setenv("BEAR_AMD_UNFUSED_MIX", "1")
        torch.manual_seed(5)
        losses = []
        params, _, _ = bear_ref.train(data.repeat(3), 1365, 3, 0, 2, "dna", 5, make, kw, 0.01, "Adam", train_ar, acc_steps=acc_steps,
                                      loss_save=losses)
        runs.append((losses, [p.detach().cpu().numpy().copy() for p in params]))
    monkeypatch.delenv("BEAR_AMD_UNFUSED_MIX")
    assert np.allclose(runs[0][0], runs[1][0], rtol=1e-10)
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.allclose(a, b, rtol=1e-7, atol=1e-9)


def test_refmix_plan_full_size_properties():
    """1e7 contexts: the fused step == the three launches; its sums are the sums of two halves (each with its own plan); repeated
    launches agree; a sampled chunk equals the oracle chain."""
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    N = 10_000_019
    t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
    g = kernels.synth_prior(3, 0, N, dev)
    ref_in = t["ref"].to(torch.float64) + 1e-7
    ref_in[:, -1] = 0
    args = (-0.3, float(np.log(1 / 30)) + 0.2, -1.5)
    hp, tp, wp = _scalars(dev, *args)
    plan = kernels.Plan(t["train"], 5)
    runs = [kernels.dm_refmix_planned_dev(plan, g, ref_in, hp, tp, wp) for _ in range(4)]
    out, rows = runs[0][0].cpu().numpy(), runs[0][1]
    for o2, r2 in runs[1:]:
        assert np.allclose(o2.cpu().numpy(), out, rtol=1e-13, atol=0) and torch.equal(r2, rows)
    f = kernels.ref_mix_forward(g, ref_in, tp, wp)
    out2, q = kernels.dm_prior_planned_dev(plan, f, hp.reshape(1), want_grad=True, normalized=True)
    rows2, sc = kernels.ref_mix_backward(g, ref_in, q, tp, wp)
    want = np.r_[out2.cpu().numpy(), sc.cpu().numpy()]
    assert np.allclose(out, want, rtol=1e-11), (out, want)
    assert float((rows - rows2).abs().max()) <= 1e-12 * float(rows2.abs().max())
    cut = 5_000_004
    parts = np.zeros(4)
    for lo, hi in ((0, cut), (cut, N)):
        tr = t["train"][lo:hi].clone()
        o_p, r_p = kernels.dm_refmix_planned_dev(kernels.Plan(tr, 5), g[lo:hi].clone(), ref_in[lo:hi].clone(), hp, tp, wp)
        parts += o_p.cpu().numpy()
        assert float((r_p - rows[lo:hi]).abs().max()) <= 1e-13 * float(rows.abs().max())
    assert np.allclose(parts, out, rtol=1e-11)
    lo, hi = 7_000_001, 7_050_001
    tr = t["train"][lo:hi].cpu().numpy().view(np.uint32)
    fc = o.ref_ar_func(g[lo:hi].cpu().numpy(), ref_in[lo:hi].cpu().numpy(), args[1], args[2])
    _, G = co.dm_prior(tr, fc, args[0], want_grad=True, nthreads=4)
    nw = np.exp(args[2])
    assert np.allclose(rows[lo:hi].cpu().numpy(), G * nw / (nw + 1), rtol=GRAD_RTOL, atol=GRAD_RTOL * np.abs(G).max())

"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same seeded
inputs.  Tolerance: north_star asks 1e-9 relative on the fp64 ELBO; we hold 1e-11 on the
ELBO, 2e-13 of a scalar gradient's own L1 mass (MASS_RTOL below) and 1e-9 of the largest entry on gradient rows."""
import os
import numpy as np
import pytest
import torch

import bear_oracle as o
import c_oracle as co
from util import dense_table, edge_table, prior_rows, sparse_table

pytestmark = pytest.mark.gpu

ELBO_RTOL = 1e-11
GRAD_RTOL = 1e-9       # gradient ROWS (per context), relative to the largest entry
# The scalar gradients (d/dh, d/dtau, d/dnet_weight) are sums of terms of both signs: their error is bounded by a multiple of
# THEIR OWN L1 mass -- the sum of the absolute values of those terms, from the C oracle (c_oracle.dm_*_mass) -- and of nothing
# else (round 2 borrowed 1e-3 |ELBO| as a stand-in for that mass).  Observed: every case passes at 1e-13, some fail at 3e-14.
MASS_RTOL = float(os.environ.get("BEAR_TEST_MASS_RTOL", "2e-13"))


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _to_dev(a, dev):
    import torch
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32).copy()).to(dev)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _close(got, want, rtol, scale=None):
    scale = abs(want) if scale is None else scale
    assert abs(got - want) <= rtol * max(scale, 1e-300), (got, want, abs(got - want) / max(scale, 1e-300))


def _mass_close(got, want, mass, what=None):
    # + 1e-14 absolute: a table whose terms all vanish exactly (no counts) leaves ~1e-17 per context (the table log of exactly 1.0
    # is ~1e-16, not 0)
    assert abs(got - want) <= MASS_RTOL * mass + 1e-14, (what, got, want, abs(got - want) / max(mass, 1e-300))


def _ref_grads_close(got, want, tr, rf, args, train_ar=False, what=None):
    mass = co.dm_ref_mass(tr, rf, *args, train_ar=train_ar, nthreads=4)
    for k in range(1, 4):
        _mass_close(got[k], want[k], mass[k - 1], (what, k))


def _grad_scale_ref(tr, rf, args):
    # L1 mass of the per-row gradient terms (sum |.| before cancellation)
    r = o.bear_ref_step(tr, rf, *args)
    return np.abs(r["d_net"]).sum() + abs(r["ll"]) * 1e-6 + 1.0


CASES_REF = {
    "ysd1": None,
    "sparse": lambda: sparse_table(20011, 11)[::2],
    "sparse_hot": lambda: sparse_table(4099, 12, lam_scale=6.0)[::2],
    "dense": lambda: dense_table(3001, 13),
    "edge": lambda: (edge_table(), edge_table(1) // 9),
    "one_row": lambda: (np.array([[3, 0, 1, 0, 0]], np.uint32), np.array([[1, 0, 0, 0, 0]], np.uint32)),
}
PARAMS = [(0.0, np.log(1 / 30), -np.log(100)), (-4.2, -1.0, 0.3), (2.5, 1.2, -7.0)]


@pytest.mark.parametrize("case", list(CASES_REF))
@pytest.mark.parametrize("train_ar", [False, True])
def test_dm_ref_parity(case, train_ar, dev, ysd1):
    from bear_amd import kernels
    if case == "ysd1":
        tr, rf = ysd1[1][:, 0].astype(np.uint32), ysd1[1][:, 2].astype(np.uint32)
    else:
        tr, rf = CASES_REF[case]()
    d_tr, d_rf = _to_dev(tr, dev), _to_dev(rf, dev)
    for args in PARAMS:
        want = co.dm_ref(tr, rf, *args, train_ar=train_ar, nthreads=4)
        got = kernels.dm_ref(d_tr, d_rf, *args, train_ar=train_ar).cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        _ref_grads_close(got, want, tr, rf, args, train_ar, (case, args))


def test_dm_ref_known_answers(dev, ysd1):
    """SURVEY.md 8c / BASELINE.md section 2 known answers on the bundled table."""
    from bear_amd import kernels
    tr, rf = ysd1[1][:, 0].astype(np.uint32), ysd1[1][:, 2].astype(np.uint32)
    got = kernels.dm_ref(_to_dev(tr, dev), _to_dev(rf, dev), 0.0, np.log(1 / 30), -np.log(100)).cpu().numpy()
    _close(got[0], -152711537.8567275, 1e-12)
    for g, w in zip(got[1:], [-4080.3988585483107, 147.56530806373428, 902.7364555205095]):
        _close(g, w, 1e-9)
    got = kernels.dm_ref(_to_dev(tr, dev), _to_dev(rf, dev), 0.0, np.log(1 / 30), -np.log(100), train_ar=True).cpu().numpy()
    _close(got[0], -155088323.57920885, 1e-12)


@pytest.mark.parametrize("case", ["ysd1", "sparse", "sparse_hot", "dense", "edge", "one_row"])
@pytest.mark.parametrize("train_ar", [False, True])
def test_dm_prior_parity(case, train_ar, dev, ysd1):
    from bear_amd import kernels
    if case == "ysd1":
        tr = ysd1[1][:, 0].astype(np.uint32)
    else:
        tr = CASES_REF[case]()[0]
    n = len(tr)
    for seed, h_s, conc in [(1, 0.0, 1.0), (2, -3.0, 0.2), (3, 1.7, 5.0)]:
        f = prior_rows(n, seed, conc)
        want, wg = co.dm_prior(tr, f, h_s, train_ar=train_ar, want_grad=True, nthreads=4)
        out, g = kernels.dm_prior(_to_dev(tr, dev), _to_dev(f, dev), h_s, train_ar=train_ar, want_grad=True)
        out, g = out.cpu().numpy(), g.cpu().numpy()
        _close(out[0], want[0], ELBO_RTOL)
        if train_ar:
            assert out[1] == 0.0 and want[1] == 0.0       # no h in the multinomial
        else:
            _mass_close(out[1], want[1], co.dm_prior_mass(tr, f, h_s, nthreads=4), (case, seed))
        assert np.allclose(g, wg, rtol=1e-9, atol=1e-9 * np.abs(wg).max())
        out2, g2 = kernels.dm_prior(_to_dev(tr, dev), _to_dev(f, dev), h_s, train_ar=train_ar, want_grad=False)
        assert g2 is None
        # want_grad=False takes the sorted-work-item kernel in BEAR mode: an independent reduction
        assert np.allclose(out2.cpu().numpy(), out, rtol=1e-12, atol=1e-12 * abs(out[0]))


def test_test_core_construction_on_gpu(dev):
    """bear_model/tests/test_core.py:7-26 re-stated: random Poisson counts [3,5,5] with a
    broadcast concentration [5,5]; DM counts_log_prob == SciPy closed form."""
    from scipy.special import loggamma
    from bear_amd import kernels
    rng = np.random.default_rng(21)
    trans = rng.poisson(size=(3, 5, 5)).astype(np.uint32)
    conc = rng.exponential(size=(5, 5))
    want = (np.sum(loggamma(conc + trans) - loggamma(conc), axis=-1)
            - (loggamma(conc.sum(-1) + trans.sum(-1)) - loggamma(conc.sum(-1)))).sum()
    c = trans.reshape(-1, 5)
    f = np.broadcast_to(conc, trans.shape).reshape(-1, 5).copy()
    out, _ = kernels.dm_prior(_to_dev(c, dev), _to_dev(f, dev), 0.0, eps=0.0)
    _close(out.cpu().numpy()[0], want, 1e-12)


@pytest.mark.parametrize("n", [0, 1, 3, 255, 1023, 1024, 1025, 4097])
def test_ragged_sizes(n, dev):
    from bear_amd import kernels
    import torch
    tr, _, rf = sparse_table(max(n, 1), 31)
    tr, rf = tr[:n], rf[:n]
    f = prior_rows(max(n, 1), 5)[:n]
    args = (0.1, -3.0, -4.0)
    want = co.dm_ref(tr, rf, *args) if n else np.zeros(4)
    d_tr = _to_dev(tr, dev) if n else torch.empty((0, 5), dtype=torch.int32, device=dev)
    d_rf = _to_dev(rf, dev) if n else torch.empty((0, 5), dtype=torch.int32, device=dev)
    got = kernels.dm_ref(d_tr, d_rf, *args).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-10, atol=1e-12)
    if n:
        want2, _ = co.dm_prior(tr, f, 0.1)
        got2, _ = kernels.dm_prior(d_tr, _to_dev(f, dev), 0.1)
        assert np.allclose(got2.cpu().numpy(), want2, rtol=1e-10, atol=1e-12)


def test_additivity_and_permutation_full_size(dev):
    """Size-independent properties at a bench-scale table (synthetic generator on device):
    the sum over the table equals the sum over its shards (what row-sharding across GPUs
    relies on), and sampled chunks agree with the oracle."""
    import torch
    from bear_amd import kernels
    N = 20_000_000
    t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
    args = (0.0, np.log(1 / 30), -np.log(100))
    full = kernels.dm_ref(t["train"], t["ref"], *args).cpu().numpy()
    cuts = [0, 1, 1023, 5_000_001, 13_333_337, N]
    parts = sum(kernels.dm_ref(t["train"][a:b].contiguous(), t["ref"][a:b].contiguous(), *args).cpu().numpy()
                for a, b in zip(cuts[:-1], cuts[1:]))
    assert np.allclose(full, parts, rtol=1e-12)
    # shard generation is position-independent: rows [a, b) generated alone equal the slice
    a, b = 7_000_003, 7_050_003
    s = kernels.synth_counts(20211012, a, b - a, dev, want=("train", "ref"))
    assert torch.equal(s["train"], t["train"][a:b]) and torch.equal(s["ref"], t["ref"][a:b])
    # sampled-chunk parity against the oracle
    tr = s["train"].cpu().numpy().view(np.uint32)
    rf = s["ref"].cpu().numpy().view(np.uint32)
    want = co.dm_ref(tr, rf, *args, nthreads=4)
    got = kernels.dm_ref(s["train"], s["ref"], *args).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-10)
    f = kernels.synth_prior(20211012, a, b - a, dev)
    want2, _ = co.dm_prior(tr, f.cpu().numpy(), -0.5, nthreads=4)
    got2, _ = kernels.dm_prior(s["train"], f, -0.5)
    assert np.allclose(got2.cpu().numpy(), want2, rtol=1e-10)
    assert abs(f.sum(1) - 1).max().item() < 1e-12


def test_item_paths_against_mpmath_grid(dev):
    """Every branch of the lgamma/digamma-difference evaluation, item by item, against
    SciPy (loggamma / digamma differences are computed where they are well conditioned,
    and by the exact rising-factorial sums for small c): product path for each c = 1..31,
    the c = 31/32 switch to the Stirling path, the x < 8 shift, tiny and huge x."""
    import torch
    from scipy.special import gammaln, digamma
    from bear_amd import kernels
    xs = np.array([1e-7, 3e-5, 0.013, 0.1 + 1e-7, 0.5, 0.999, 1.0, 1.7, 4.0, 7.99, 8.0, 8.01, 31.4, 250.0,
                   1e4, 1e7, 2.0 ** 30, 2.0 ** 31, 1e12])
    cs = np.array(list(range(0, 40)) + [63, 64, 65, 100, 1000, 254715, 10 ** 7, 4_000_000_000], dtype=np.uint64)
    X, C = np.meshgrid(xs, cs, indexing="ij")
    X, C = X.ravel(), C.ravel()
    # reference values: exact sums for c <= 64, SciPy differences above
    Dw, Pw = np.zeros_like(X), np.zeros_like(X)
    for i, (x, c) in enumerate(zip(X, C)):
        c = int(c)
        if c <= 64:
            j = np.arange(c, dtype=np.float64)
            Dw[i] = np.sum(np.log(x + j))
            Pw[i] = np.sum(1.0 / (x + j))
        else:
            import mpmath as mp
            mp.mp.dps = 40
            Dw[i] = float(mp.loggamma(mp.mpf(x) + c) - mp.loggamma(mp.mpf(x)))
            Pw[i] = float(mp.digamma(mp.mpf(x) + c) - mp.digamma(mp.mpf(x)))
    dx = torch.from_numpy(X).to(dev)
    dc = torch.from_numpy(C.astype(np.uint32).view(np.int32)).to(dev)
    for path in (0, 1, 2):
        D, P = kernels.dm_items(dx, dc, path=path)
        D, P = D.cpu().numpy(), P.cpu().numpy()
        # absolute tolerance scaled by the L1 mass of the sum (sum |log(x+j)|), relative 1e-13
        massD = np.array([np.sum(np.abs(np.log(x + np.arange(min(int(c), 64))))) + abs(d) + abs(gammaln(x)) * (c > 0)
                          for x, c, d in zip(X, C, Dw)])
        assert np.all(np.abs(D - Dw) <= 4e-13 * massD + 1e-15)  # 1e-15: the table log of exactly 1.0 is ~1e-16, not 0, (path, np.max(np.abs(D - Dw) / (massD + 1e-300)))
        assert np.all(np.abs(P - Pw) <= 1e-13 * np.abs(Pw) + 1e-300), (path, np.max(np.abs(P - Pw) / (np.abs(Pw) + 1e-300)))
    # out-of-domain concentration -> NaN, never a silent number
    bad = torch.tensor([0.0, -1.0, float("nan")], dtype=torch.float64, device=dev)
    D, P = kernels.dm_items(bad, torch.tensor([3, 3, 3], dtype=torch.int32, device=dev))
    assert torch.isnan(D).all() and torch.isnan(P).all()


def test_sorted_kernel_matches_rows_kernel(dev):
    """The sorted-work-item kernel and the row-per-thread kernel (kept for AR mode and the
    gradient-row variant) are two independent reductions of the same table."""
    from bear_amd import kernels
    tr, _, rf = sparse_table(100_003, 77)
    f = prior_rows(len(tr), 9)
    d_tr, d_f = _to_dev(tr, dev), _to_dev(f, dev)
    a, _ = kernels.dm_prior(d_tr, d_f, -0.2)                  # sorted
    b, g = kernels.dm_prior(d_tr, d_f, -0.2, want_grad=True)  # rows
    assert np.allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-12)
    # d/dh_signed is also recoverable from the gradient rows: sum g * (-prior)
    dh = -(g * d_f).sum().item()
    assert abs(dh - a.cpu().numpy()[1]) <= 1e-10 * abs(dh)


@pytest.mark.parametrize("case", ["ysd1", "sparse", "sparse_hot", "dense", "edge", "one_row"])
def test_planned_kernels_parity(case, dev, ysd1):
    """The planned (sort-at-load-time) kernels against the oracle, and against the unplanned
    kernels, on every case incl. ragged tiles, Stirling-path items and uint32-range counts."""
    from bear_amd import kernels
    if case == "ysd1":
        tr, rf = ysd1[1][:, 0].astype(np.uint32), ysd1[1][:, 2].astype(np.uint32)
    else:
        tr, rf = CASES_REF[case]()
    d_tr, d_rf = _to_dev(tr, dev), _to_dev(rf, dev)
    plan_r = kernels.Plan(d_tr, 4)
    plan_n = kernels.Plan(d_tr, 5)
    assert plan_r.nbytes > 0 and plan_n.nbytes >= plan_r.nbytes
    for args in PARAMS:
        want = co.dm_ref(tr, rf, *args, nthreads=4)
        got = kernels.dm_ref_planned(plan_r, d_rf, *args).cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        _ref_grads_close(got, want, tr, rf, args, False, (case, args))
        _ref_grads_close(kernels.dm_ref(d_tr, d_rf, *args).cpu().numpy(), got, tr, rf, args, False, (case, args, "unplanned"))
    n = len(tr)
    for seed, h_s, conc in [(1, 0.0, 1.0), (2, -3.0, 0.2), (3, 1.7, 5.0)]:
        f = prior_rows(n, seed, conc)
        if seed == 3:
            f = f * np.linspace(0.5, 3.0, n)[:, None]  # rows that do not sum to one: every context owns its A
        want, _ = co.dm_prior(tr, f, h_s, nthreads=4)
        mass_h = co.dm_prior_mass(tr, f, h_s, nthreads=4)
        got = kernels.dm_prior_planned(plan_n, _to_dev(f, dev), h_s).cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        _mass_close(got[1], want[1], mass_h, (case, seed))
        wantg = co.dm_prior(tr, f, h_s, want_grad=True, nthreads=4)[1]
        for norm in ([False, True] if seed != 3 else [False]):   # planned kernel that also writes the gradient rows
            got, g = kernels.dm_prior_planned(plan_n, _to_dev(f, dev), h_s, normalized=norm, want_grad=True)
            got, g = got.cpu().numpy(), g.cpu().numpy()
            _close(got[0], want[0], ELBO_RTOL)
            _mass_close(got[1], want[1], mass_h, (case, seed, norm))
            assert np.allclose(g, wantg, rtol=1e-9, atol=1e-9 * np.abs(wantg).max()), (case, seed, np.abs(g - wantg).max())
        if seed != 3:  # rows sum to one: the caller may assert it (context terms from the plan's histogram)
            got = kernels.dm_prior_planned(plan_n, _to_dev(f, dev), h_s, normalized=True).cpu().numpy()
            _close(got[0], want[0], ELBO_RTOL)
            _mass_close(got[1], want[1], mass_h, (case, seed, "normalized"))


@pytest.mark.parametrize("case", ["ysd1", "sparse", "sparse_hot", "dense", "edge", "one_row"])
def test_dense_form_of_the_plan(case, dev, ysd1):
    """bear_plan_create_auto: the dense tables (the reference's own ysd1 table, SURVEY 8d's dense stress table) get the plan's dense
    form -- nothing kept per item, the mode-N entry points stream the count and prior rows -- and the sparse ones the sorted
    encoding; either way the oracle's sums, scalar gradients and gradient rows at the planned kernels' tolerances, in BEAR and
    multinomial mode, parameters by value and from device memory; the fused steps turn a dense-form plan away."""
    import torch
    from bear_amd import _lib, kernels
    if case == "ysd1":
        tr = ysd1[1][:, 0].astype(np.uint32)
    else:
        tr, _ = CASES_REF[case]()
    n = len(tr)
    d_tr = _to_dev(tr, dev)
    plan = kernels.Plan(d_tr, 5, rows_if_dense=True)
    assert plan.rowwise == (case in ("ysd1", "dense")), (case, plan.rowwise)
    if plan.rowwise:
        assert plan.nbytes < 64 * 1024                       # histograms only
        with pytest.raises(_lib.BearError):
            plan.pair_contexts(torch.zeros(n, dtype=torch.int64, device=dev), 5)
    for seed, h_s, conc in [(1, 0.0, 1.0), (2, -3.0, 0.2), (3, 1.7, 5.0)]:
        f = prior_rows(n, seed, conc)
        if seed == 3:
            f = f * np.linspace(0.5, 3.0, n)[:, None]
        d_f = _to_dev(f, dev)
        want, wantg = co.dm_prior(tr, f, h_s, want_grad=True, nthreads=4)
        mass_h = co.dm_prior_mass(tr, f, h_s, nthreads=4)
        for norm in ([False, True] if seed != 3 else [False]):
            got = kernels.dm_prior_planned(plan, d_f, h_s, normalized=norm).cpu().numpy()
            _close(got[0], want[0], ELBO_RTOL)
            _mass_close(got[1], want[1], mass_h, (case, seed, norm))
            got, g = kernels.dm_prior_planned(plan, d_f, h_s, normalized=norm, want_grad=True)
            _close(got.cpu().numpy()[0], want[0], ELBO_RTOL)
            _mass_close(got.cpu().numpy()[1], want[1], mass_h, (case, seed, norm, "rows"))
            g = g.cpu().numpy()
            assert np.allclose(g, wantg, rtol=1e-9, atol=1e-9 * np.abs(wantg).max()), (case, seed, np.abs(g - wantg).max())
        h_dev = torch.tensor([h_s], dtype=torch.float64, device=dev)
        got, g = kernels.dm_prior_planned_dev(plan, d_f, h_dev, want_grad=True)
        _close(got.cpu().numpy()[0], want[0], ELBO_RTOL)
        assert np.allclose(g.cpu().numpy(), wantg, rtol=1e-9, atol=1e-9 * np.abs(wantg).max()), (case, seed, "dev")
        want_ar, wantg_ar = co.dm_prior(tr, f, h_s, train_ar=True, want_grad=True, nthreads=4)
        got, g = kernels.dm_prior_planned(plan, d_f, h_s, train_ar=True, want_grad=True)
        _close(got.cpu().numpy()[0], want_ar[0], ELBO_RTOL)
        assert np.allclose(g.cpu().numpy(), wantg_ar, rtol=1e-12, atol=1e-12 * (np.abs(wantg_ar).max() + 1e-300)), (case, seed, "ar")


@pytest.mark.parametrize("case", ["ysd1", "sparse", "sparse_hot", "dense", "edge", "one_row"])
def test_planned_ar_mode_parity(case, dev, ysd1):
    """train_ar (multinomial, core.py:138-139) on the plan: sum c log(f + eps), gradients w.r.t. tau_s, nu_s
    (mode R) and the prior rows (mode N) against the oracle."""
    from bear_amd import kernels
    if case == "ysd1":
        tr, rf = ysd1[1][:, 0].astype(np.uint32), ysd1[1][:, 2].astype(np.uint32)
    else:
        tr, rf = CASES_REF[case]()
    d_tr, d_rf = _to_dev(tr, dev), _to_dev(rf, dev)
    plan_r, plan_n = kernels.Plan(d_tr, 4), kernels.Plan(d_tr, 5)
    for args in PARAMS:
        want = co.dm_ref(tr, rf, *args, train_ar=True, nthreads=4)
        got = kernels.dm_ref_planned(plan_r, d_rf, *args, train_ar=True).cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        assert got[1] == 0.0 and want[1] == 0.0          # no h in the multinomial
        _ref_grads_close(got, want, tr, rf, args, True, (case, args))
    n = len(tr)
    for seed, conc in [(1, 1.0), (2, 0.2), (3, 5.0)]:
        f = prior_rows(n, seed, conc)
        if seed == 3:
            f = f * np.linspace(0.5, 3.0, n)[:, None]
        want, wantg = co.dm_prior(tr, f, 0.3, train_ar=True, want_grad=True, nthreads=4)
        got = kernels.dm_prior_planned(plan_n, _to_dev(f, dev), 0.3, train_ar=True).cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        assert got[1] == 0.0
        got, g = kernels.dm_prior_planned(plan_n, _to_dev(f, dev), 0.3, train_ar=True, want_grad=True)
        _close(got.cpu().numpy()[0], want[0], ELBO_RTOL)
        assert np.allclose(g.cpu().numpy(), wantg, rtol=1e-12, atol=1e-12 * (np.abs(wantg).max() + 1e-300)), case


@pytest.mark.parametrize("case", ["ysd1", "sparse", "ties", "edge"])
def test_eval_kernel_parity(case, dev, ysd1):
    """bear_eval_f64 (bear_net.py:323-371 in one pass): log-likelihoods within ELBO_RTOL of the oracle, the
    correct-transition counts and total length exactly (same hashed arg-max noise), for any row sharding."""
    from bear_amd import kernels
    rng = np.random.default_rng(5)
    if case == "ysd1":
        tr, te = ysd1[1][:, 0].astype(np.uint32), ysd1[1][:, 1].astype(np.uint32)
    elif case == "sparse":
        tr, te, _ = sparse_table(20011, 3)
    elif case == "ties":    # uniform prior, no evidence: every arg-max is decided by the noise stream
        te, _, _ = sparse_table(5000, 9)
        tr = np.zeros_like(te)
    else:
        tr, _ = CASES_REF["edge"]()
        te = tr[::-1].copy()
    n = len(te)
    f = prior_rows(n, 4, 1.0) if case != "ties" else np.full((n, 5), 0.2)
    hs, van = np.array([0.05, 1.0, 37.0]), np.array([0.1, 1.0, 10.0])
    d_te, d_tr, d_f = _to_dev(te, dev), _to_dev(tr, dev), _to_dev(f, dev)
    for use_train in (True, False):
        want = o.evaluation_step(te, f, hs, van, tr if use_train else None, rng=o.HashNoise(77, 1000, n))
        got = kernels.evaluate(d_te, d_f, hs, van, d_tr if use_train else None, noise_seed=77, row_base=1000).cpu().numpy()
        H, V = 3, 3
        parts = (got[:H], got[H], got[H + 1:H + 1 + V], got[H + V + 1:2 * H + V + 1], got[2 * H + V + 1],
                 got[2 * H + V + 2:2 * H + 2 * V + 2], got[-1])
        for k in (0, 1, 2):
            assert np.allclose(parts[k], want[k], rtol=ELBO_RTOL, atol=0), (case, k, parts[k], want[k])
        for k in (3, 4, 5, 6):
            assert np.array_equal(np.asarray(parts[k]), np.asarray(want[k])), (case, k, parts[k], want[k])
        # sharding: two launches with the matching row_base add up (integers exactly, sums to rounding)
        cut = (n // 3) // 4 * 4
        a = kernels.evaluate(d_te[:cut], d_f[:cut], hs, van, d_tr[:cut] if use_train else None, noise_seed=77, row_base=1000)
        b = kernels.evaluate(d_te[cut:], d_f[cut:], hs, van, d_tr[cut:] if use_train else None, noise_seed=77,
                             row_base=1000 + cut)
        both = (a + b).cpu().numpy()
        assert np.array_equal(both[H + V + 1:], got[H + V + 1:])
        assert np.allclose(both[:H + V + 1], got[:H + V + 1], rtol=1e-12)
    # the BMM marginal entry is the vanilla model without training counts at eps = 0 (dataloader.py:111-118)
    alpha = np.array([0.1, 1.0, 10.0])
    wantb = np.array([o.dm_counts_log_prob(np.full(5, a), te.astype(np.float64)).sum() for a in alpha])
    assert np.allclose(kernels.bmm(d_te, alpha).cpu().numpy(), wantb, rtol=ELBO_RTOL)


def test_eval_kernel_degenerate(dev):
    import torch
    from bear_amd import kernels
    e = torch.zeros((0, 5), dtype=torch.int32, device=dev)
    out = kernels.evaluate(e, torch.zeros((0, 5), dtype=torch.float64, device=dev), [1.0], [1.0]).cpu().numpy()
    assert out.shape == (7,) and np.all(out == 0)
    with pytest.raises(Exception):
        kernels.evaluate(e, None, [1.0], [1.0])
        kernels.evaluate(torch.zeros((4, 5), dtype=torch.int32, device=dev), None, [1.0], [1.0])


def _linear_oracle(tr, codes, mat, h_s, train_ar):
    """sum LL, d/dh_s and d/d mat through the oracle: prior rows from ar_func_linear, gradient rows from the C
    oracle, softmax + einsum backward in NumPy (ar_funcs.py:41-45)."""
    lag = codes.shape[1]
    onehot = np.zeros((len(codes), lag, 5))
    for l in range(lag):
        ok = codes[:, l] >= 0
        onehot[np.nonzero(ok)[0], l, codes[ok, l]] = 1.0
    f = o.ar_func_linear(onehot, mat)
    out, G = co.dm_prior(tr, f, h_s, train_ar=train_ar, want_grad=True, nthreads=4)
    gz = f * (G - (f * G).sum(-1, keepdims=True))
    mass_h = 0.0 if train_ar else co.dm_prior_mass(tr, f, h_s, nthreads=4)
    return np.append(out, mass_h), np.einsum("njk,nl->jkl", onehot, gz)


@pytest.mark.parametrize("case", ["sparse", "sparse_hot", "dense", "edge", "ysd1"])
@pytest.mark.parametrize("lag", [5, 13, 21])
def test_fused_linear_head_parity(case, lag, dev, ysd1):
    """bear_dm_linear_f64: ELBO, d/dh and d/d mat of the whole linear-head step against the oracle chain
    (ar_func_linear -> DM gradient rows -> softmax/einsum backward), BEAR and multinomial mode, incl. unknown
    letters and the start symbol."""
    import torch
    from bear_amd import kernels
    if case == "ysd1":
        tr = ysd1[1][:, 0].astype(np.uint32)
    else:
        tr = CASES_REF[case]()[0]
    n = len(tr)
    rng = np.random.default_rng(lag * 7 + n)
    codes = rng.integers(0, 4, size=(n, lag)).astype(np.int8)
    codes[rng.random((n, lag)) < 0.03] = 4        # start symbol
    codes[rng.random((n, lag)) < 0.02] = -1       # unknown letter: all-zero one-hot row (core.py:173)
    mat = rng.normal(size=(lag, 5, 5)) * 0.4
    d_tr = _to_dev(tr, dev)
    plan = kernels.Plan(d_tr, 5)
    packed = kernels.linear_index(kernels.pack_kmers(torch.from_numpy(codes).to(dev)), lag)
    d_mat = torch.from_numpy(mat).to(dev)
    for h_s, ar in [(0.0, False), (-2.5, False), (1.5, False), (0.3, True)]:
        want, wantg = _linear_oracle(tr, codes, mat, h_s, ar)
        got, g = kernels.dm_linear(plan, packed, d_mat, h_s, train_ar=ar)
        got, g = got.cpu().numpy(), g.cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        _mass_close(got[1], want[1], want[2], (case, lag, h_s, ar))        # want[2]: the L1 mass of d/dh (0 in AR mode: both are 0)
        assert np.allclose(g, wantg, rtol=1e-9, atol=1e-9 * np.abs(wantg).max()), (case, lag, h_s, ar, np.abs(g - wantg).max())


def _sorted_by_kmer(codes):
    clean = np.where((codes >= 0) & (codes <= 4), codes, 5).astype(np.int64)
    key = np.zeros(len(codes), dtype=object)
    for l in range(codes.shape[1]):
        key = key * 6 + clean[:, l]
    return np.array(sorted(range(len(codes)), key=lambda i: (key[i], i)), dtype=np.int64)


@pytest.mark.parametrize("lag", [1, 2, 3, 4, 5, 8, 13, 21])
@pytest.mark.parametrize("case", ["sparse", "dense", "ysd1"])
def test_fused_linear_head_paired_contexts(case, lag, dev, ysd1, monkeypatch):
    """bear_plan_pair_contexts: the fused step over PAIRED lists (two neighbouring contexts with equal leading letters per
    thread) against the oracle chain and against the plain form of the same launch -- k-mer-sorted tables whose runs of equal
    leading letters have every length from one context up (start symbols and unknown letters among them), BEAR and multinomial
    mode, tables of exponentials and tables of logits; a pointer or lag the pairing was not made for takes the plain form."""
    import torch
    from bear_amd import kernels
    tr = ysd1[1][:, 0].astype(np.uint32) if case == "ysd1" else CASES_REF[case]()[0]
    n = len(tr)
    rng = np.random.default_rng(lag * 11 + n)
    # few distinct prefixes (runs of many contexts) for most rows, random contexts (runs of one or two) for the rest
    n_pre = max(1, n // 40)
    pre = rng.integers(0, 4, size=(n_pre, max(lag - 3, 0))).astype(np.int8)
    codes = rng.integers(0, 4, size=(n, lag)).astype(np.int8)
    half = n - n // 33
    codes[:half, :max(lag - 3, 0)] = pre[rng.integers(0, n_pre, size=half)]
    odd = rng.choice(n, size=max(2, n // 50), replace=False)       # one row in fifty holds start symbols / unknown letters somewhere
    for r in odd:
        codes[r, rng.integers(0, lag)] = 4 if rng.random() < 0.6 else -1
    order = _sorted_by_kmer(codes)
    codes, tr = codes[order], np.ascontiguousarray(tr[order])
    d_tr = _to_dev(tr, dev)
    plan = kernels.Plan(d_tr, 5)
    idx = kernels.linear_index(kernels.pack_kmers(torch.from_numpy(codes).to(dev)), lag)
    assert plan.pair_info()[0] == 0
    assert plan.pair_contexts(idx, lag) is True
    n_paired, n_plain = plan.pair_info()
    assert n_paired >= 1 and n_paired + n_plain == len(plan.tiles()[0])
    other = idx.clone()
    for scale, cases in ((0.4, [(0.0, False), (-2.5, False), (0.3, True)]), (150.0, [(0.2, False)]), (3000.0, [(0.0, True)])):
        mat = rng.normal(size=(lag, 5, 5)) * scale
        if scale > 1:
            mat[0, 0] *= 0.001                  # contexts starting with letter 0 keep small logits from that position (as above)
        d_mat = torch.from_numpy(mat).to(dev)
        for h_s, ar in cases:
            want, wantg = _linear_oracle(tr, codes, mat, h_s, ar)
            got, g = kernels.dm_linear(plan, idx, d_mat, h_s, train_ar=ar)
            plain, gp = kernels.dm_linear(plan, other, d_mat, h_s, train_ar=ar)        # another buffer: the plain form
            monkeypatch.setenv("BEAR_AMD_LINEAR_UNPAIRED", "1")
            plain2, gp2 = kernels.dm_linear(plan, idx, d_mat, h_s, train_ar=ar)
            monkeypatch.delenv("BEAR_AMD_LINEAR_UNPAIRED")
            got, g, plain, gp = got.cpu().numpy(), g.cpu().numpy(), plain.cpu().numpy(), gp.cpu().numpy()
            _close(got[0], want[0], ELBO_RTOL)
            _mass_close(got[1], want[1], want[2], (case, lag, h_s, ar))
            tol = 1e-9 if scale < 1000 else 1e-8
            # (a fully saturated softmax leaves gradients of 1e-12 and less: held to the scale of ONE count then, not to their own)
            assert np.allclose(g, wantg, rtol=tol, atol=1e-9 * max(np.abs(wantg).max(), 1.0 if scale > 1000 else 1e-300)), (
                case, lag, scale, h_s, ar, np.abs(g - wantg).max())
            # (the order of a sum varies between the forms and, with ticketed item units, from launch to launch: d/dh is held to its L1 mass)
            same = lambda a, b: np.isclose(a[0], b[0], rtol=1e-13, atol=0) and abs(a[1] - b[1]) <= 1e-13 * max(want[2], abs(b[1]))
            assert same(got, plain) and np.allclose(g, gp, rtol=0, atol=1e-11 * max(np.abs(gp).max(), 1.0 if scale > 1000 else 1e-300))
            assert same(plain2.cpu().numpy(), plain)
    # a plan holds one pairing: pairing it again (for the other buffer) replaces the first
    assert plan.pair_contexts(other, lag) is True
    a, ga = kernels.dm_linear(plan, other, d_mat, 0.1)
    b, gb = kernels.dm_linear(plan, idx, d_mat, 0.1)
    assert torch.allclose(a, b, rtol=1e-13, atol=0) and float((ga - gb).abs().max()) <= 1e-11 * float(gb.abs().max())


def test_paired_lists_are_dealt_permutations_of_the_live_contexts(dev):
    """What plan_pair_kernel may and may not do (kernels_linear.h): a tile's paired list holds every context of the tile that has
    counts exactly once; entries 2 j and 2 j + 1 share every pair group (a lane's two contexts read the same leading table rows);
    an empty entry only ever sits in the second slot; the lanes of a run of equal leading letters stay together.  Inside such a run
    the builder is free, and uses it: the 16 lanes of a pass of an LDS atomic meet on few bank pairs (triple row mod 16) -- held
    here to the level the dealing reaches on a table as dense in k-mer space as the 1e8-context benchmark (the sorted order
    alone: ~11 turns per instruction, 4 = no conflict)."""
    import ctypes
    import torch
    from bear_amd import _lib, kernels
    lag, n = 13, 400_000
    gen = torch.Generator(dev).manual_seed(3)
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=gen)
    codes[:, :4] = torch.tensor([2, 0, 3, 1], dtype=torch.int8, device=dev)          # 4^9 k-mers behind a fixed prefix: 1.5 contexts per k-mer
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for l in range(lag):
        key = key * 6 + codes[:, l].to(torch.int64)
    order = torch.argsort(key)
    codes = codes[order].contiguous()
    tr = kernels.synth_counts(11, 0, n, dev, want=("train",))["train"][order].contiguous()
    idx = kernels.linear_index(kernels.pack_kmers(codes), lag)
    plan = kernels.Plan(tr, 5)
    assert plan.pair_contexts(idx, lag) is True
    n_tiles = len(plan.tiles()[0])
    L = _lib.lib()
    L.bear_debug_pair_lists.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    stride = L.bear_debug_pair_lists(None, 0, 0, None, None)        # LIN_LIVE2_STRIDE: entries, then one level word per pair
    assert stride >= 2 * (1024 - 128) + 8
    lists, row0 = np.zeros((n_tiles, stride), dtype=np.uint16), np.zeros(n_tiles, dtype=np.uint64)
    assert L.bear_debug_pair_lists(plan._h, 0, n_tiles, lists.ctypes.data, row0.ctypes.data) == 0
    words = idx.cpu().numpy().view(np.uint64)
    live = (tr != 0).any(dim=1).cpu().numpy()
    npair = (lag - 3 + 1) // 2
    pair_mask = np.uint64((1 << (6 * npair)) - 1)
    ends = np.append(row0[1:], np.uint64(n)).astype(np.int64)
    turns, instr = 0, 0
    for t in range(n_tiles):
        m = int(lists[t, 0])
        assert m % 2 == 0 and m > 0
        e = lists[t, 2:2 + m].astype(np.int64)
        assert not (e[0::2] == 0xffff).any()
        rows = e[e != 0xffff]
        want = np.flatnonzero(live[int(row0[t]):ends[t]])
        assert len(rows) == len(want) and np.array_equal(np.sort(rows), want), t
        w = words[int(row0[t]) + np.where(e == 0xffff, 0, e)]
        blk = w & pair_mask
        second = e[1::2] != 0xffff
        assert np.array_equal(blk[1::2][second], blk[0::2][second]), t
        lane_blk = blk[0::2]                                        # runs of equal leading letters stay together
        starts = np.flatnonzero(np.append(True, lane_blk[1:] != lane_blk[:-1]))
        assert len(np.unique(lane_blk[starts])) == len(starts), t
        # the level words behind the entries: leading pair groups a pair's unit of 64 / row of 16 / quad of 4 share (the list's last
        # pair fills the last unit) -- what phase C's adds are gated by, restated from the k-mers
        n_pairs = m // 2
        n_lev = (n_pairs + 63) // 64 * 64
        lev = lists[t, 2 + m:2 + m + n_lev].astype(np.int64).reshape(-1, 64)
        cv = np.concatenate([lane_blk, np.repeat(lane_blk[-1:], n_lev - n_pairs)]).reshape(-1, 64)
        step = cv ^ np.roll(cv, 1, axis=1)

        def shared(width):
            s = step.reshape(-1, 64 // width, width).copy()
            s[:, :, 0] = 0 if width < 64 else s[:, :, 0]            # a group's first lane steps in from the group before
            o = np.bitwise_or.reduce(s, axis=2)
            low = np.array([[(int(x) & -int(x)).bit_length() - 1 for x in r] for r in o])
            return np.repeat(np.where(o == 0, npair, low // 6), width, axis=1)
        assert np.array_equal(lev & 15, shared(64)), t
        assert np.array_equal((lev >> 4) & 15, shared(16)), t
        assert np.array_equal((lev >> 8) & 15, shared(4)), t
        cl = np.where(e == 0xffff, -1, ((w >> np.uint64(6 * npair)) & np.uint64(255)).astype(np.int64) % 16)
        twin = (e[1::2] != 0xffff) & (w[1::2] == w[0::2])           # copies of one k-mer in a lane: one add
        cl[1::2] = np.where(twin, -1, cl[1::2])
        cl = np.concatenate([cl, -np.ones((-m) % 128, dtype=np.int64)]).reshape(-1, 64, 2)
        for slot in (0, 1):
            q = cl[:, :, slot].reshape(-1, 4, 16)
            mx = np.zeros(q.shape[:2], dtype=np.int64)
            for v in range(16):
                mx = np.maximum(mx, (q == v).sum(-1))
            turns += int(mx.sum())
            instr += int((q >= 0).any(-1).any(-1).sum())
    assert turns / instr < 7.0, turns / instr


def test_pairing_declines_a_sparse_table(dev):
    """A table of random 13-mers has runs of one context: a paired list would be twice the plain one and no longer fit the
    kernel's row threads -- bear_plan_pair_contexts says so, leaves the plan as it was, and the step runs in its plain form."""
    import torch
    from bear_amd import kernels
    n, lag = 200_000, 13
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train",))["train"]
    gen = torch.Generator(dev).manual_seed(5)
    codes = torch.randint(0, 4, (n, lag), dtype=torch.int8, device=dev, generator=gen)
    idx = kernels.linear_index(kernels.pack_kmers(codes), lag)
    mat = (0.3 * torch.randn(lag, 5, 5, dtype=torch.float64, device=dev, generator=gen)).contiguous()
    plan = kernels.Plan(t, 5)
    before = plan.nbytes
    want, gw = kernels.dm_linear(plan, idx, mat, 0.2)
    assert plan.pair_contexts(idx, lag) is False and plan.nbytes == before
    got, g = kernels.dm_linear(plan, idx, mat, 0.2)
    assert torch.allclose(got, want, rtol=1e-12, atol=0) and float((g - gw).abs().max()) <= 1e-11 * float(gw.abs().max())


def test_fused_linear_head_saturated_logits(dev):
    """Logits of tens (the product-of-exponentials form of the group tables, partial products up to e^+-500), of hundreds
    (tables of logits, un-shifted softmax while the wave's |logit| < 600) and of thousands (max-shifted: a saturated softmax):
    parity with the oracle chain in every form, also with a mix inside one table."""
    import torch
    from bear_amd import kernels
    tr = CASES_REF["sparse"]()[0]
    n, lag = len(tr), 5
    rng = np.random.default_rng(99)
    codes = rng.integers(0, 4, size=(n, lag)).astype(np.int8)
    plan = kernels.Plan(_to_dev(tr, dev), 5)
    idx = kernels.linear_index(kernels.pack_kmers(torch.from_numpy(codes).to(dev)), lag)
    for scale in (8.0, 25.0, 150.0, 400.0, 3000.0):
        mat = rng.normal(size=(lag, 5, 5)) * scale
        mat[0, 0] *= 0.001                      # contexts starting with letter 0 keep small logits from that position
        for h_s, ar in [(0.0, False), (0.4, True)]:
            want, wantg = _linear_oracle(tr, codes, mat, h_s, ar)
            got, g = kernels.dm_linear(plan, idx, torch.from_numpy(mat).to(dev), h_s, train_ar=ar)
            got, g = got.cpu().numpy(), g.cpu().numpy()
            assert np.all(np.isfinite(got)) and np.all(np.isfinite(g))
            _close(got[0], want[0], ELBO_RTOL)
            assert np.allclose(g, wantg, rtol=1e-8, atol=1e-9 * max(np.abs(wantg).max(), 1e-300)), (scale, h_s, ar)


def test_fused_linear_head_full_size(dev):
    """BASELINE configs[2] at size (1e7 contexts, lag 13, bear_dm_linear_f64): the sums of a k-mer-sorted table equal those of
    the same rows in random order; a table's sums are the sums of its two halves (each with its own plan); every
    d/d mat[l][a][:] sums to zero over the output letter (softmax backward); a sampled chunk equals the oracle chain."""
    import torch
    from bear_amd import kernels
    N, lag = 10_000_019, 13
    t = kernels.synth_counts(20211012, 0, N, dev, want=("train",))["train"]
    gen = torch.Generator(dev).manual_seed(77)
    codes = torch.randint(0, 4, (N, lag), dtype=torch.int8, device=dev, generator=gen)
    codes[torch.rand(N, lag, device=dev, generator=gen) < 0.01] = 4      # start symbol
    codes[torch.rand(N, lag, device=dev, generator=gen) < 0.005] = -1    # unknown letter
    mat = (0.3 * torch.randn(lag, 5, 5, dtype=torch.float64, device=dev, generator=gen)).contiguous()

    def step(tr, cd, h_s=-0.4, ar=False):
        out, g = kernels.dm_linear(kernels.Plan(tr, 5), kernels.linear_index(kernels.pack_kmers(cd), lag), mat, h_s, train_ar=ar)
        return out.cpu().numpy(), g.cpu().numpy()

    out_r, g_r = step(t, codes)
    key = torch.zeros(N, dtype=torch.int64, device=dev)
    for l in range(lag):
        c = codes[:, l].to(torch.int64)
        key = key * 6 + torch.where(c >= 0, c, torch.full_like(c, 5))
    order = torch.argsort(key)
    del key
    ts, cs = t[order].contiguous(), codes[order].contiguous()
    del order
    scale = np.abs(g_r).max()
    for ar in (False, True):
        o_r, gg_r = (out_r, g_r) if not ar else step(t, codes, ar=True)
        o_s, gg_s = step(ts, cs, ar=ar)
        assert np.allclose(o_s, o_r, rtol=1e-11), (ar, o_s, o_r)
        assert np.allclose(gg_s, gg_r, rtol=0, atol=1e-10 * np.abs(gg_r).max())
        assert np.abs(gg_s.sum(-1)).max() <= 1e-9 * np.abs(gg_r).max()
    # launches of the same step agree to rounding (LDS atomics reorder sums; a race between the waves of a block -- rows of
    # the next tile written before every wave has read the previous tile's back -- would show as a lost contribution)
    plan_s, idx_s = kernels.Plan(ts, 5), kernels.linear_index(kernels.pack_kmers(cs), lag)
    runs = [kernels.dm_linear(plan_s, idx_s, mat, -0.4) for _ in range(6)]
    for out_k, g_k in runs[1:]:
        assert torch.allclose(out_k, runs[0][0], rtol=1e-13, atol=0)
        assert float((g_k - runs[0][1]).abs().max()) <= 1e-11 * scale
    # ... and the paired form of the same step (what bear_net.train runs on a sorted batch), repeated as well
    assert plan_s.pair_contexts(idx_s, lag) is True
    for _ in range(4):
        out_k, g_k = kernels.dm_linear(plan_s, idx_s, mat, -0.4)
        assert torch.allclose(out_k, runs[0][0], rtol=1e-13, atol=0)
        assert float((g_k - runs[0][1]).abs().max()) <= 1e-11 * scale
    del plan_s, idx_s, runs
    cut = 5_000_007
    o_a, g_a = step(ts[:cut].clone(), cs[:cut].clone())
    o_b, g_b = step(ts[cut:].clone(), cs[cut:].clone())
    o_s, g_s = step(ts, cs)
    assert np.allclose(o_a + o_b, o_s, rtol=1e-11)
    assert np.allclose(g_a + g_b, g_s, rtol=0, atol=1e-10 * scale)
    lo, hi = 6_000_001, 6_020_001
    tr_c, cd_c = ts[lo:hi].clone(), cs[lo:hi].clone()
    got, g = step(tr_c, cd_c, h_s=0.2)
    want, wantg = _linear_oracle(tr_c.cpu().numpy().view(np.uint32), cd_c.cpu().numpy(), mat.cpu().numpy(), 0.2, False)
    _close(got[0], want[0], ELBO_RTOL)
    assert np.allclose(g, wantg, rtol=1e-9, atol=1e-9 * np.abs(wantg).max())


def test_planned_full_size_chunks(dev):
    """Bench-scale table: planned == unplanned on the whole table; planned on sampled chunks ==
    oracle; a plan refuses a different buffer."""
    import torch
    from bear_amd import kernels, _lib
    N = 10_000_019
    t = kernels.synth_counts(20211012, 0, N, dev, want=("train", "ref"))
    f = kernels.synth_prior(20211012, 0, N, dev)
    args = (0.3, np.log(1 / 30) + 0.1, -np.log(100))
    pr, pn = kernels.Plan(t["train"], 4), kernels.Plan(t["train"], 5)
    a = kernels.dm_ref_planned(pr, t["ref"], *args).cpu().numpy()
    b = kernels.dm_ref(t["train"], t["ref"], *args).cpu().numpy()
    assert np.allclose(a, b, rtol=1e-12)
    a = kernels.dm_prior_planned(pn, f, -0.4).cpu().numpy()
    b = kernels.dm_prior(t["train"], f, -0.4)[0].cpu().numpy()
    assert np.allclose(a, b, rtol=1e-12)
    # the light forms (rows asserted normalised, multinomial): their DMA waves wait for a tile with prefetch instructions still
    # in flight behind it (s_waitcnt vmcnt(k)) -- a tile read before it has landed would show as a launch that disagrees
    norm = [kernels.dm_prior_planned(pn, f, -0.4, normalized=True).cpu().numpy() for _ in range(20)]
    assert np.allclose(norm[0], a, rtol=1e-11)             # the synthetic prior rows are normalised
    assert all(np.allclose(x, norm[0], rtol=1e-13, atol=0) for x in norm[1:])
    ar = [kernels.dm_prior_planned(pn, f, -0.4, train_ar=True).cpu().numpy() for _ in range(20)]
    assert np.allclose(ar[0][0], kernels.dm_prior(t["train"], f, -0.4, train_ar=True)[0].cpu().numpy()[0], rtol=1e-12)
    assert all(np.allclose(x, ar[0], rtol=1e-13, atol=0) for x in ar[1:])
    lo, hi = 4_000_003, 4_100_003
    tr, rf = t["train"][lo:hi].clone(), t["ref"][lo:hi].clone()
    fc = f[lo:hi].clone()
    got = kernels.dm_prior_planned(kernels.Plan(tr, 5), fc, -0.4).cpu().numpy()
    want, _ = co.dm_prior(tr.cpu().numpy().view(np.uint32), fc.cpu().numpy(), -0.4, nthreads=4)
    assert np.allclose(got, want, rtol=1e-10)
    with pytest.raises(_lib.BearError):
        other = t["train"].clone()
        pn2 = kernels.Plan(other, 5)
        pn2.counts = t["train"]  # wrong buffer for this plan
        kernels.dm_prior_planned(pn2, f, 0.0)


@pytest.mark.parametrize("n,kind", [(1, "sparse"), (3, "dense"), (1664, "sparse"), (8192 * 2 + 5, "dense"), (8192 * 3, "sparse"), (32768 * 2 + 5, "dense"),
                                    (250_007, "mixed"), (1_000_003, "sparse")])
def test_tile_cut_on_the_device_equals_the_sequential_definition(n, kind, dev, monkeypatch):
    """bear_plan_create cuts tiles on the device (chunks walked from every possible entry point, then chained): the same tiles
    -- first row, rows, items, stream offsets -- as the sequential greedy loop over the per-group counters (BEAR_PLAN_CUT=host),
    for tables that span chunk boundaries, dense / sparse / mixed; the planned sums agree."""
    import torch
    from bear_amd import kernels
    rng = np.random.default_rng(n)
    if kind == "dense":
        tr = rng.integers(0, 60, size=(n, 5)).astype(np.uint32)
    elif kind == "sparse":
        tr = (rng.random((n, 5)) < 0.25).astype(np.uint32) * rng.integers(1, 6, size=(n, 5)).astype(np.uint32)
    else:   # stretches of empty, sparse and very dense rows: both limits (contexts, items) bind in turn
        tr = np.zeros((n, 5), dtype=np.uint32)
        seg = rng.integers(0, 3, size=n // 997 + 1).repeat(997)[:n]
        tr[seg == 1] = (rng.random(((seg == 1).sum(), 5)) < 0.3) * rng.integers(1, 30, size=((seg == 1).sum(), 5))
        tr[seg == 2] = rng.integers(1, 25, size=((seg == 2).sum(), 5))
    d = _to_dev(tr, dev)
    f = torch.from_numpy(prior_rows(n, seed=n % 1000)).to(dev)
    for ncol in (5, 4):
        plan_dev = kernels.Plan(d, ncol)
        monkeypatch.setenv("BEAR_PLAN_CUT", "host")
        plan_host = kernels.Plan(d, ncol)
        monkeypatch.delenv("BEAR_PLAN_CUT")
        for a, b in zip(plan_dev.tiles(), plan_host.tiles()):
            assert np.array_equal(a, b)
        assert plan_dev.nbytes == plan_host.nbytes
        r0, rows, _, _ = plan_dev.tiles()
        assert r0[0] == 0 and np.array_equal(r0[1:], (r0 + rows)[:-1]) and r0[-1] + rows[-1] == n      # a partition of the rows
        if ncol == 5:
            a = kernels.dm_prior_planned(plan_dev, f, -0.3).cpu().numpy()
            b = kernels.dm_prior_planned(plan_host, f, -0.3).cpu().numpy()
            assert np.allclose(a, b, rtol=1e-13)


def test_planned_randomized_shapes(dev):
    """Randomized tables (sizes around the tile / unit boundaries; sparse, dense, mostly-empty, mixed and
    single-column tables; normalised and un-normalised prior rows): planned kernels == oracle."""
    from bear_amd import kernels
    rng = np.random.default_rng(2024)
    for it in range(24):
        n = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 1663, 1664, 1665, 4096, 10007, 50021]))
        kind = it % 5
        if kind == 0:
            tr, _, rf = sparse_table(n, int(rng.integers(1e9)), lam_scale=float(rng.choice([0.05, 0.3, 1, 4, 20])))
        elif kind == 1:
            tr, rf = dense_table(n, int(rng.integers(1e9)))
        elif kind == 2:
            tr, _, rf = sparse_table(n, int(rng.integers(1e9)))
            tr[rng.random(n) < 0.5] = 0
        elif kind == 3:
            tr, _, rf = sparse_table(n, int(rng.integers(1e9)))
            d, _ = dense_table(n, 7)
            m = rng.random(n) < 0.1
            tr[m] = d[m]
        else:
            tr, rf = np.zeros((n, 5), np.uint32), np.zeros((n, 5), np.uint32)
            tr[:, int(rng.integers(5))] = rng.integers(0, 60, n)
        f = prior_rows(n, int(rng.integers(1e9)), float(rng.choice([0.2, 1, 5])))
        if it % 3 == 0:
            f = f * rng.uniform(0.5, 2.0, size=(n, 1))
        h = float(rng.uniform(-4, 3))
        args = (h, float(rng.uniform(-5, 1)), float(rng.uniform(-6, 1)))
        d_tr, d_rf, d_f = _to_dev(tr, dev), _to_dev(rf, dev), _to_dev(f, dev)
        wr = co.dm_ref(tr, rf, *args, nthreads=4)
        wn, wg = co.dm_prior(tr, f, h, want_grad=True, nthreads=4)
        gr = kernels.dm_ref_planned(kernels.Plan(d_tr, 4), d_rf, *args).cpu().numpy()
        pn = kernels.Plan(d_tr, 5)
        gn = kernels.dm_prior_planned(pn, d_f, h).cpu().numpy()
        gg, g = kernels.dm_prior_planned(pn, d_f, h, want_grad=True)
        assert abs(gr[0] - wr[0]) <= ELBO_RTOL * abs(wr[0]) + 1e-300, (it, n, kind)
        _ref_grads_close(gr, wr, tr, rf, args, False, (it, n, kind))
        mass_h = co.dm_prior_mass(tr, f, h, nthreads=4)
        for got in (gn, gg.cpu().numpy()):
            assert abs(got[0] - wn[0]) <= ELBO_RTOL * abs(wn[0]) + 1e-300, (it, n, kind)
            _mass_close(got[1], wn[1], mass_h, (it, n, kind))
        assert np.allclose(g.cpu().numpy(), wg, rtol=1e-9, atol=1e-9 * (np.abs(wg).max() + 1e-300)), (it, n, kind)


def test_eval_many_models_across_launch_chunks(dev):
    """20 h values + 5 van_reg values = 25 DM models = four launches of at most 8 models (kernels_eval.h): every slot of the
    output vector against the oracle, accuracies exactly."""
    from bear_amd import kernels
    tr, te, _ = sparse_table(7001, 11)
    n = len(te)
    f = prior_rows(n, 12, 0.8)
    hs, van = np.geomspace(1e-3, 1e2, 20), np.array([0.05, 0.3, 1.0, 4.0, 25.0])
    want = o.evaluation_step(te, f, hs, van, tr, rng=o.HashNoise(5, 0, n))
    got = kernels.evaluate(_to_dev(te, dev), _to_dev(f, dev), hs, van, _to_dev(tr, dev), noise_seed=5).cpu().numpy()
    H, V = 20, 5
    parts = (got[:H], got[H], got[H + 1:H + 1 + V], got[H + V + 1:2 * H + V + 1], got[2 * H + V + 1],
             got[2 * H + V + 2:2 * H + 2 * V + 2], got[-1])
    for k in (0, 1, 2):
        assert np.allclose(parts[k], want[k], rtol=ELBO_RTOL, atol=0), k
    for k in (3, 4, 5, 6):
        assert np.array_equal(np.asarray(parts[k]), np.asarray(want[k])), k


def _eval_parts(got, H, V):
    return (got[:H], got[H], got[H + 1:H + 1 + V], got[H + V + 1:2 * H + V + 1], got[2 * H + V + 1],
            got[2 * H + V + 2:2 * H + 2 * V + 2], got[-1])


@pytest.mark.parametrize("case", ["ysd1", "sparse", "ties", "edge", "dense", "tiny", "empty_rows"])
def test_eval_plan_kernel_parity(case, dev, ysd1):
    """bear_eval_plan_f64 (sorted plan of the test column, kernels_evalplan.h): log-likelihoods within ELBO_RTOL of the oracle,
    correct-transition counts and total length exactly, with and without training counts, for any row sharding; and equal to
    the unplanned entry."""
    from bear_amd import kernels
    if case == "ysd1":
        tr, te = ysd1[1][:, 0].astype(np.uint32), ysd1[1][:, 1].astype(np.uint32)
    elif case == "sparse":
        tr, te, _ = sparse_table(20011, 3)          # 29 tiles, the last one ragged
    elif case == "ties":    # uniform prior, no evidence: every arg-max is decided by the noise stream
        te, _, _ = sparse_table(5000, 9)
        tr = np.zeros_like(te)
    elif case == "dense":   # every cell on the Stirling path, every row heavy
        tr, _ = dense_table(3000, 2)
        te = (tr // 3).astype(np.uint32)
    elif case == "tiny":    # fewer rows than one unit
        tr, te, _ = sparse_table(37, 5, lam_scale=3.0)
    elif case == "empty_rows":   # whole tiles without a test transition
        tr, te, _ = sparse_table(4000, 6)
        te[:2900] = 0
    else:
        tr, _ = CASES_REF["edge"]()
        te = tr[::-1].copy()
    n = len(te)
    f = prior_rows(n, 4, 1.0) if case != "ties" else np.full((n, 5), 0.2)
    hs, van = np.array([0.05, 1.0, 37.0]), np.array([0.1, 1.0, 10.0])
    d_te, d_tr, d_f = _to_dev(te, dev), _to_dev(tr, dev), _to_dev(f, dev)
    H, V = 3, 3
    for use_train in (True, False):
        plan = kernels.EvalPlan(d_te, d_tr if use_train else None)
        assert plan.nbytes > 0
        want = o.evaluation_step(te, f, hs, van, tr if use_train else None, rng=o.HashNoise(77, 1000, n))
        got = kernels.evaluate_planned(plan, d_f, hs, van, noise_seed=77, row_base=1000).cpu().numpy()
        parts = _eval_parts(got, H, V)
        for k in (0, 1, 2):
            assert np.allclose(parts[k], want[k], rtol=ELBO_RTOL, atol=0), (case, use_train, k, parts[k], want[k])
        for k in (3, 4, 5, 6):
            assert np.array_equal(np.asarray(parts[k]), np.asarray(want[k])), (case, use_train, k, parts[k], want[k])
        old = kernels.evaluate(d_te, d_f, hs, van, d_tr if use_train else None, noise_seed=77, row_base=1000).cpu().numpy()
        assert np.array_equal(old[H + V + 1:], got[H + V + 1:]) and np.allclose(old[:H + V + 1], got[:H + V + 1], rtol=1e-12)
        if n >= 8:
            cut = (n // 3) // 4 * 4
            pa = kernels.EvalPlan(d_te[:cut].clone(), d_tr[:cut].clone() if use_train else None)
            pb = kernels.EvalPlan(d_te[cut:].clone(), d_tr[cut:].clone() if use_train else None)
            a = kernels.evaluate_planned(pa, d_f[:cut].clone(), hs, van, noise_seed=77, row_base=1000)
            b = kernels.evaluate_planned(pb, d_f[cut:].clone(), hs, van, noise_seed=77, row_base=1000 + cut)
            both = (a + b).cpu().numpy()
            assert np.array_equal(both[H + V + 1:], got[H + V + 1:])
            assert np.allclose(both[:H + V + 1], got[:H + V + 1], rtol=1e-12)
        # a COMPACTED batch -- only the rows with held-out counts, their table rows as row_ids (what evaluation() keeps
        # resident): the same seven sums, accuracies exactly (the noise is keyed by the table row, not by the position)
        keep = np.flatnonzero(te.any(axis=1))
        if 0 < len(keep) < n:
            ids = torch.from_numpy(keep.astype(np.int32)).to(dev)
            pc = kernels.EvalPlan(_to_dev(te[keep], dev), _to_dev(tr[keep], dev) if use_train else None)
            comp = kernels.evaluate_planned(pc, _to_dev(f[keep], dev), hs, van, noise_seed=77, row_base=1000, row_ids=ids).cpu().numpy()
            assert np.array_equal(comp[H + V + 1:], got[H + V + 1:]), (case, use_train)
            assert np.allclose(comp[:H + V + 1], got[:H + V + 1], rtol=1e-12), (case, use_train)
            with pytest.raises(ValueError):
                kernels.evaluate_planned(pc, _to_dev(f[keep], dev), hs, van, row_ids=ids[:-1])
    # no AR model, vanilla models only, no prior at all (what bmm-style callers pass)
    plan = kernels.EvalPlan(d_te, d_tr)
    got = kernels.evaluate_planned(plan, None, None, van, with_ar=False, noise_seed=3).cpu().numpy()
    want = o.evaluation_step(te, np.full((n, 5), 0.2), 1.0, van, tr, rng=o.HashNoise(3, 0, n))
    assert np.allclose(got[1:1 + V], want[2], rtol=ELBO_RTOL) and np.array_equal(got[V + 2:2 * V + 2], want[5]) and got[-1] == want[6]


def test_eval_plan_many_models_and_zero_rows(dev):
    """20 h values + 5 van_reg values = 25 DM models = seven launches of at most 4 models on one plan (an h_scan,
    bear_net.py:465-531); an empty shard is a valid plan."""
    from bear_amd import kernels
    tr, te, _ = sparse_table(7001, 11)
    n = len(te)
    f = prior_rows(n, 12, 0.8)
    hs, van = np.geomspace(1e-3, 1e2, 20), np.array([0.05, 0.3, 1.0, 4.0, 25.0])
    want = o.evaluation_step(te, f, hs, van, tr, rng=o.HashNoise(5, 0, n))
    plan = kernels.EvalPlan(_to_dev(te, dev), _to_dev(tr, dev))
    got = kernels.evaluate_planned(plan, _to_dev(f, dev), hs, van, noise_seed=5).cpu().numpy()
    parts = _eval_parts(got, 20, 5)
    for k in (0, 1, 2):
        assert np.allclose(parts[k], want[k], rtol=ELBO_RTOL, atol=0), k
    for k in (3, 4, 5, 6):
        assert np.array_equal(np.asarray(parts[k]), np.asarray(want[k])), k
    # the plan decides the vanilla arg-max on the integer counts: values for which that is not the arg-max of count + van_reg +
    # eps + noise are refused (include/bear_hip.h); the unplanned entry takes them
    for bad in (dict(eps=1e-3), dict(van_reg=[2.0 ** 31])):
        with pytest.raises(ValueError):
            kernels.evaluate_planned(plan, _to_dev(f, dev), [1.0], bad.get("van_reg", [1.0]), eps=bad.get("eps", 1e-7))
    empty = torch.zeros((0, 5), dtype=torch.int32, device=dev)
    pe = kernels.EvalPlan(empty, empty.clone())
    z = kernels.evaluate_planned(pe, torch.zeros((0, 5), dtype=torch.float64, device=dev), [1.0], [1.0]).cpu().numpy()
    assert np.all(z == 0.0)


def test_eval_plan_full_size_properties(dev):
    """2e7 synthetic contexts (the shard size of BASELINE configs[4]): shard additivity of the planned evaluation, agreement
    with the unplanned kernel (itself held to the oracle above), sampled-chunk oracle parity, accuracies exactly."""
    from bear_amd import kernels
    n = 20_000_000
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "test"))
    f = kernels.synth_prior(20211012, 0, n, dev)
    hs, van = [0.7], [0.1, 1.0, 10.0]
    plan = kernels.EvalPlan(t["test"], t["train"])
    got = kernels.evaluate_planned(plan, f, hs, van, noise_seed=9).cpu().numpy()
    old = kernels.evaluate(t["test"], f, hs, van, t["train"], noise_seed=9).cpu().numpy()
    assert np.array_equal(got[5:], old[5:]) and np.allclose(got[:5], old[:5], rtol=1e-12)
    cut = 7_000_000 // 448 * 448 + 64          # not a tile boundary of the whole table
    pa, pb = kernels.EvalPlan(t["test"][:cut], t["train"][:cut]), kernels.EvalPlan(t["test"][cut:], t["train"][cut:])
    a = kernels.evaluate_planned(pa, f[:cut], hs, van, noise_seed=9)
    b = kernels.evaluate_planned(pb, f[cut:], hs, van, noise_seed=9, row_base=cut)
    both = (a + b).cpu().numpy()
    assert np.array_equal(both[5:], got[5:]) and np.allclose(both[:5], got[:5], rtol=1e-12)
    lo, m = 12_345_678 // 4 * 4, 30_000
    ps = kernels.EvalPlan(t["test"][lo:lo + m].clone(), t["train"][lo:lo + m].clone())
    gs = kernels.evaluate_planned(ps, f[lo:lo + m].clone(), hs, van, noise_seed=9, row_base=lo).cpu().numpy()
    want = o.evaluation_step(t["test"][lo:lo + m].cpu().numpy().view(np.uint32), f[lo:lo + m].cpu().numpy(), 0.7, np.array(van),
                             t["train"][lo:lo + m].cpu().numpy().view(np.uint32), rng=o.HashNoise(9, lo, m))
    assert np.isclose(gs[0], want[0], rtol=ELBO_RTOL) and np.isclose(gs[1], want[1], rtol=ELBO_RTOL) and np.allclose(gs[2:5], want[2], rtol=ELBO_RTOL)
    assert gs[5] == want[3] and gs[6] == want[4] and np.array_equal(gs[7:10], want[5]) and gs[10] == want[6]
    # the batch as evaluation() keeps it since round 3: the contexts with held-out counts only, their table rows as row_ids --
    # the same sums at full size, accuracies exactly, also when the compacted batch is itself cut into two shards
    keep = (t["test"] != 0).any(dim=1).nonzero().squeeze(1)
    assert 0.3 * n < keep.numel() < 0.7 * n
    te_k, tr_k, f_k, ids = (t["test"][keep].contiguous(), t["train"][keep].contiguous(), f[keep].contiguous(), keep.to(torch.int32))
    comp = kernels.evaluate_planned(kernels.EvalPlan(te_k, tr_k), f_k, hs, van, noise_seed=9, row_ids=ids).cpu().numpy()
    assert np.array_equal(comp[5:], got[5:]) and np.allclose(comp[:5], got[:5], rtol=1e-12)
    half = keep.numel() // 2 // 4 * 4
    parts = None
    for a0, a1 in ((0, half), (half, keep.numel())):
        r = kernels.evaluate_planned(kernels.EvalPlan(te_k[a0:a1].clone(), tr_k[a0:a1].clone()), f_k[a0:a1].clone(), hs, van, noise_seed=9,
                                     row_ids=ids[a0:a1].clone())
        parts = r if parts is None else parts + r
    parts = parts.cpu().numpy()
    assert np.array_equal(parts[5:], got[5:]) and np.allclose(parts[:5], got[:5], rtol=1e-12)


@pytest.mark.parametrize("case", ["ysd1", "sparse", "sparse_hot", "dense", "edge", "one_row", "no_ref", "all_ref"])
@pytest.mark.parametrize("train_ar", [False, True])
def test_reference_aware_plan_parity(case, train_ar, dev, ysd1):
    """bear_plan_create_ref (kernels_refplan.h): contexts without reference counts folded into a histogram, the others streamed
    as sorted item records -- against the oracle and against the streaming plan, incl. tables where no / every context has
    reference counts, Stirling-path items and uint32-range counts."""
    from bear_amd import kernels
    if case == "ysd1":
        tr, rf = ysd1[1][:, 0].astype(np.uint32), ysd1[1][:, 2].astype(np.uint32)
    elif case == "no_ref":
        tr = sparse_table(5003, 21)[0]
        rf = np.zeros_like(tr)
    elif case == "all_ref":
        tr, _, rf = sparse_table(5003, 22, lam_scale=4.0)
        rf[:, :4] += 1
    else:
        tr, rf = CASES_REF[case]()
    d_tr, d_rf = _to_dev(tr, dev), _to_dev(rf, dev)
    plan = kernels.Plan(d_tr, 4, ref=d_rf)
    stream = kernels.Plan(d_tr, 4)
    if case in ("ysd1", "dense"):        # tables of large counts: the plan's dense form (nothing kept per item, the rows streamed)
        assert plan.nbytes < 64 * 1024 < stream.nbytes
    else:
        assert plan.nbytes >= stream.nbytes
    for args in PARAMS:
        want = co.dm_ref(tr, rf, *args, train_ar=train_ar, nthreads=4)
        got = kernels.dm_ref_planned(plan, d_rf, *args, train_ar=train_ar).cpu().numpy()
        _close(got[0], want[0], ELBO_RTOL)
        _ref_grads_close(got, want, tr, rf, args, train_ar, (case, args))
        old = kernels.dm_ref_planned(stream, d_rf, *args, train_ar=train_ar).cpu().numpy()
        _close(old[0], got[0], ELBO_RTOL)
        _ref_grads_close(old, got, tr, rf, args, train_ar, (case, args, "streaming plan"))
    other = d_rf.clone()
    with pytest.raises(Exception):      # the plan is bound to the reference buffer it was built from
        kernels.dm_ref_planned(plan, other, *PARAMS[0])


def test_reference_aware_plan_full_size(dev):
    """1e7 synthetic contexts (BASELINE configs[1]): reference-aware plan == streaming plan == unplanned kernel; shard additivity."""
    from bear_amd import kernels
    n = 10_000_000
    t = kernels.synth_counts(20211012, 0, n, dev, want=("train", "ref"))
    args = (0.0, float(np.log(1 / 30)), float(-np.log(100)))
    a = kernels.dm_ref_planned(kernels.Plan(t["train"], 4, ref=t["ref"]), t["ref"], *args).cpu().numpy()
    b = kernels.dm_ref_planned(kernels.Plan(t["train"], 4), t["ref"], *args).cpu().numpy()
    c = kernels.dm_ref(t["train"], t["ref"], *args).cpu().numpy()
    assert np.allclose(a, b, rtol=1e-12) and np.allclose(a, c, rtol=1e-12)
    cut = 3_333_332
    parts = sum(kernels.dm_ref_planned(kernels.Plan(t["train"][lo:hi], 4, ref=t["ref"][lo:hi]), t["ref"][lo:hi], *args).cpu().numpy()
                for lo, hi in ((0, cut), (cut, n)))
    assert np.allclose(parts, a, rtol=1e-12)
    m = 40_000
    want = co.dm_ref(t["train"][:m].cpu().numpy().view(np.uint32), t["ref"][:m].cpu().numpy().view(np.uint32), *args, nthreads=4)
    rf_m = t["ref"][:m].clone()
    got = kernels.dm_ref_planned(kernels.Plan(t["train"][:m].clone(), 4, ref=rf_m), rf_m, *args).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-10)


@pytest.mark.parametrize("lag", [5, 13, 14, 9])
def test_fused_linear_head_group_count_specialisations(lag, monkeypatch):
    """`launch_linear` takes a kernel with the number of letter groups as a compile-time constant for the common lags (12 / 13: 6
    groups, 14 / 15: 7, 4 / 5: 2) and the run-time form (`LIN_FOR_NG`) for the rest; BEAR_AMD_LINEAR_GENERIC=1 forces the latter.
    Same arithmetic in the same order: sum LL and d/dh equal to rounding of the draw of the work units, d/d mat to 1e-12 of its largest
    entry, plain and paired lists, both modes."""
    from bear_amd import kernels
    dev = torch.device("cuda", 0)
    n = 300_000
    t = kernels.synth_counts(11, 0, n, dev, want=("train",))["train"]
    codes = kernels.synth_kmer_codes(5, 0, min(n, 4 ** lag), lag, dev, sort=True)
    if codes.shape[0] < n:                      # (lag 5: 1024 k-mers exist -- repeat them: copies of a k-mer are neighbours)
        codes = codes.repeat_interleave(-(-n // codes.shape[0]), dim=0)[:n].contiguous()
    idx = kernels.linear_index(kernels.pack_kmers(codes), lag)
    mat = 0.2 * torch.randn(lag, 5, 5, dtype=torch.float64, device=dev, generator=torch.Generator(dev).manual_seed(2))
    plan = kernels.Plan(t, 5)
    for paired in (False, True):
        if paired:
            plan.pair_contexts(idx, lag)
        for train_ar in (False, True):
            monkeypatch.delenv("BEAR_AMD_LINEAR_GENERIC", raising=False)
            a = [x.clone() for x in kernels.dm_linear(plan, idx, mat, -0.3, train_ar=train_ar)]
            monkeypatch.setenv("BEAR_AMD_LINEAR_GENERIC", "1")
            b = [x.clone() for x in kernels.dm_linear(plan, idx, mat, -0.3, train_ar=train_ar)]
            assert torch.allclose(a[0], b[0], rtol=1e-12, atol=0), (lag, paired, train_ar)
            assert float((a[1] - b[1]).abs().max()) <= 1e-12 * float(b[1].abs().max()), (lag, paired, train_ar)
    monkeypatch.delenv("BEAR_AMD_LINEAR_GENERIC", raising=False)

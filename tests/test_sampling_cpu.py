"""CPU tests of the posterior-sampling oracle: the counter-based restatement of
log_gamma.log_gamma / get_var_probs.get_pdf against (a) the reference's own KS criterion
(bear_model/tests/test_log_gamma.py:5-19), (b) quantiles of the REFERENCE sampler run in the build container
(tests/golden/log_gamma_reference_quantiles.npz, made by tests/golden/make_log_gamma_fixture.py), and
(c) the closed forms of bear_model/tests/test_var_prob.py."""
import os

import numpy as np
from scipy import stats as st
from scipy.special import digamma

import bear_oracle as o
from conftest import GOLDEN

CONCS = np.array([0.01, 0.1, 0.5, 0.99, 1, 5, 100])   # test_log_gamma.py:10


def ks_two_sample_vs_quantiles(draws, probs, quantiles, n_ref):
    """sup |F_draws - F_ref| with F_ref known on a quantile grid, and the two-sample KS 1 % critical value."""
    F = np.searchsorted(np.sort(draws), quantiles, side="right") / draws.size
    d = np.max(np.abs(F - probs))
    crit = 1.63 * np.sqrt(1.0 / draws.size + 1.0 / n_ref) + (probs[1] - probs[0])
    return d, crit


def test_oracle_sampler_reference_ks_criterion():
    """test_log_gamma.py:12-19 applied to the restated sampler: exp(draws) ~ Gamma(conc), p > 0.1/6."""
    n, n_tile = 50000, 3
    for conc in CONCS:
        x = o.log_gamma_hash(np.full(n, conc), [n_tile], seed=0)
        assert x.shape == (n_tile, n)
        assert st.kstest(np.exp(x.reshape(-1)), cdf="gamma", args=[conc]).pvalue > 0.1 / 6


def test_oracle_sampler_vs_reference_quantiles():
    ref = np.load(os.path.join(GOLDEN, "log_gamma_reference_quantiles.npz"))
    assert np.array_equal(ref["concs"], CONCS)
    for j, conc in enumerate(CONCS):
        x = o.log_gamma_hash(np.full(100000, conc), [1], seed=7).reshape(-1)
        d, crit = ks_two_sample_vs_quantiles(x, ref["probs"], ref["quantiles"][j], int(ref["n"]))
        assert d < crit, (conc, d, crit)
        # E log G = psi(conc): the log-space moments survive where exp() underflows
        se = np.sqrt(ref["var"][j] / x.size)
        assert abs(x.mean() - digamma(conc)) < 5 * se


def test_oracle_sampler_shape_rule():
    """log_gamma.py:31,76: shape = size + concs.shape."""
    assert o.log_gamma_hash(np.ones((4, 5)), [2, 3], seed=1).shape == (2, 3, 4, 5)
    assert o.log_gamma_hash(np.ones(7), [], seed=1).shape == (7,)


def test_oracle_get_pdf_map_closed_form():
    """get_var_probs.py:174-175 against test_var_prob.py:57-58: log((seen + van) / (all + 5 van))."""
    counts = np.array([[1, 0, 0, 4, 2], [0, 0, 0, 1, 0], [0, 0, 0, 0, 0]])
    vans = np.array([0.1, 1, 10])
    lp = o.get_pdf_numpy(counts, None, None, vans, 1, True)
    assert lp.shape == (3, 5, 3, 1)
    for i, van in enumerate(vans):
        want = np.log((counts + van) / (counts.sum(1, keepdims=True) + 5 * van))
        assert np.allclose(lp[:, :, i, 0], want, rtol=1e-14)


def test_oracle_get_pdf_samples_are_log_dirichlet():
    """Normalised log-gamma draws are log Dirichlet(conc): rows sum to one and E log p_b = psi(c_b) - psi(sum c)."""
    counts = np.array([[1, 0, 0, 4, 2]])
    ar = np.array([[0.1, 0.2, 0.3, 0.35, 0.05]])
    h = np.array([0.5, 2.0])
    vans = np.array([0.1, 1.0])
    lp = o.get_pdf_numpy(counts, ar, h, vans, 20000, False, seed=3)
    assert lp.shape == (1, 5, 4, 20000)
    assert np.allclose(np.exp(lp).sum(axis=1), 1.0, rtol=1e-12)
    concs = o.get_pdf_concs(counts, ar, h, vans, False)
    want = digamma(concs) - digamma(concs.sum(-1, keepdims=True))          # [M, 1, 5]
    got = lp.mean(axis=-1)[0].T                                           # [M, 5]
    sd = np.sqrt(st.loggamma(concs[:, 0]).var() + 1e-3)
    assert np.all(np.abs(got - want[:, 0]) < 6 * sd / np.sqrt(20000) + 1e-3)

"""CPU tests: the oracle against the reference's own golden vectors / closed forms, the C
restatement against the NumPy one, and gradients against torch autograd + mpmath."""
import numpy as np
import pytest
from scipy.special import loggamma

import bear_oracle as o
import c_oracle as co
from util import edge_table, prior_rows, sparse_table


def test_dataloader_golden_first_batch(ysd1):
    # bear_model/tests/test_dataloader.py:20-32
    kmers, counts = ysd1
    assert kmers[:3] == ["TAATC", "CGGTC", "ACGCT"]
    counts_real = [[[14837, 15127, 22260, 16279, 446], [5029, 5095, 7408, 5487, 134], [16, 16, 23, 17, 0]],
                   [[61890, 729, 39733, 35956, 1017], [20524, 239, 13199, 12046, 309], [69, 0, 45, 39, 0]],
                   [[13965, 23135, 73870, 37045, 1035], [4705, 7591, 24532, 12305, 385], [14, 25, 81, 39, 0]]]
    assert np.all(counts[:3] == np.array(counts_real))
    assert counts.dtype == np.float64
    assert len(kmers) == 1365


def test_dm_closed_form_reference_test():
    # bear_model/tests/test_core.py:7-26 (same random construction, fixed seed)
    rng = np.random.default_rng(3)
    shape = np.array([3, 5])
    trans = rng.poisson(size=np.r_[shape, 5]).astype(float)
    total = trans.sum(-1)
    conc = rng.exponential(size=np.r_[shape[1], 5])
    sum_conc = conc.sum(-1)
    want = (np.sum(loggamma(conc + trans) - loggamma(conc), axis=-1)
            - (loggamma(sum_conc + total) - loggamma(sum_conc)))
    assert np.allclose(o.dm_counts_log_prob(conc, trans), want, rtol=1e-13)


def test_multinomial_closed_form_reference_test():
    # bear_model/tests/test_core.py:42-60
    rng = np.random.default_rng(4)
    trans = rng.poisson(size=(3, 5, 5)).astype(float)
    conc = rng.exponential(size=(5, 5))
    conc /= conc.sum(-1, keepdims=True)
    assert np.allclose(o.multinomial_counts_log_prob(conc, trans), np.sum(np.log(conc) * trans, axis=-1))


def test_bmm_known_answers(ysd1):
    # tests/test_dataloader.py:34-49 closed form + SURVEY.md 8c known answers
    _, counts = ysd1
    alpha = np.array([0.1, 1.0, 10.0])
    true = np.sum((np.sum(loggamma(counts[:, :, None, :] + alpha[:, None]), axis=-1)
                   - loggamma(np.sum(counts[:, :, None, :] + alpha[:, None], axis=-1)))
                  - (np.sum(loggamma(0 * counts[:, :, None, :] + alpha[:, None]), axis=-1)
                     - loggamma(np.sum(0 * counts[:, :, None, :] + alpha[:, None], axis=-1))), axis=0)
    got = o.bmm_likelihood(counts, alpha)
    assert np.allclose(true, got, rtol=1e-13)
    train_eps = o.bmm_likelihood(counts, alpha + 1e-7)[0]
    assert np.allclose(train_eps, [-152712571.34208858, -152709051.39618373, -152745386.28243095], rtol=1e-14)
    perp = np.exp(-train_eps / counts[:, 0].sum())
    assert np.allclose(perp, [3.7914698221871186, 3.7913533527306233, 3.7925557889613084], rtol=1e-12)
    # docs/usage.rst:261 "BMM 3.79"
    assert np.all(np.round(perp, 2) == 3.79)


def test_bear_ref_known_answers(ysd1):
    _, counts = ysd1
    r = o.bear_ref_step(counts[:, 0], counts[:, 2], 0.0, np.log(1 / 30), -np.log(100))
    assert np.isclose(r["ll"], -152711537.8567275, rtol=1e-14)
    assert np.allclose([r["d_h_signed"], r["d_tau_signed"], r["d_nu_signed"]],
                       [-4080.3988585483107, 147.56530806373428, 902.7364555205095], rtol=1e-11)
    ar = o.bear_ref_step(counts[:, 0], counts[:, 2], 0.0, np.log(1 / 30), -np.log(100), train_ar=True)
    assert np.isclose(ar["ll"], -155088323.57920885, rtol=1e-14)


def test_row0_mpmath():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    c = [14837, 15127, 22260, 16279, 446]
    a = mp.mpf("0.1") + mp.mpf("1e-7")
    ll = sum(mp.loggamma(a + x) - mp.loggamma(a) for x in c) - (mp.loggamma(5 * a + sum(c)) - mp.loggamma(5 * a))
    got = o.dm_counts_log_prob(np.full(5, 0.1 + 1e-7), np.array(c, float))
    assert abs(float(ll) - got) < 1e-9 * abs(got)
    assert abs(got - (-96677.60834099352)) < 1e-8


def test_gradients_vs_torch_autograd():
    torch = pytest.importorskip("torch")
    train, _, ref = sparse_table(2000, seed=5)
    h_s, tau_s, nu_s = -0.7, np.log(1 / 30) + 0.2, -np.log(100) + 0.5
    eps = 1e-7
    c = torch.tensor(train.astype(np.float64))
    r = torch.tensor(ref.astype(np.float64)) + eps
    r = r * torch.tensor([1.0, 1, 1, 1, 0], dtype=torch.float64)
    p = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (h_s, tau_s, nu_s)]
    nw, tau, h = torch.exp(p[2]), torch.exp(p[1]), torch.exp(p[0])
    shape = torch.tensor([1.0, 1, 1, 1, 0], dtype=torch.float64)
    stop = torch.tensor([0.0, 0, 0, 0, 1], dtype=torch.float64)
    norm = r / r.abs().sum(-1, keepdim=True)
    f = (nw * stop + 0.25 * shape + torch.exp(-tau) * (norm - 0.25 * shape)) / (nw + 1)
    a = f / h + eps
    A, n = a.sum(-1), c.sum(-1)
    ll = ((torch.lgamma(a + c) - torch.lgamma(a)).sum(-1) - (torch.lgamma(A + n) - torch.lgamma(A))).sum()
    ll.backward()
    got = o.bear_ref_step(train, ref, h_s, tau_s, nu_s)
    assert np.isclose(got["ll"], ll.item(), rtol=1e-13)
    want = np.array([x.grad.item() for x in p])
    have = np.array([got["d_h_signed"], got["d_tau_signed"], got["d_nu_signed"]])
    assert np.allclose(have, want, rtol=1e-10)
    # bear_net step: gradient rows and d/dh
    f2 = torch.tensor(prior_rows(2000, 7), requires_grad=True)
    hs = torch.tensor(0.3, dtype=torch.float64, requires_grad=True)
    a2 = f2 / torch.exp(hs) + eps
    ll2 = ((torch.lgamma(a2 + c) - torch.lgamma(a2)).sum(-1)
           - (torch.lgamma(a2.sum(-1) + n) - torch.lgamma(a2.sum(-1)))).sum()
    ll2.backward()
    got2 = o.bear_net_step(train, f2.detach().numpy(), 0.3)
    assert np.isclose(got2["ll"], ll2.item(), rtol=1e-13)
    assert np.isclose(got2["d_h_signed"], hs.grad.item(), rtol=1e-10)
    assert np.allclose(got2["d_prior"], f2.grad.numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("train_ar", [False, True])
def test_c_oracle_matches_numpy(train_ar, ysd1):
    _, counts = ysd1
    args = (0.25, np.log(1 / 30), -np.log(100))
    for tr, rf in [(counts[:, 0], counts[:, 2]), sparse_table(5000, 2)[::2], (edge_table(), edge_table(1) // 7)]:
        want = o.bear_ref_step(tr, rf, *args, train_ar=train_ar)
        got = co.dm_ref(tr, rf, *args, train_ar=train_ar, nthreads=2)
        assert np.isclose(got[0], want["ll"], rtol=1e-12)
        assert np.allclose(got[1:], [want["d_h_signed"], want["d_tau_signed"], want["d_nu_signed"]], rtol=1e-9, atol=1e-9)
        f = prior_rows(len(tr), 3)
        want = o.bear_net_step(tr, f, -0.4, train_ar=train_ar)
        out, g = co.dm_prior(tr, f, -0.4, train_ar=train_ar, want_grad=True, nthreads=2)
        assert np.isclose(out[0], want["ll"], rtol=1e-12)
        assert np.isclose(out[1], want["d_h_signed"], rtol=1e-9, atol=1e-9)
        assert np.allclose(g, want["d_prior"], rtol=1e-9, atol=1e-9)


def test_one_hot_and_linear():
    oh = o.one_hot(["AC[", "T[G", "NNA"], "dna")
    assert oh.shape == (3, 3, 5)
    assert oh[0, 0, 0] == 1 and oh[0, 1, 1] == 1 and oh[0, 2, 4] == 1
    assert oh[2, 0].sum() == 0 and oh[2, 2, 0] == 1
    rng = np.random.default_rng(0)
    mat = rng.standard_normal((3, 5, 5))
    f = o.ar_func_linear(oh, mat)
    assert np.allclose(f.sum(-1), 1.0)
    z = mat[0, 0] + mat[1, 1] + mat[2, 4]
    assert np.allclose(f[0], np.exp(z) / np.exp(z).sum())

"""GPU tests of the posterior-sampling path: bear_log_gamma_f64 / bear_logdir_sample_f64 through the
C ABI and the get_var_probs host mirror, against the oracle (same counter stream), the reference sampler's
quantiles, the reference's KS criterion (bear_model/tests/test_log_gamma.py) and the closed forms of
bear_model/tests/test_var_prob.py on the bundled ex_seqs table."""
import os

import numpy as np
import pytest
from scipy import stats as st
from scipy.special import digamma, logsumexp

import bear_oracle as o
from conftest import GOLDEN
from test_sampling_cpu import CONCS, ks_two_sample_vs_quantiles

pytestmark = pytest.mark.gpu

EX_SEQS = os.path.join(GOLDEN, "ex_seqs_kmap_for_var_pred.csv")
DRAW_RTOL = 1e-9     # library log / cos differ by an ulp between libm and the device: draws agree to ~1e-13


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def test_log_gamma_matches_oracle_stream(dev):
    import torch
    from bear_amd import kernels
    rng = np.random.default_rng(0)
    conc = np.concatenate([CONCS, 10.0 ** rng.uniform(-7, 4, size=3000), [1e-7, 1e-3, 1 - 1e-12, 1 + 1e-12, 1e6]])
    got = kernels.log_gamma(torch.from_numpy(conc).to(dev), 5, seed=11).cpu().numpy()
    want = o.log_gamma_hash(conc, [5], seed=11)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=DRAW_RTOL, atol=1e-12)


def test_log_gamma_reference_ks_criterion(dev):
    """bear_model/tests/test_log_gamma.py:9-19 with the HIP sampler behind the reference's log_gamma signature."""
    from bear_amd import log_gamma
    n, n_tile = 100000, 3
    concs_tile = (np.ones([len(CONCS), n]) * CONCS[:, None]).flatten()
    samples = log_gamma.log_gamma(concs_tile, size=[n_tile], seed=0).reshape([n_tile, len(CONCS), n])
    for i, conc in enumerate(CONCS):
        assert st.kstest(np.exp(samples[:, i].flatten()), cdf="gamma", args=[conc]).pvalue > 0.1 / 6


def test_log_gamma_vs_reference_quantiles(dev):
    from bear_amd import log_gamma
    ref = np.load(os.path.join(GOLDEN, "log_gamma_reference_quantiles.npz"))
    for j, conc in enumerate(ref["concs"]):
        x = log_gamma.log_gamma(np.full(400000, conc), seed=5)
        d, crit = ks_two_sample_vs_quantiles(x, ref["probs"], ref["quantiles"][j], int(ref["n"]))
        assert d < crit, (conc, d, crit)
        assert abs(x.mean() - digamma(conc)) < 5 * np.sqrt(ref["var"][j] / x.size)


def test_log_gamma_global_seed_reproducible(dev):
    from bear_amd import log_gamma
    np.random.seed(3)
    a = log_gamma.log_gamma(np.full(100, 0.3), size=[2])
    np.random.seed(3)
    b = log_gamma.log_gamma(np.full(100, 0.3), size=[2])
    assert a.shape == (2, 100) and np.array_equal(a, b)


@pytest.mark.parametrize("get_map", [False, True])
def test_logdir_sample_matches_oracle(dev, get_map):
    import torch
    from bear_amd import kernels
    rng = np.random.default_rng(2)
    K = 37
    counts = rng.poisson(3.0, size=(K, 5)).astype(np.uint32)
    counts[0] = 0
    counts[1] = [4000000000, 0, 1, 0, 0]
    ar = rng.dirichlet(np.full(5, 0.7), size=K) + 1e-7
    h = np.array([0.05, 1.0, 30.0])
    vans = np.array([0.1, 1, 10])
    mc = 1 if get_map else 9
    got = kernels.logdir_sample(torch.from_numpy(counts.view(np.int32)).to(dev), torch.from_numpy(ar).to(dev), h, vans, mc,
                                get_map=get_map, with_ar=get_map, seed=99, row_base=1000).cpu().numpy()
    want = o.get_pdf_numpy(counts, ar, h, vans, mc, get_map, seed=99, row_base=1000)
    assert got.shape == want.shape == (K, 5, 6 + int(get_map), mc)
    assert np.allclose(got, want, rtol=DRAW_RTOL, atol=1e-11)
    # vanilla models only, unseen k-mers (counts = NULL): get_var_probs.py:441-444
    got = kernels.logdir_sample(None, None, None, vans, mc, get_map=get_map, seed=5, n_rows=4, device=dev).cpu().numpy()
    want = o.get_pdf_numpy(np.zeros((4, 5)), None, None, vans, mc, get_map, seed=5)
    assert np.allclose(got, want, rtol=DRAW_RTOL, atol=1e-11)


def test_logdir_sample_sharding_invariant(dev):
    import torch
    from bear_amd import kernels
    rng = np.random.default_rng(4)
    counts = torch.from_numpy(rng.poisson(2.0, size=(64, 5)).astype(np.int32)).to(dev)
    full = kernels.logdir_sample(counts, None, None, [0.5], 7, seed=1)
    a = kernels.logdir_sample(counts[:24].contiguous(), None, None, [0.5], 7, seed=1)
    b = kernels.logdir_sample(counts[24:].contiguous(), None, None, [0.5], 7, seed=1, row_base=24)
    assert torch.equal(full, torch.cat([a, b]))


def _data():
    from bear_amd import dataloader
    return dataloader.sparse_dataloader(EX_SEQS, "dna", 500, 1)


def _e_log_beta(seen, all_, van, A=4):
    """E log Beta(seen + van, all - seen + A van) -- what test_var_prob.py:35-36 estimates with 500000 draws."""
    return digamma(seen + van) - digamma(all_ + (A + 1) * van)


def test_get_bear_probs_mc_and_map(dev):
    """bear_model/tests/test_var_prob.py:19-73 (count-table path): sequences TTTAT, TTCTT, TTTTT, TTTTT."""
    from bear_amd import get_var_probs
    wt_seq = "TTTAT"
    vars_ = np.array(["A3T", "T2C"])
    vans = np.array([0.1, 1, 10])
    scores = get_var_probs.get_bear_probs(None, wt_seq, vars_, 0, data=_data(), mc_samples=200000, vans=vans, lag=3,
                                          alphabet_name="dna", seed=0)
    assert scores.shape == (2, 3, 200000)
    true = np.empty([2, 3])
    for i, van in enumerate(vans):
        g = lambda s, a: _e_log_beta(s, a, van)
        true[0, i] = (2 * g(4, 7) + g(2, 7)) - (g(1, 7) + 2 * g(1, 1))
        true[1, i] = (g(1, 4) + g(0, 1) + 2 * g(0, 0)) - (g(3, 4) + g(1, 7) + 2 * g(1, 1))
    assert np.all(np.abs((scores.mean(-1) - true) / true) < 0.02)          # test_var_prob.py:49-51

    scores = get_var_probs.get_bear_probs(None, wt_seq, vars_, 0, data=_data(), get_map=True, vans=vans, lag=3,
                                          alphabet_name="dna")
    q = lambda s, a, van: np.log((s + van) / (a + 5 * van))
    for i, van in enumerate(vans):
        true[0, i] = (2 * q(4, 7, van) + q(2, 7, van)) - (q(1, 7, van) + 2 * q(1, 1, van))
        true[1, i] = (q(1, 4, van) + q(0, 1, van) + 2 * q(0, 0, van)) - (q(3, 4, van) + q(1, 7, van) + 2 * q(1, 1, van))
    assert np.allclose(scores, true)                                        # test_var_prob.py:73


def test_get_bear_probs_seqs_mc_marg_map(dev):
    """bear_model/tests/test_var_prob.py:76-167 (count-table path)."""
    from bear_amd import get_var_probs
    seqs = ["TTTAT", "TTCAT", "TTTTTTTTTT"]
    vans = np.array([0.1, 1, 10])
    kw = dict(vans=vans, lag=3, alphabet_name="dna")
    scores = get_var_probs.get_bear_probs_seqs(None, seqs, 0, data=_data(), mc_samples=20000, seed=1, **kw)
    margs = get_var_probs.get_bear_probs_seqs(None, seqs, 0, data=_data(), get_marg=True, **kw)
    assert scores.shape == (3, 3, 20000) and margs.shape == (3, 3)

    N = 50000
    rng = np.random.default_rng(0)
    def ld(seen, all_, van):
        return np.log(rng.beta(seen + van, all_ - seen + 4 * van, size=N))
    true = np.empty([3, 3, N])
    for i, van in enumerate(vans):
        true[0, i] = ld(4, 4, van) + ld(4, 4, van) + ld(3, 4, van) + ld(1, 7, van) + ld(1, 1, van) + ld(1, 1, van)
        true[1, i] = ld(4, 4, van) + ld(4, 4, van) + ld(1, 4, van) + ld(0, 1, van) + ld(0, 0, van) + ld(0, 0, van)
        ttt_dirs = np.log(rng.beta(4 + van, 2 + van, size=N))              # TTT -> T given not A/C/G ...
        ttt_mod = np.log(rng.beta(6 + 2 * van, 1 + 3 * van, size=N))
        true[2, i] = (ld(4, 4, van) + ld(4, 4, van) + ld(3, 4, van) + 7 * (ttt_dirs + ttt_mod)
                      + (np.log1p(-np.exp(ttt_dirs)) + ttt_mod))
    av = true.mean(-1)
    assert np.all(np.abs((scores.mean(-1) - av) / av) < 0.01)               # test_var_prob.py:127-129
    av = logsumexp(true, axis=-1) - np.log(N)
    assert np.all(np.abs((margs - av) / av) < 0.01)                         # test_var_prob.py:134-136

    scores = get_var_probs.get_bear_probs_seqs(None, seqs, 0, data=_data(), get_map=True, **kw)
    q = lambda s, a, van: np.log((s + van) / (a + 5 * van))
    want = np.empty([3, 3])
    for i, van in enumerate(vans):
        want[0, i] = 2 * q(4, 4, van) + q(3, 4, van) + q(1, 7, van) + 2 * q(1, 1, van)
        want[1, i] = 2 * q(4, 4, van) + q(1, 4, van) + q(0, 1, van) + 2 * q(0, 0, van)
        want[2, i] = 2 * q(4, 4, van) + q(3, 4, van) + 7 * q(4, 7, van) + q(2, 7, van)
    assert np.allclose(scores, want)                                         # test_var_prob.py:167


def test_get_pdf_with_ar_func_outputs(dev):
    """get_pdf with a BEAR model: 'numpy' / 'df' / 'func' views of one table agree, model order AR, BEAR, vanilla."""
    import torch
    from bear_amd import ar_funcs, get_var_probs
    g = torch.Generator(device=dev).manual_seed(1)
    ar_func, _ = ar_funcs.make_ar_func_linear(3, 4, device=dev, generator=g)
    kmers = np.array(["TTT", "TTA", "[[T", "ACG"])
    counts = np.array([[[1, 0, 0, 4, 2]], [[0, 0, 0, 1, 0]], [[0, 0, 0, 4, 0]], [[0, 0, 0, 0, 0]]])
    h, vans = np.array([0.5, 4.0]), np.array([1.0])
    arr = get_var_probs.get_pdf(kmers, counts, h, ar_func, 6, vans, 0, "dna", False, output="numpy", seed=8)
    df = get_var_probs.get_pdf(kmers, counts, h, ar_func, 6, vans, 0, "dna", False, output="df", seed=8)
    fn = get_var_probs.get_pdf(kmers, counts, h, ar_func, 6, vans, 0, "dna", False, summed=False, seed=8)
    assert arr.shape == (4, 5, 3, 6) and df.shape == (20, 18)
    assert np.array_equal(df.loc["TTAG"].to_numpy().reshape(3, 6), arr[1, 2])
    assert np.array_equal(fn(["TTT]", "ACGA"]), np.stack([arr[0, 4], arr[3, 0]]))
    with torch.no_grad():
        f = ar_func(torch.from_numpy(__import__("bear_amd").core.encode_kmers(kmers, "dna")).to(dev)).cpu().numpy()
    want = o.get_pdf_numpy(counts[:, 0], f, h, vans, 6, False, seed=8)
    assert np.allclose(arr, want, rtol=DRAW_RTOL, atol=1e-11)
    m = get_var_probs.get_pdf(kmers, counts, h, ar_func, 6, vans, 0, "dna", True, output="numpy")
    assert m.shape == (4, 5, 4, 1) and np.allclose(np.exp(m[:, :, 0, 0]), f, rtol=1e-12)


EX_SEQS_LIST = ["TTTAT", "TTCTT", "TTTTT", "TTTTT"]          # bear_model/tests/test_var_prob.py:13


def test_sequence_counter_reference_values(dev):
    """bear_model/tests/test_var_prob.py:9-21 (test_kmc_counter) with the device-built table in place of the KMC databases."""
    from bear_amd import get_var_probs
    counter = get_var_probs.make_sequence_counter(EX_SEQS_LIST, 3, reverse=False, no_end=False)
    assert np.all(counter(np.array(["TTT", "TTA", "[[T"])) == [[1, 0, 0, 4, 2], [0, 0, 0, 1, 0], [0, 0, 0, 4, 0]])
    counter = get_var_probs.make_sequence_counter(EX_SEQS_LIST, 3, reverse=True, no_end=False)
    assert np.all(counter(np.array(["TTT", "[AT", "AAA"])) == [[1, 0, 0, 4, 2], [1, 0, 0, 0, 0], [4, 0, 0, 0, 3]])
    counter = get_var_probs.make_sequence_counter(EX_SEQS_LIST, 3, reverse=False, no_end=True)
    assert np.all(counter(np.array([["TTT", "[[T"]])) == [[[1, 0, 0, 4, 0], [0, 0, 0, 0, 0]]])
    assert np.all(counter(np.array(["GGG"])) == 0)


def test_get_bear_probs_through_the_sequence_counter(dev):
    """test_var_prob.py:31-33, 53-55, 60-61: the counter path gives the same scores as scanning the table."""
    from bear_amd import get_var_probs
    vans = np.array([0.1, 1, 10])
    counter = get_var_probs.make_sequence_counter(EX_SEQS_LIST, 3, reverse=False)
    kw = dict(vans=vans, lag=3, alphabet_name="dna")
    a = get_var_probs.get_bear_probs(None, "TTTAT", np.array(["A3T", "T2C"]), 0, data=_data(), get_map=True, **kw)
    b = get_var_probs.get_bear_probs(None, "TTTAT", np.array(["A3T", "T2C"]), 0, counter=counter, get_map=True, **kw)
    assert np.allclose(a, b)
    seqs = ["TTTAT", "TTCAT", "TTTTTTTTTT"]
    a = get_var_probs.get_bear_probs_seqs(None, seqs, 0, data=_data(), get_marg=True, **kw)
    b = get_var_probs.get_bear_probs_seqs(None, seqs, 0, counter=counter, get_marg=True, **kw)
    assert np.allclose(a, b, rtol=1e-12)
    m = get_var_probs.get_bear_probs_seqs(None, seqs, 0, counter=counter, mc_samples=20000, seed=4, **kw)
    n = get_var_probs.get_bear_probs_seqs(None, seqs, 0, data=_data(), mc_samples=20000, seed=5, **kw)
    assert np.all(np.abs((m.mean(-1) - n.mean(-1)) / n.mean(-1)) < 0.01)


def test_trained_model_folder_restart_and_variant_scores(dev, tmp_path):
    """The reference's workflow across modules: train_bear_net writes config.cfg + results.pickle (models/train_bear_net.py:
    142-149), a second run restarts from them (:113-118), get_var_probs.load_bear reads the folder (get_var_probs.py:58-82)
    and get_bear_probs scores variants with the BEAR model next to the BMMs (MAP: closed form)."""
    import configparser
    from conftest import ROOT
    from bear_amd import get_var_probs
    from bear_amd.models import train_bear_net
    config = configparser.ConfigParser()
    config.read(os.path.join(ROOT, "bear_amd", "models", "config_files", "bear_lin_bear.cfg"))
    config["train"]["epochs"] = "30"
    config["train"]["batch_size"] = "1500"
    config["general"]["out_folder"] = str(tmp_path / "m1") + "*"
    config["test"]["test"] = "False"
    config["test"]["train_test"] = "False"
    assert train_bear_net.main(config) == 1
    h1 = float(config["results"]["h"])
    # restart from the saved parameters: h continues from where the first run stopped
    config2 = configparser.ConfigParser()
    config2.read(str(tmp_path / "m1" / "config.cfg"))
    config2["general"]["out_folder"] = str(tmp_path / "m2") + "*"
    config2["train"]["restart"] = "True"
    config2["train"]["restart_path"] = str(tmp_path / "m1")
    config2["train"]["epochs"] = "1"
    assert train_bear_net.main(config2) == 1
    assert abs(np.log(float(config2["results"]["h"])) - np.log(h1)) < 0.05
    # variant scores from the trained folder
    lag, alphabet, h, ar_func, data = get_var_probs.load_bear(str(tmp_path / "m1"))
    assert lag == 5 and alphabet == "dna" and np.isclose(h, h1) and data.num_rows == 1365
    wt = "ACGTACGTTAGC"
    vars_ = np.array(["G2T", "T7A"])
    vans = np.array([1.0])
    s_map = get_var_probs.get_bear_probs(str(tmp_path / "m1"), wt, vars_, 0, get_map=True, vans=vans)
    assert s_map.shape == (2, 3) and np.all(np.isfinite(s_map))            # models: AR (MAP only), BEAR h, BMM
    s_mc = get_var_probs.get_bear_probs(str(tmp_path / "m1"), wt, vars_, 0, mc_samples=4000, vans=vans, seed=2)
    assert s_mc.shape == (2, 2, 4000) and np.all(np.isfinite(s_mc))
    # with lots of data per k-mer (lag 5, 1e8 transitions) posterior samples concentrate on the MAP values
    assert np.all(np.abs(s_mc.mean(-1) - s_map[:, 1:]) < 0.02 * np.abs(s_map[:, 1:]) + 0.02)

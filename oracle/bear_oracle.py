"""CPU oracle for the BEAR empirical-Bayes training hot path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy/SciPy restatement of the arithmetic the reference executes in
TensorFlow / TensorFlow-Probability on its training + evaluation path.  It is the
*checker* for the HIP kernels in ``bear_amd/csrc``; nothing in the product package
imports it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module.

Pinning status (see DESIGN.md "Oracle"):
  * TensorFlow, tensorflow_probability and tensorflow_io are not installable in the
    build container or on the GPU box, so the reference modules (all of which import
    TF at module scope) cannot be executed here.  The arithmetic itself lives in the
    un-vendored dependency ``tensorflow_probability==0.11.1`` (reference
    ``requirements.txt:11``): ``DirichletMultinomial.log_prob``,
    ``Multinomial.log_prob`` and ``tfp.math.log_combinations``.
  * PINNED by the reference's own tests: the Dirichlet-multinomial closed form
    (``bear_model/tests/test_core.py:23-26``), the multinomial closed form
    (``test_core.py:59-60``), the parsed first batch of the bundled ysd1 table
    (``tests/test_dataloader.py:25-32``), and the BMM marginal closed form
    (``tests/test_dataloader.py:42-49``, ``tests/test_run.py:26-30``).  Those closed
    forms are written with ``scipy.special.loggamma`` in the reference tests; this
    module evaluates exactly the same expressions.
  * PARITY UNPINNED by any reference-held vector: the BEAR-mode ELBO value on real
    data and every gradient.  They are pinned here only through the closed forms
    above plus (a) torch-fp64 autograd of the same closed form and (b) mpmath at 50
    digits (tests/test_oracle.py).

Every function cites the reference lines it follows (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import numpy as np
from scipy.special import gammaln, digamma

# keras epsilon: bear_model/core.py:8, bear_net.py:4, bear_ref.py:6
EPSILON = 1e-7

# bear_model/core.py:142-153 (start symbol '[' / stop symbol ']' share the last column)
ALPHABETS = {
    "dna": "ACGT",
    "rna": "ACGU",
    "prot": "ARNDCEQGHILKMFPSTWYV",
}


# --------------------------------------------------------------------------- core
def dm_counts_log_prob(concentration, counts):
    """DM log-probability of an *ordered* sequence of transitions.

    bear_model/core.py:73-74: ``counts_dist.log_prob(value) - log_combinations(n, value)``
    with TFP's ``DirichletMultinomial.log_prob = lbeta(conc + counts) - lbeta(conc)
    + log_combinations`` -- the two multinomial coefficients cancel, leaving
    ``sum_b[lgamma(a_b + c_b) - lgamma(a_b)] - [lgamma(A + n) - lgamma(A)]``;
    pinned by bear_model/tests/test_core.py:23-26.
    """
    a = np.asarray(concentration, dtype=np.float64)
    c = np.asarray(counts, dtype=np.float64)
    a_b, c_b = np.broadcast_arrays(a, c)
    A = a_b.sum(-1)
    n = c_b.sum(-1)
    return (gammaln(a_b + c_b) - gammaln(a_b)).sum(-1) - (gammaln(A + n) - gammaln(A))


def dm_grad_concentration(concentration, counts):
    """d counts_log_prob / d concentration (what tf.GradientTape yields through
    lgamma -> digamma; bear_net.py:193, bear_ref.py:255):
    ``g_b = psi(a_b + c_b) - psi(a_b) - psi(A + n) + psi(A)``."""
    a = np.asarray(concentration, dtype=np.float64)
    c = np.asarray(counts, dtype=np.float64)
    a_b, c_b = np.broadcast_arrays(a, c)
    A = a_b.sum(-1, keepdims=True)
    n = c_b.sum(-1, keepdims=True)
    return digamma(a_b + c_b) - digamma(a_b) - digamma(A + n) + digamma(A)


def multinomial_counts_log_prob(probs, counts):
    """Ordered multinomial log-probability, bear_model/core.py:138-139
    (``sum_b c_b log p_b``; pinned by tests/test_core.py:59-60).  TFP uses
    ``multiply_no_nan`` so a zero count times log(0) contributes 0."""
    p = np.asarray(probs, dtype=np.float64)
    c = np.asarray(counts, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(c == 0, 0.0, c * np.log(p))
    return t.sum(-1)


def one_hot(kmers, alphabet="dna"):
    """bear_model/core.py:156-174 -- one-hot of equal-length strings; the start
    symbol '[' is the last column; unknown letters give an all-zero row."""
    letters = ALPHABETS[alphabet] + "["
    lut = np.full(256, -1, dtype=np.int64)
    for i, ch in enumerate(letters):
        lut[ord(ch)] = i
    arr = np.frombuffer("".join(kmers).encode(), dtype=np.uint8).reshape(len(kmers), -1)
    idx = lut[arr]
    out = np.zeros(idx.shape + (len(letters),), dtype=np.float64)
    valid = idx >= 0
    ii, jj = np.nonzero(valid)
    out[ii, jj, idx[ii, jj]] = 1.0
    return out


# --------------------------------------------------------------------------- ar_funcs
def softmax(z):
    z = z - z.max(-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(-1, keepdims=True)


def ar_func_linear(onehot, mat):
    """bear_model/ar_funcs.py:43-45: softmax(einsum('...jk,jkl->...l', kmers, mat))."""
    return softmax(np.einsum("...jk,jkl->...l", onehot, mat))


def ar_func_stop(onehot, alphabet_size=4):
    """bear_model/ar_funcs.py:121-126: constant [0,...,0,1]."""
    stop = np.zeros(alphabet_size + 1)
    stop[-1] = 1.0
    return stop


def _normalize_layer(x, axes):
    """bear_model/ar_funcs.py:5-20 (tf.nn.moments = biased variance)."""
    mean = x.mean(axis=axes, keepdims=True)
    var = x.var(axis=axes, keepdims=True)
    return (x - mean) / np.sqrt(var + 1e-5)


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0.0)))


def ar_func_cnn(onehot, params):
    """bear_model/ar_funcs.py:91-97.  ``params`` in the reference's return order
    (ar_funcs.py:98-99): filters, intercept0, weights1, intercept1, weights2,
    intercept2, scale0, scale1."""
    filters, int0, w1, int1, w2, int2, sc0, sc1 = params
    fw = filters.shape[0]
    L = onehot.shape[-2]
    # conv1d VALID, stride 1: out[..., p, f] = sum_{w,k} x[..., p+w, k] filters[w, k, f]
    windows = np.stack([onehot[..., p:p + fw, :] for p in range(L - fw + 1)], axis=-3)
    conv = np.einsum("...pwk,wkf->...pf", windows, filters)
    nn0 = sc0 * _normalize_layer(conv, (-1,)) + int0
    t1 = np.tensordot(_elu(nn0), w1, axes=[[-2, -1], [0, 1]])
    nn1 = sc1 * _normalize_layer(t1, (-1,)) + int1
    nn2 = np.tensordot(_elu(nn1), w2, axes=[[-1], [0]]) + int2
    return softmax(nn2)


# --------------------------------------------------------------------------- bear_ref prior
def ref_input(ref_counts, eps=EPSILON):
    """bear_ref.py:332-337: ``(counts[:, ds_loc_ref] + epsilon) * not_stop``."""
    r = np.asarray(ref_counts, dtype=np.float64) + eps
    r[..., -1] = 0.0
    return r


def counts_to_probs(ref_in, tau):
    """bear_ref.py:9-33 (Jukes-Cantor mutation of the L1-normalised reference row)."""
    A = ref_in.shape[-1] - 1
    norm = ref_in / np.abs(ref_in).sum(-1, keepdims=True)
    shape = np.r_[np.ones(A), 0.0]
    return (1.0 / A) * shape + np.exp(-tau) * (norm - (1.0 / A) * shape)


def ref_ar_func(net_probs, ref_in, tau_signed, nu_signed):
    """bear_ref.py:63-68: ``(nw*net(kmers) + counts_to_probs(ref, tau)) / (nw + 1)``."""
    nw = np.exp(nu_signed)
    tau = np.exp(tau_signed)
    return (nw * net_probs + counts_to_probs(ref_in, tau)) / (nw + 1.0)


# --------------------------------------------------------------------------- train steps
def bear_net_step(counts, prior, h_signed, train_ar=False, eps=EPSILON):
    """Sum log-likelihood of one batch and its gradients for ``bear_net._train_step``
    (bear_net.py:146-197) *before* the ``-(num_kmers / B)`` scaling of
    bear_net.py:190-191.

    counts: [B, A+1] transition counts; prior: [B, A+1] = ar_func(kmers).
    Returns dict(ll, d_h_signed, d_prior[B, A+1]).
      BEAR mode (bear_net.py:43-44): alpha = prior / exp(h_signed) + eps.
      AR mode   (bear_net.py:68-69): probs = prior + eps (h_signed gets no gradient,
      bear_net.py:194-196).
    """
    c = np.asarray(counts, dtype=np.float64)
    f = np.asarray(prior, dtype=np.float64)
    f = np.broadcast_to(f, c.shape)
    if train_ar:
        p = f + eps
        ll = multinomial_counts_log_prob(p, c).sum()
        return dict(ll=ll, d_h_signed=0.0, d_prior=c / p)
    h = np.exp(h_signed)
    a = f / h + eps
    ll = dm_counts_log_prob(a, c).sum()
    g = dm_grad_concentration(a, c)
    return dict(ll=ll, d_h_signed=float((g * (-f / h)).sum()), d_prior=g / h)


def bear_ref_step(counts, ref_counts, h_signed, tau_signed, nu_signed,
                  train_ar=False, net_probs=None, eps=EPSILON):
    """``bear_ref._train_step`` (bear_ref.py:207-259) for one batch, unscaled.

    ``net_probs`` defaults to the stop function (ar_funcs.py:121-126, config 2/4).
    Returns dict(ll, d_h_signed, d_tau_signed, d_nu_signed, d_net[B, A+1]).
    """
    c = np.asarray(counts, dtype=np.float64)
    r = ref_input(ref_counts, eps)
    Asz = c.shape[-1] - 1
    if net_probs is None:
        net_probs = ar_func_stop(None, Asz)
    g_net = np.broadcast_to(np.asarray(net_probs, dtype=np.float64), c.shape)
    nw = np.exp(nu_signed)
    tau = np.exp(tau_signed)
    E = np.exp(-tau)
    shape = np.r_[np.ones(Asz), 0.0]
    norm = r / r.sum(-1, keepdims=True)
    base = (1.0 / Asz) * shape + E * (norm - (1.0 / Asz) * shape)
    f = (nw * g_net + base) / (nw + 1.0)
    # partials of f w.r.t. the signed parameters
    df_dtau_s = (-tau * E) * (norm - (1.0 / Asz) * shape) / (nw + 1.0)
    df_dnu_s = nw * (g_net - f) / (nw + 1.0)
    if train_ar:
        p = f + eps
        ll = multinomial_counts_log_prob(p, c).sum()
        dLdf = c / p
        d_h = 0.0
    else:
        h = np.exp(h_signed)
        a = f / h + eps
        ll = dm_counts_log_prob(a, c).sum()
        g = dm_grad_concentration(a, c)
        dLdf = g / h
        d_h = float((g * (-f / h)).sum())
    return dict(ll=ll, d_h_signed=d_h,
                d_tau_signed=float((dLdf * df_dtau_s).sum()),
                d_nu_signed=float((dLdf * df_dnu_s).sum()),
                d_net=dLdf * nw / (nw + 1.0))


# --------------------------------------------------------------------------- BMM / evaluation
def bmm_likelihood(counts, alpha):
    """dataloader.py:111-113, 120-147: for every dataset column and every alpha,
    ``sum_i lbeta(c_i + alpha) - lbeta(alpha)``.  counts [N, num_ds, A+1] -> [num_ds, V].
    Closed form pinned by tests/test_dataloader.py:42-49."""
    c = np.asarray(counts, dtype=np.float64)[:, :, None, :]
    al = np.asarray(alpha, dtype=np.float64)[:, None]
    x = c + al
    lb1 = gammaln(x).sum(-1) - gammaln(x.sum(-1))
    y = 0 * c + al
    lb0 = gammaln(y).sum(-1) - gammaln(y.sum(-1))
    return (lb1 - lb0).sum(0)


def _ml_output(values, noise_scale, rng):
    """core.py:69-71 / 134-136: argmax with Gaussian tie-breaking noise."""
    if rng is None:
        return np.argmax(values, axis=-1)
    return np.argmax(values + noise_scale * rng.standard_normal(values.shape), axis=-1)


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _u01(hash64):
    return ((hash64 >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


EVAL_ID_ARM, EVAL_ID_VAN = 1000, 2000


def eval_noise(seed, model, rows, width=5):
    """The counter-based N(0,1) stream the evaluation entry point uses in place of tf.random.normal
    (core.py:69-71, 134-136): one Box-Muller draw per (seed, model id, global row, letter), both uniforms from
    one 64-bit hash word (32 bits each).  Model ids: j for the j-th h, EVAL_ID_ARM for the AR model,
    EVAL_ID_VAN + k for the k-th van_reg."""
    rows = np.asarray(rows, dtype=np.uint64)
    cell = rows[:, None] * np.uint64(width) + np.arange(width, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        key = _mix64(_mix64(np.uint64(seed) + np.uint64(model)) ^ cell)
    u1 = ((key >> np.uint64(32)).astype(np.float64) + 0.5) * 2.0 ** -32
    u2 = ((key & np.uint64(0xFFFFFFFF)).astype(np.float64) + 0.5) * 2.0 ** -32
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586 * u2)


class HashNoise:
    """rng stand-in for evaluation_step: hands out eval_noise streams by model id."""

    def __init__(self, seed, row_base, n_rows):
        self.seed, self.rows = seed, np.arange(row_base, row_base + n_rows, dtype=np.uint64)

    def normal(self, model):
        return eval_noise(self.seed, model, self.rows)


def evaluation_step(test_counts, prior, h, van_reg, train_counts=None, eps=EPSILON, rng=None):
    """bear_net._evaluation_step (bear_net.py:323-371) on one batch.

    prior [B, A+1] = ar_func(...) rows; ``h`` scalar or [H] (h_scan, bear_net.py:523).
    With rng=None ties are broken by first index (deterministic oracle); accuracy
    parity on tied rows is statistical in the reference.  With rng a
    ``HashNoise`` the tie-breaking noise is the counter-based stream of ``eval_noise``.
    Returns the 7 partial sums of bear_net.py:370-371.
    """
    if isinstance(rng, HashNoise):
        return _evaluation_step_hash(test_counts, prior, h, van_reg, train_counts, eps, rng)
    ct = np.asarray(test_counts, dtype=np.float64)
    f = np.broadcast_to(np.asarray(prior, dtype=np.float64), ct.shape)
    van = np.asarray(van_reg, dtype=np.float64)
    hs = np.atleast_1d(np.asarray(h, dtype=np.float64))
    if train_counts is not None:
        ctr = np.asarray(train_counts, dtype=np.float64)
        van_cond = ctr[:, None, :] + van[:, None]
        cond = ctr
    else:
        van_cond = van[:, None] * np.ones((1, ct.shape[-1]))
        van_cond = np.broadcast_to(van_cond, (ct.shape[0],) + van_cond.shape)
        cond = 0.0
    conc_ear = f[None] / hs[:, None, None] + cond + eps          # [H, B, A+1]
    ll_ear = dm_counts_log_prob(conc_ear, ct[None]).sum(-1)       # [H]
    probs = f + eps
    ll_arm = multinomial_counts_log_prob(probs, ct).sum()
    conc_van = 0.0 / 1.0 + van_cond + eps                          # [B, V, A+1]
    ll_van = dm_counts_log_prob(conc_van, ct[:, None, :]).sum(0)   # [V]
    ml_ear = _ml_output(conc_ear, 100 * eps, rng)                  # [H, B]
    ml_arm = _ml_output(probs, eps, rng)
    ml_van = _ml_output(conc_van, 100 * eps, rng)                  # [B, V]
    cor_ear = np.take_along_axis(np.broadcast_to(ct[None], conc_ear.shape),
                                 ml_ear[..., None], -1)[..., 0].sum(-1)
    cor_arm = np.take_along_axis(ct, ml_arm[:, None], -1).sum()
    cor_van = np.take_along_axis(np.broadcast_to(ct[:, None, :], conc_van.shape),
                                 ml_van[..., None], -1)[..., 0].sum(0)
    total_len = ct.sum()
    if np.ndim(h) == 0:
        ll_ear, cor_ear = ll_ear[0], cor_ear[0]
    return ll_ear, ll_arm, ll_van, cor_ear, cor_arm, cor_van, total_len


def _evaluation_step_hash(test_counts, prior, h, van_reg, train_counts, eps, noise):
    """evaluation_step with the HashNoise streams (same arithmetic, per-model noise lookup)."""
    ct = np.asarray(test_counts, dtype=np.float64)
    f = np.broadcast_to(np.asarray(prior, dtype=np.float64), ct.shape)
    hs = np.atleast_1d(np.asarray(h, dtype=np.float64))
    van = np.atleast_1d(np.asarray(van_reg, dtype=np.float64))
    ctr = np.asarray(train_counts, dtype=np.float64) if train_counts is not None else np.zeros_like(ct)

    def correct(values, scale, model):
        idx = np.argmax(values + scale * noise.normal(model), axis=-1)
        return np.take_along_axis(ct, idx[:, None], -1).sum()

    ll_ear, cor_ear, ll_van, cor_van = [], [], [], []
    for j, hv in enumerate(hs):
        conc = f / hv + ctr + eps
        ll_ear.append(dm_counts_log_prob(conc, ct).sum())
        cor_ear.append(correct(conc, 100 * eps, j))
    probs = f + eps
    ll_arm = multinomial_counts_log_prob(probs, ct).sum()
    cor_arm = correct(probs, eps, EVAL_ID_ARM)
    for k, v in enumerate(van):
        conc = ctr + v + eps
        ll_van.append(dm_counts_log_prob(conc, ct).sum())
        cor_van.append(correct(conc, 100 * eps, EVAL_ID_VAN + k))
    ll_ear, cor_ear = np.array(ll_ear), np.array(cor_ear)
    if np.ndim(h) == 0:
        ll_ear, cor_ear = ll_ear[0], cor_ear[0]
    return ll_ear, ll_arm, np.array(ll_van), cor_ear, cor_arm, np.array(cor_van), ct.sum()


# --------------------------------------------------------------------------- count-table text format
def parse_counts_tsv(path, num_ds):
    """Restates what dataloader.py:35-46 yields for a whole file: k-mer strings and a
    float64 [N, num_ds, A+1] tensor.  Row format from summarize.py:429-449:
    ``kmer \t [[g0 A,C,G,T,$],[g1 ...],...]``."""
    import json
    kmers, rows = [], []
    with open(path) as fh:
        for line in fh:
            if not line.strip():
                continue
            k, m = line.rstrip("\n").split("\t")
            kmers.append(k)
            rows.append(json.loads(m))
    arr = np.asarray(rows, dtype=np.float64)
    assert arr.shape[1] == num_ds
    return kmers, arr


# --------------------------------------------------------------------------- posterior sampling
SMP_MAX_ROUNDS = 64


def parse_sparse_rows(path, num_ds, width, header=True):
    """The sparse row format, ``kmer; [[ds, col], ...]; [value, ...]`` with a header line (bear_model/dataloader.py:52-109:
    CsvDataset(field_delim=';'), the index / value lists through decode_json, SparseTensor -> to_dense).
    -> (list of k-mer bytes, uint32 [N, num_ds, width])."""
    import json
    kmers, rows = [], []
    with open(path) as fh:
        if header:
            fh.readline()
        for line in fh:
            if not line.strip():
                continue
            k, pos, vals = [t.strip() for t in line.rstrip("\n").split(";")]
            dense = np.zeros((num_ds, width), dtype=np.uint32)
            for (d, c), v in zip(json.loads(pos), json.loads(vals)):
                dense[d, c] = v
            kmers.append(k.encode())
            rows.append(dense)
    return kmers, (np.stack(rows) if rows else np.zeros((0, num_ds, width), dtype=np.uint32))


def _gauss_key(key):
    """N(0,1) from a 64-bit counter, the same Box-Muller pair as eval_noise."""
    with np.errstate(over="ignore"):
        u1 = _u01(_mix64(key))
        u2 = _u01(_mix64(key ^ np.uint64(0x5851F42D4C957F2D)))
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586 * u2)


def log_gamma_keys(concs, keys):
    """One log Gamma(conc, 1) draw per element from its 64-bit counter key: the distribution sampled by
    log_gamma.log_gamma (log_gamma.py:17-76 -- log of a standard gamma, computed so that tiny concentrations
    do not underflow).  The reference uses vectorised rejection rounds on numpy's global generator
    (log_gamma.py:47-75); a counter-based restatement cannot share that stream, so this follows the HIP
    kernel's scheme (kernels_sample.h:smp_log_gamma): Marsaglia-Tsang for shape >= 1 and, below 1, the boost
    log G_a = log G_{a+1} + log(U)/a.  Pinned to the reference distributionally (tests/test_sampling_*.py)."""
    a = np.asarray(concs, dtype=np.float64).reshape(-1)
    keys = np.asarray(keys, dtype=np.uint64).reshape(-1)
    small = a < 1.0
    a1 = np.where(small, a + 1.0, a)
    d = a1 - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    ld = np.log(d)
    lg = ld.copy()
    live = np.ones(a.shape, dtype=bool)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        for t in range(SMP_MAX_ROUNDS):
            if not live.any():
                break
            idx = np.nonzero(live)[0]
            k = keys[idx]
            x = _gauss_key(k + np.uint64(2 * t + 1))
            u = _u01(_mix64(k + np.uint64(2 * t + 2)))
            w = c[idx] * x + 1.0          # fma(c, x, 1): the product c*x is exact to 1 ulp of a sub-unit term
            pos = w > 0.0
            v = w * w * w
            lv = np.log(np.where(pos, v, 1.0))
            lg[idx[pos]] = ld[idx[pos]] + lv[pos]
            acc = pos & (np.log(u) < (0.5 * x * x + d[idx]) - d[idx] * v + d[idx] * lv)
            live[idx[acc]] = False
        boost = np.log(_u01(_mix64(keys))) / a
    return np.where(small, lg + boost, lg)


def _cell_key(seed, model, sample, cell):
    with np.errstate(over="ignore"):
        return _mix64(_mix64(_mix64(np.uint64(seed) + np.asarray(model, dtype=np.uint64)) + np.asarray(sample, dtype=np.uint64))
                      ^ np.asarray(cell, dtype=np.uint64))


def log_gamma_hash(concs, size, seed):
    """log_gamma.log_gamma(concs, size) (log_gamma.py:17-76) on the counter stream of bear_log_gamma_f64:
    out[s, i] keyed by (seed, model 0, sample s, element i); shape = size + concs.shape (log_gamma.py:31, 76)."""
    concs = np.asarray(concs, dtype=np.float64)
    n_s = int(np.prod(size)) if len(size) else 1
    flat = concs.reshape(-1)
    s = np.repeat(np.arange(n_s, dtype=np.uint64), flat.size)
    i = np.tile(np.arange(flat.size, dtype=np.uint64), n_s)
    out = log_gamma_keys(np.tile(flat, n_s), _cell_key(seed, 0, s, i))
    return out.reshape(tuple(int(v) for v in size) + concs.shape)


def get_pdf_concs(counts, ar_vals, h, vans, get_map):
    """Concentrations of get_var_probs.get_pdf (get_var_probs.py:134-153): [num_models, K, A+1] in the order
    (AR if get_map and an AR function is present), BEAR h_0.., vanilla van_0..  counts: training column [K, A+1]."""
    counts = np.asarray(counts, dtype=np.float64)
    parts = []
    if ar_vals is not None:
        parts.append(np.asarray(ar_vals)[None, :, :] / np.asarray(h, dtype=np.float64)[:, None, None])
    if len(vans) > 0:
        parts.append(np.asarray(vans, dtype=np.float64)[:, None, None] * np.ones(counts.shape)[None, ...])
    concs = np.concatenate(parts, axis=0) + counts[None, :, :]
    if ar_vals is not None and get_map:
        concs = np.concatenate([np.asarray(ar_vals)[None, ...], concs], axis=0)
    return concs


def get_pdf_numpy(counts, ar_vals, h, vans, mc_samples, get_map, seed=0, row_base=0):
    """get_var_probs.get_pdf(..., output='numpy') (get_var_probs.py:128-183): normalised log transition
    probabilities [K, A+1, num_models, mc_samples]; sampling on the counter stream of bear_logdir_sample_f64
    (key = (seed, model, sample, (row_base + k) * 5 + letter))."""
    concs = get_pdf_concs(counts, ar_vals, h, vans, get_map)
    M, K, A1 = concs.shape
    if get_map:
        lp = np.log(concs / np.sum(concs, axis=-1)[..., None]).reshape(1, M, K, A1)      # :174-175
        return np.transpose(lp, [2, 3, 1, 0])
    s, m, k, b = np.meshgrid(np.arange(mc_samples, dtype=np.uint64), np.arange(M, dtype=np.uint64),
                             np.arange(K, dtype=np.uint64), np.arange(A1, dtype=np.uint64), indexing="ij")
    keys = _cell_key(seed, m, s, (np.uint64(row_base) + k) * np.uint64(A1) + b)
    g = log_gamma_keys(np.broadcast_to(concs[None], s.shape).reshape(-1), keys.reshape(-1)).reshape(s.shape)
    mx = g.max(axis=-1, keepdims=True)
    lse = mx + np.log(np.exp(g - mx).sum(axis=-1, keepdims=True))                        # :177-178
    return np.transpose(g - lse, [2, 3, 1, 0])                                           # :181-183


# --------------------------------------------------------------------------- row shuffle
def shuffle_perm(n, seed):
    """Source row of every shuffled row, ``dst[i] = src[perm[i]]``: the keyed bijection of bear_shuffle_rows
    (kernels_shuffle.h) -- a 4-round Feistel network on 2*half_bits >= log2(n) bits, cycle-walked into [0, n).
    Stands in for the `shuf` of docs/usage.rst:191-200 (any uniform-looking permutation serves that purpose)."""
    n = int(n)
    hb = 1
    while hb < 32 and (1 << (2 * hb)) < n:
        hb += 1
    mask = np.uint64((1 << hb) - 1)
    x = np.arange(n, dtype=np.uint64)
    todo = np.ones(n, dtype=bool)
    with np.errstate(over="ignore"):
        while todo.any():
            idx = np.nonzero(todo)[0]
            v = x[idx]
            l, r = v >> np.uint64(hb), v & mask
            for k in range(4):
                f = _mix64(np.uint64(seed) ^ (r + np.uint64((k + 1) << 58))) & mask
                l, r = r, l ^ f
            v = (l << np.uint64(hb)) | r
            x[idx] = v
            todo[idx] = v >= np.uint64(n)
    return x.astype(np.int64)


# --------------------------------------------------------------------------- summarize
def count_transitions(seqs, groups, max_lag, reverse=False):
    """The k-mer transition counts summarize.py produces, as its own test computes them in memory
    (bear_model/tests/test_summarize.py:88-115; check_summarize.py:36-66): for every lag L and sequence,
    ``full = '[' * L + seq + ']'`` and ``counts[full[j-L:j]][group][full[j]] += 1``; with ``reverse`` the reverse
    complement of every sequence is counted as well.  Returns a list (index L-1) of dicts kmer -> int array [n_groups, 5]."""
    letters = {"A": 0, "C": 1, "G": 2, "T": 3, "]": 4}            # summarize.py:380
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    n_groups = max(groups) + 1
    out = [dict() for _ in range(max_lag)]
    for li in range(max_lag):
        lag = li + 1
        for seq, g in zip(seqs, groups):
            variants = [seq] + (["".join(comp[c] for c in reversed(seq))] if reverse else [])
            for s in variants:
                full = "[" * lag + s + "]"
                for j in range(lag, len(full)):
                    row = out[li].setdefault(full[j - lag:j], np.zeros((n_groups, 5), dtype=np.int64))
                    row[g, letters[full[j]]] += 1
    return out

"""Likelihood core: host mirror of ``bear_model/core.py`` over the HIP kernels.

Same names and argument meaning as the reference (``core.py:11-174``) with torch tensors in place of
TF tensors: ``DirichletMultinomialPerm`` / ``MultinomialPerm`` with ``counts_log_prob`` and
``ml_output``, the alphabets, and the one-hot encoder.  The lgamma / digamma arithmetic runs in
``libbear_hip.so`` (``bear_dm_items_f64``); there is no CPU path.
"""
import numpy as np
import torch

from . import kernels

epsilon = 1e-7  # tf.keras.backend.epsilon(), core.py:8

# core.py:142-153 -- the last symbol is the start token '[' when encoding contexts and the stop token ']'
# when naming output columns
alphabets_tf = {
    "prot": np.array([b"A", b"R", b"N", b"D", b"C", b"E", b"Q", b"G", b"H", b"I", b"L",
                      b"K", b"M", b"F", b"P", b"S", b"T", b"W", b"Y", b"V", b"["]),
    "dna": np.array([b"A", b"C", b"G", b"T", b"["]),
    "rna": np.array([b"A", b"C", b"G", b"U", b"["]),
}
alphabets_en = {
    "prot": np.array(["A", "R", "N", "D", "C", "E", "Q", "G", "H", "I", "L",
                      "K", "M", "F", "P", "S", "T", "W", "Y", "V", "]"]),
    "dna": np.array(["A", "C", "G", "T", "]"]),
    "rna": np.array(["A", "C", "G", "U", "]"]),
}


def _as_counts_u32(value):
    v = torch.as_tensor(value)
    if v.dtype.is_floating_point:
        r = v.round()
        if not torch.equal(r, v):
            raise ValueError("transition counts must be integers")
        v = r
    if (v < 0).any() or (v > 4294967295).any():
        raise ValueError("transition counts must lie in [0, 2^32)")
    return v.to(torch.int64)


class DirichletMultinomialPerm:
    """``core.tfpDirichletMultinomialPerm`` (core.py:11-74): DM distribution over *ordered* transition
    sequences.  ``counts_log_prob(value)`` = TFP ``DirichletMultinomial.log_prob`` minus the multinomial
    coefficient = ``lbeta(concentration + value) - lbeta(concentration)`` (core.py:73-74)."""

    def __init__(self, total_count, concentration, validate_args=False, allow_nan_stats=True,
                 name="DirichletMultinomialPerm"):
        self.total_count = total_count
        self.concentration = torch.as_tensor(concentration)
        self.alphabet_size = self.concentration.shape[-1] - 1
        self.dtype = self.concentration.dtype
        self.name = name

    def _sample_n(self, n, seed=None, dummy=True):
        """core.py:64-67: a dummy sampler of zeros."""
        shape = (n,) + tuple(torch.as_tensor(self.total_count).shape) + (self.alphabet_size + 1,)
        return torch.zeros(shape, dtype=self.dtype, device=self.concentration.device)

    def ml_output(self):
        """core.py:69-71: argmax of the concentration, ties broken by Gaussian noise of scale 100 eps."""
        noise = 100 * epsilon * torch.randn_like(self.concentration)
        return torch.argmax(self.concentration + noise, dim=-1).to(self.dtype)

    def counts_log_prob(self, value):
        conc = self.concentration
        if not conc.is_cuda:
            raise RuntimeError("bear_amd computes on an MI355X only: move the concentration to a CUDA/HIP device")
        c = _as_counts_u32(value).to(conc.device)
        conc_b, c_b = torch.broadcast_tensors(conc.to(torch.float64), c)
        A = conc_b.sum(-1)
        n = c_b.sum(-1)
        if (n > 4294967295).any():
            raise ValueError("row totals beyond 2^32 are only supported by the fused table kernels")
        x = torch.cat([conc_b.reshape(-1), A.reshape(-1)]).contiguous()
        cc = torch.cat([c_b.reshape(-1), n.reshape(-1)])
        # uint32 bit pattern carried in int32 storage
        cc32 = torch.where(cc >= 2 ** 31, cc - 2 ** 32, cc).to(torch.int32).contiguous()
        D, _ = kernels.dm_items(x, cc32)
        m = conc_b.numel()
        out = D[:m].reshape(conc_b.shape).sum(-1) - D[m:].reshape(A.shape)
        return out.to(self.dtype)


class MultinomialPerm:
    """``core.tfpMultinomialPerm`` (core.py:77-139): ordered multinomial, ``sum_b c_b log p_b`` with a zero
    count times log 0 contributing 0 (TFP ``multiply_no_nan``)."""

    def __init__(self, total_count, probs, validate_args=False, allow_nan_stats=True, name="MultinomialPerm"):
        self.total_count = total_count
        self.probs = torch.as_tensor(probs)
        self.alphabet_size = self.probs.shape[-1] - 1
        self.dtype = self.probs.dtype
        self.name = name

    def _sample_n(self, n, seed=None):
        shape = (n,) + tuple(torch.as_tensor(self.total_count).shape) + (self.alphabet_size + 1,)
        return torch.zeros(shape, dtype=self.dtype, device=self.probs.device)

    def ml_output(self):
        """core.py:134-136."""
        return torch.argmax(self.probs + epsilon * torch.randn_like(self.probs), dim=-1).to(self.dtype)

    def counts_log_prob(self, value):
        v = torch.as_tensor(value, dtype=self.probs.dtype, device=self.probs.device)
        return torch.special.xlogy(v, self.probs).sum(-1)


# reference spellings
tfpDirichletMultinomialPerm = DirichletMultinomialPerm
tfpMultinomialPerm = MultinomialPerm


def encode_kmers(kmers, alphabet="dna"):
    """k-mer strings (list / numpy bytes array / uint8 [N, lag] ASCII matrix) -> int8 codes [N, lag]:
    0..A-1 letters, A = start symbol '[', -1 = anything else (all-zero one-hot row, core.py:173)."""
    letters = b"".join(alphabets_tf[alphabet].tolist())
    lut = np.full(256, -1, dtype=np.int8)
    for i, ch in enumerate(letters):
        lut[ch] = i
    if isinstance(kmers, np.ndarray) and kmers.dtype == np.uint8 and kmers.ndim == 2:
        arr = kmers
    else:
        ks = [k if isinstance(k, bytes) else str(k).encode() for k in np.asarray(kmers).reshape(-1).tolist()]
        if not ks:
            return np.zeros((0, 0), dtype=np.int8)
        lag = len(ks[0])
        if any(len(k) != lag for k in ks):
            raise ValueError("k-mers must all have the same length")
        arr = np.frombuffer(b"".join(ks), dtype=np.uint8).reshape(len(ks), lag)
    return lut[arr]


def tf_one_hot(seq, alphabet, dtype=torch.float64, device=None):
    """core.tf_one_hot (core.py:156-174): strings -> one-hot [..., lag, alphabet_size + 1].  Accepts
    strings or the int8 codes of ``encode_kmers``."""
    codes = seq if (isinstance(seq, (np.ndarray, torch.Tensor)) and np.asarray(seq.cpu() if isinstance(seq, torch.Tensor) else seq).dtype == np.int8) \
        else encode_kmers(seq, alphabet)
    codes = torch.as_tensor(np.asarray(codes.cpu() if isinstance(codes, torch.Tensor) else codes), device=device).long()
    A1 = len(alphabets_tf[alphabet])
    oh = torch.zeros(codes.shape + (A1,), dtype=dtype, device=codes.device)
    valid = codes >= 0
    oh.scatter_(-1, codes.clamp(min=0).unsqueeze(-1), valid.unsqueeze(-1).to(dtype))
    return oh


one_hot = tf_one_hot

"""log-Gamma sampler: host mirror of ``bear_model/log_gamma.py``.

``log_gamma(concs, size=[])`` keeps the reference signature and shape rule (log_gamma.py:17, 31, 76:
draws of shape ``size + concs.shape``), sampling ``log(Gamma(conc, 1))`` accurately for tiny
concentrations.  The draws come from one launch of ``bear_log_gamma_f64`` (kernels_sample.h); the
reference consumes numpy's global generator, here the stream is counter-based and seeded explicitly
(``seed=None`` draws a fresh seed from numpy's global generator, so ``np.random.seed`` still makes runs
reproducible).
"""
import numpy as np
import torch

from . import kernels


def _next_seed():
    return int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))


def log_gamma_pdf(conc, xs):
    """log_gamma.py:14-15."""
    conc, xs = torch.as_tensor(conc, dtype=torch.float64), torch.as_tensor(xs, dtype=torch.float64)
    return torch.exp(conc * xs - torch.exp(xs) - torch.lgamma(conc))


def log_gamma(concs, size=[], seed=None, device=None, as_numpy=True):
    """log_gamma.log_gamma (log_gamma.py:17-76).  Returns a numpy array like the reference
    (``as_numpy=False``: the device tensor)."""
    if not torch.cuda.is_available():
        raise RuntimeError("bear_amd samples on an MI355X only (libbear_hip.so has no CPU fallback)")
    if isinstance(concs, torch.Tensor) and concs.is_cuda:
        dev_concs = concs.to(torch.float64)
    else:
        dev_concs = torch.as_tensor(np.asarray(concs, dtype=np.float64), device=torch.device(device or "cuda"))
    shape = tuple(int(v) for v in size) + tuple(dev_concs.shape)
    n_samples = int(np.prod(size)) if len(size) else 1
    out = kernels.log_gamma(dev_concs.reshape(-1).contiguous(), n_samples, _next_seed() if seed is None else seed)
    out = out.reshape(shape)
    return out.cpu().numpy() if as_numpy else out

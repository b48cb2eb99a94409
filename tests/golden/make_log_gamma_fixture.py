"""Generates tests/golden/log_gamma_reference_quantiles.npz by running the REFERENCE sampler
(/root/reference/bear_model/log_gamma.py, importable without TensorFlow) in the build container.

The reference cannot travel to the GPU box, so the vectors are committed: for each concentration of the
reference's own test (bear_model/tests/test_log_gamma.py:10) the 2001 equally spaced quantiles and the
mean / variance of 400000 draws of log_gamma.log_gamma.  Run:  python tests/golden/make_log_gamma_fixture.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from bear_model import log_gamma  # noqa: E402

CONCS = np.array([0.01, 0.1, 0.5, 0.99, 1, 5, 100])   # test_log_gamma.py:10
N = 400000


def main():
    np.random.seed(20211012)
    tile = (np.ones([len(CONCS), N]) * CONCS[:, None]).flatten()
    draws = log_gamma.log_gamma(tile, size=[1]).reshape(len(CONCS), N)
    q = np.linspace(0.0, 1.0, 2001)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "log_gamma_reference_quantiles.npz")
    np.savez_compressed(out, concs=CONCS, n=N, probs=q, quantiles=np.quantile(draws, q, axis=1).T,
                        mean=draws.mean(axis=1), var=draws.var(axis=1))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

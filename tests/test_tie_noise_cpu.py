"""The counter-based N(0,1) stream behind the arg-max tie-breaking (`oracle.eval_noise`, which restates the HIP kernels' stream
bit for bit; the reference draws `tf.random.normal`, core.py:69-71, 134-136) as a random stream in its own right: marginally
standard normal (Kolmogorov-Smirnov), and independent across letters, rows, models and seeds -- the properties the reference's
tie-breaking test relies on (tests/test_core.py:29-39).  The kernels' own draws are held to the reference's criterion in
tests/test_tie_noise_gpu.py."""
import os
import sys

import numpy as np
import scipy.stats as st

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import bear_oracle as o

N = 200_000


def test_eval_noise_is_standard_normal():
    for seed, model, row0 in ((0, 0, 0), (11, o.EVAL_ID_ARM, 10 ** 9), (20211012, o.EVAL_ID_VAN + 2, 2 ** 40)):
        z = o.eval_noise(seed, model, np.arange(row0, row0 + N, dtype=np.uint64))
        assert z.shape == (N, 5) and np.isfinite(z).all()
        for b in range(5):
            assert st.kstest(z[:, b], "norm").pvalue > 1e-3, (seed, model, b)
        assert st.kstest(z.reshape(-1), "norm").pvalue > 1e-3
        assert abs(z.mean()) < 5 / np.sqrt(5 * N) and abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / (5 * N))
        # tails: the Box-Muller draw from 32-bit uniforms reaches beyond 4 sigma as often as it should
        assert abs((np.abs(z) > 3.0).mean() - 2 * st.norm.sf(3.0)) < 5 * np.sqrt(2 * st.norm.sf(3.0) / (5 * N))


def test_eval_noise_streams_are_uncorrelated():
    rows = np.arange(N, dtype=np.uint64)
    base = o.eval_noise(5, 0, rows)
    lim = 5.0 / np.sqrt(N)              # five standard deviations of a sample correlation of independent streams
    # letters of one row
    c = np.corrcoef(base.T)
    assert np.abs(c - np.eye(5)).max() < lim
    # neighbouring rows, other models, other seeds: letter by letter and across letters
    others = [o.eval_noise(5, 0, rows + np.uint64(1)), o.eval_noise(5, 1, rows), o.eval_noise(5, o.EVAL_ID_ARM, rows),
              o.eval_noise(5, o.EVAL_ID_VAN, rows), o.eval_noise(5, o.EVAL_ID_VAN + 1, rows), o.eval_noise(6, 0, rows),
              o.eval_noise(5 + o.EVAL_ID_ARM, 0, rows)]
    for k, z in enumerate(others):
        cc = np.corrcoef(base.T, z.T)[:5, 5:]
        if k == 0:
            cc = cc.copy()              # (row r + 1 of the shifted stream IS row r + 1 of the base stream one row down: compare unshifted)
        assert np.abs(cc).max() < lim, (k, np.abs(cc).max())
    # the arg-max of a tied pair is a fair coin whose flips do not depend on the neighbouring row's
    win = base[:, 0] > base[:, 2]
    assert abs(win.mean() - 0.5) < 5 * 0.5 / np.sqrt(N)
    assert abs(np.corrcoef(win[:-1], win[1:])[0, 1]) < lim

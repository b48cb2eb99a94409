"""Seeded input generators shared by the CPU and GPU tests (NumPy only)."""
import numpy as np


def sparse_table(n, seed=0, lam_scale=1.0):
    """NumPy twin (same recipe, not the same bits) of the 'k=13 sparse' synthetic table."""
    rng = np.random.default_rng(seed)
    lam = np.exp(0.5 + 1.5 * rng.standard_normal(n)) * lam_scale
    w = -np.log(rng.random((n, 4))) * rng.random((n, 4)) ** (10 / 3)
    p = w / w.sum(1, keepdims=True) * (1 - 1 / 150)
    p = np.concatenate([p, np.full((n, 1), 1 / 150)], 1)
    train = rng.poisson(lam[:, None] * p).astype(np.uint32)
    test = rng.poisson(lam[:, None] * p / 3).astype(np.uint32)
    ref = rng.poisson(0.02 * lam[:, None] * p).astype(np.uint32)
    ref[:, 4] = 0
    return train, test, ref


def dense_table(n, seed=0):
    rng = np.random.default_rng(seed)
    lam = 1e4 * np.exp(rng.random(n) * np.log(30))
    p = rng.dirichlet(np.full(5, 2.0), size=n)
    train = rng.poisson(lam[:, None] * p).astype(np.uint32)
    ref = rng.poisson(1e-3 * lam[:, None] * p).astype(np.uint32)
    ref[:, 4] = 0
    return train, ref


def prior_rows(n, seed=0, conc=1.0):
    rng = np.random.default_rng(seed + 1000)
    return rng.dirichlet(np.full(5, conc), size=n)


def edge_table(seed=0):
    """Rows that force every branch: all-zero, single transition, product/Stirling boundary,
    huge counts (uint32 range), one-hot rows."""
    rng = np.random.default_rng(seed)
    rows = [
        [0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [0, 0, 0, 0, 1], [1, 1, 1, 1, 1],
        [16, 0, 0, 0, 0], [17, 0, 0, 0, 0], [0, 16, 1, 0, 0], [8, 8, 0, 0, 1],
        [15, 16, 17, 18, 19], [100, 0, 3, 0, 0], [1000, 2000, 3000, 4000, 5],
        [4000000000, 0, 0, 0, 0], [4000000000, 4000000000, 4000000000, 4000000000, 100000],
        [254715, 3, 0, 1, 0], [0, 0, 65, 0, 0], [2, 0, 0, 31, 33],
    ]
    extra = rng.integers(0, 40, size=(48, 5))
    return np.asarray(rows + extra.tolist(), dtype=np.uint32)

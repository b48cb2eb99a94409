"""Dev helper: randomized planned-vs-oracle stress over table shapes (run on the GPU box)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import c_oracle as co
from bear_amd import kernels
from util import sparse_table, dense_table, prior_rows
dev = torch.device("cuda", 0)
rng = np.random.default_rng(123)
def to_dev(a):
    return torch.from_numpy(a.view(np.int32).copy() if a.dtype == np.uint32 else np.ascontiguousarray(a)).to(dev)
worst = 0.0
for it in range(60):
    n = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 1663, 1664, 1665, 4096, 10007, 50021, 200003]))
    kind = rng.integers(0, 5)
    if kind == 0: tr, _, rf = sparse_table(n, int(rng.integers(1e9)), lam_scale=float(rng.choice([0.05, 0.3, 1, 4, 20])))
    elif kind == 1: tr, rf = dense_table(n, int(rng.integers(1e9)))
    elif kind == 2:
        tr, _, rf = sparse_table(n, int(rng.integers(1e9))); tr[rng.random(n) < 0.5] = 0      # many empty contexts
    elif kind == 3:
        tr, _, rf = sparse_table(n, int(rng.integers(1e9))); d, _ = dense_table(n, 7); m = rng.random(n) < 0.1; tr[m] = d[m]  # mixed
    else:
        tr = np.zeros((n, 5), np.uint32); rf = np.zeros((n, 5), np.uint32); tr[:, int(rng.integers(5))] = rng.integers(0, 60, n)
    f = prior_rows(n, int(rng.integers(1e9)), float(rng.choice([0.2, 1, 5])))
    if rng.random() < 0.3: f = f * rng.uniform(0.5, 2.0, size=(n, 1))
    h = float(rng.uniform(-4, 3)); args = (h, float(rng.uniform(-5, 1)), float(rng.uniform(-6, 1)))
    dtr, drf, df = to_dev(tr), to_dev(rf), to_dev(f)
    wr = co.dm_ref(tr, rf, *args, nthreads=4); wn, wg = co.dm_prior(tr, f, h, want_grad=True, nthreads=4)
    gr = kernels.dm_ref_planned(kernels.Plan(dtr, 4), drf, *args).cpu().numpy()
    pn = kernels.Plan(dtr, 5)
    gn = kernels.dm_prior_planned(pn, df, h).cpu().numpy()
    gg, g = kernels.dm_prior_planned(pn, df, h, want_grad=True); gg, g = gg.cpu().numpy(), g.cpu().numpy()
    sc_r = np.abs(wr[1:]).max() + abs(wr[0]) * 1e-3 + 1e-300
    errs = [abs(gr[0] - wr[0]) / (abs(wr[0]) + 1e-300), np.abs(gr[1:] - wr[1:]).max() / sc_r * 1e-2,
            abs(gn[0] - wn[0]) / (abs(wn[0]) + 1e-300), abs(gn[1] - wn[1]) / (abs(wn[1]) + abs(wn[0]) * 1e-3 + 1e-300) * 1e-2,
            abs(gg[0] - wn[0]) / (abs(wn[0]) + 1e-300), np.abs(g - wg).max() / (np.abs(wg).max() + 1e-300) * 1e-2]
    e = max(errs); worst = max(worst, e)
    if e > 1e-11: print("BAD", it, n, kind, errs)
print("worst scaled error", worst)

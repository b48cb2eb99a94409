import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bear_amd") else os.getcwd())
import numpy as np, torch, mpmath as mp
from bear_amd import kernels
mp.mp.dps = 40
xs = np.array([1e-7, 3e-5, 0.013, 0.5, 1.0, 4.0, 7.99, 8.0, 8.01, 31.4, 250.0, 1e4, 1e7, 2.0**30, 2.0**31, 1e12])
cs = np.array([1, 16, 17, 24, 25, 32, 40, 100, 1000, 254715, 10**7, 4_000_000_000], dtype=np.uint64)
X, C = np.meshgrid(xs, cs, indexing="ij"); X, C = X.ravel(), C.ravel()
Dw = np.array([float(mp.loggamma(mp.mpf(float(x)) + int(c)) - mp.loggamma(mp.mpf(float(x)))) for x, c in zip(X, C)])
Pw = np.array([float(mp.digamma(mp.mpf(float(x)) + int(c)) - mp.digamma(mp.mpf(float(x)))) for x, c in zip(X, C)])
dev = torch.device("cuda", 0)
dx = torch.from_numpy(X).to(dev); dc = torch.from_numpy(C.astype(np.uint32).view(np.int32)).to(dev)
for path in (1, 2):
    D, P = kernels.dm_items(dx, dc, path=path); D, P = D.cpu().numpy(), P.cpu().numpy()
    rd = np.abs(D - Dw) / (np.abs(Dw) + np.abs([float(mp.loggamma(float(x))) for x in X]) + 1e-300); rp = np.abs(P - Pw) / np.abs(Pw)
    i, j = rd.argmax(), rp.argmax()
    print("path", path, "max relD %.2e at x=%g c=%d (D=%g)" % (rd[i], X[i], C[i], Dw[i]), " max relP %.2e at x=%g c=%d" % (rp[j], X[j], C[j]))
    bad = np.where(rp > 1e-13)[0][:8]
    for b in bad: print("   P off: x=%g c=%d got %.17g want %.17g" % (X[b], C[b], P[b], Pw[b]))

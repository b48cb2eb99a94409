"""Dev helper: where the 0.4 s of reading a 1e8-row table go: row count (its own mapping), sniffing, allocation, parse."""
import ctypes, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bear_amd import _lib, dataloader
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(0)
km = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 13))]
c = rng.poisson(1.2, size=(3, n, 5)).astype(np.uint32)
fd, path = tempfile.mkstemp(suffix=".tsv"); os.close(fd)
dataloader.write_counts_tsv(path, km, c)
del km, c
L = _lib.lib()
for rep in range(2):
    t0 = time.perf_counter(); rows = dataloader.count_rows(path); t1 = time.perf_counter()
    kmers = np.zeros((rows, 13), dtype=np.uint8); counts = np.zeros((3, rows, 5), dtype=np.uint32); t2 = time.perf_counter()
    got = ctypes.c_uint64()
    _lib.check(L.bear_parse_counts_tsv(path.encode(), 3, 13, rows, kmers.ctypes.data, counts.ctypes.data, ctypes.byref(got)), "parse"); t3 = time.perf_counter()
    print("count_rows %.3f s | np.zeros %.3f s | parse %.3f s | total %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t3 - t0))
    del kmers, counts
os.remove(path)

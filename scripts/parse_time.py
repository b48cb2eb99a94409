"""Host-side: throughput of the threaded count-table reader (bear_parse_counts_tsv) by thread count."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bear_amd import _lib
rng = np.random.default_rng(0)
n, lag = int(float(sys.argv[1])) if len(sys.argv) > 1 else 8_000_000, 13
letters = np.frombuffer(b"ACGT", dtype=np.uint8)
kmers = letters[rng.integers(0, 4, size=(n, lag))]
counts = rng.poisson(1.5, size=(3, n, 5)).astype(np.uint32)
L = _lib.lib(); path = "/tmp/bear_parse_time.tsv"
L.bear_write_counts_tsv(path.encode(), kmers.ctypes.data, counts.ctypes.data, n, lag, 3, 0, 1, 0)
km = np.ones((n, lag), dtype=np.uint8); cn = np.ones((3, n, 5), dtype=np.uint32); got = ctypes.c_uint64()
mb = os.path.getsize(path) / 1e6
for th in ("1", "2", "4", "8", "16", "32"):
    os.environ["BEAR_PARSE_THREADS"] = th
    best = 1e9
    for _ in range(2):
        t = time.time(); st = L.bear_parse_counts_tsv(path.encode(), 3, lag, n, km.ctypes.data, cn.ctypes.data, ctypes.byref(got)); best = min(best, time.time() - t)
    assert st == 0 and np.array_equal(cn, counts)
    print(th, "threads: %.3f s  %.0f MB/s  %.1f Mrows/s" % (best, mb / best, n / best / 1e6), flush=True)
os.remove(path)
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))

"""Dev helper: parse time of a 1e8-row table against the number of host threads (BEAR_PARSE_THREADS)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
code = r'''
import os, sys, time
sys.path.insert(0, %r)
from bear_amd import dataloader
t0 = time.perf_counter(); d = dataloader.dataloader(sys.argv[1], "dna", 10**9, 3); print("%%s threads: parse %%.3f s" %% (os.environ.get("BEAR_PARSE_THREADS"), time.perf_counter() - t0))
''' % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bear_amd import dataloader
rng = np.random.default_rng(0)
km = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 13))]
c = rng.poisson(1.2, size=(3, n, 5)).astype(np.uint32)
fd, path = tempfile.mkstemp(suffix=".tsv"); os.close(fd)
t0 = time.perf_counter(); dataloader.write_counts_tsv(path, km, c); print("write %.2f s, %.2f GB" % (time.perf_counter() - t0, os.path.getsize(path) / 1e9))
del km, c
for nt in (16, 32, 64, 128, 256):
    subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, BEAR_PARSE_THREADS=str(nt)))
os.remove(path)

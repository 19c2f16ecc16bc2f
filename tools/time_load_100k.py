import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from fastsk_amd import _native
rng = np.random.Generator(np.random.PCG64(20201214))
X = rng.integers(1, 5, size=(100000, 300), dtype=np.int32)
tok = X.reshape(-1); off = np.arange(100001, dtype=np.int64) * 300
e = _native.Engine(12, 8)
for _ in range(4):
    t0 = time.perf_counter(); e.load_sequences(tok, off, 100000, 0); print("load 100k x 300: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
e.close()

#!/usr/bin/env python3
"""Config-5 count kernel on its own (for rocprofv3 --pmc passes): all 495 combos' count panels,
then the tiles of the first 128 rows only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsk_amd import _native
N, L = int(os.environ.get("N", 100000)), 300
rng = np.random.Generator(np.random.PCG64(20201214))
X = rng.integers(1, 5, size=(N, L), dtype=np.int32)
e = _native.Engine(12, 8, profile=False)
e.load_sequences(X.reshape(-1), np.arange(N + 1, dtype=np.int64) * L, N, 0)
combos = np.arange(495, dtype=np.int32)
for _ in range(int(os.environ.get("REPS", 3))):
    e.reset_counts()
    e.accumulate_rows(combos, 0, 128)
    e.synchronize()
print(e.stats()["ms_count"])

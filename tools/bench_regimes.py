#!/usr/bin/env python3
"""Dense vs sparse dataflow across key-space sizes (DNA, k = 4..8), to calibrate the cost model
behind path=auto (dense_is_cheaper in fsk_engine.hip). Prints seconds per combo for both."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsk_amd import _native  # noqa: E402


def run(tokens, offsets, N, g, m, combos, path):
    e = _native.Engine(g, m, path=path, profile=False)
    e.load_sequences(tokens, offsets, N, 0)
    e.accumulate(combos)  # warm-up with the same list: every buffer reaches its final size
    e.synchronize()
    e.reset_counts()
    t0 = time.perf_counter()
    e.accumulate(combos)
    e.synchronize()
    dt = time.perf_counter() - t0
    st = e.stats()
    e.close()
    return dt / len(combos), st


def main():
    rng = np.random.default_rng(0)
    for N, L in [(4000, 100), (4000, 300), (16000, 200)]:
        X = rng.integers(1, 5, size=(N, L), dtype=np.int32)
        tokens, offsets = _native.flatten(X)
        for k in (4, 5, 6, 7, 8):
            g, m = k + 4, 4
            nc = _native.library().num_combos(g, m)
            combos = np.arange(min(nc, 24), dtype=np.int32)
            row = dict(N=N, L=L, k=k, V=4 ** k)
            for name, path in (("dense", 1), ("sparse", 2), ("auto", 0)):
                if path == 1 and 4 ** k > 16384:
                    continue
                dt, st = run(tokens, offsets, N, g, m, combos, path)
                row[name + "_ms_per_combo"] = round(dt * 1e3, 3)
                if path == 0:
                    row["auto_picked"] = "dense" if st["path_used"] == 1 else "sparse"
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

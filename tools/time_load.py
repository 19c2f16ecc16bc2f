#!/usr/bin/env python3
"""Host time of fsk_load_sequences (alphabet scan, bit-packing, H2D enqueued) on BASELINE configs 1-4 and on the
100,000 x 300 workload of config 5: best of several calls.   tools/time_load.py [--100k]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
for name in ["f7_cfg2_ep300_exact", "f7_cfg3_ep47848_100combos", "f7_cfg4_prot219_exact", "f7_cfg1_prot11_approx_t1"]:
    d = load_golden(name)
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    e = _native.Engine(d["g"], d["m"])
    best = 1e9
    for _ in range(20):
        t0 = time.perf_counter(); e.load_sequences(tokens, offsets, ntr, nte); best = min(best, time.perf_counter() - t0)
    e.synchronize()
    print(name, "tokens", len(tokens), "load %.3f ms" % (best * 1e3))
    e.close()
if "--100k" in sys.argv:
    rng = np.random.Generator(np.random.PCG64(20201214))
    X = rng.integers(1, 5, size=(100000, 300), dtype=np.int32)
    tok, off = X.reshape(-1), np.arange(100001, dtype=np.int64) * 300
    e = _native.Engine(12, 8)
    for _ in range(4):
        t0 = time.perf_counter(); e.load_sequences(tok, off, 100000, 0); print("load 100k x 300: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    e.close()

#!/usr/bin/env python3
"""The reference's CI configuration (test/run_check.py:45): EP300, g=10 m=6, approx=True, t=1 — variance
mode on the dense dataflow. Wall time of compute_kernel through the C ABI (second call)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_tokens
from fastsk_amd import _native
tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
for path in (0, 2):
    e = _native.Engine(10, 6, t=1, approx=True, path=path)
    e.set_seed(7)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); e.compute(tokens, offsets, ntr, nte); best = min(best, time.perf_counter() - t0)
    st = e.stats()
    print({"path": "dense" if st["path_used"] == 1 else "sparse", "iterations": len(e.get_stdevs()), "ms": round(best * 1e3, 2),
           "ms_per_iteration": round(best * 1e3 / max(1, len(e.get_stdevs())), 3)})
    e.close()

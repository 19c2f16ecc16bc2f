#!/usr/bin/env python3
"""Time fsk_sequential_sum (device) against the host's left-to-right loop on config-1-sized input."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsk_amd import _native
e = _native.Engine(4, 2)
rng = np.random.default_rng(3)
for n in (2_736_630, 21_000_000):
    x = (rng.integers(0, 30, n) ** 2) * (1.0 - 1.0 / 11.0)
    x[rng.random(n) < 0.6] = 0.0
    t0 = time.perf_counter(); want = float(np.add.accumulate(x)[-1]); t_host = time.perf_counter() - t0
    e.sequential_sum(x)
    t0 = time.perf_counter(); got = e.sequential_sum(x); t_dev = time.perf_counter() - t0
    print({"n": n, "equal": got == want, "host_accumulate_ms": 1e3 * t_host, "device_call_ms_incl_H2D": 1e3 * t_dev})

#!/usr/bin/env python3
"""Where the wall time of the small dense configs goes: load_sequences (host packing + H2D + zeroing K),
accumulate (count + tile kernels), finalize — through the staged C ABI, best of 5."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
for name in ("f7_cfg2_ep300_exact", "f7_cfg3_ep47848_100combos"):
    d = load_golden(name)
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    e = _native.Engine(d["g"], d["m"])
    combos = np.arange(210, dtype=np.int32) if not d["approx"] else np.asarray(d["order"][:100], dtype=np.int32)
    best = [1e9] * 4
    for _ in range(5):
        t0 = time.perf_counter(); e.load_sequences(tokens, offsets, ntr, nte); e.synchronize()
        t1 = time.perf_counter(); e.accumulate(combos); e.synchronize()
        t2 = time.perf_counter(); e.finalize()
        t3 = time.perf_counter()
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t3 - t0)):
            best[i] = min(best[i], v)
    print(name, {"load_ms": round(best[0] * 1e3, 3), "accumulate_ms": round(best[1] * 1e3, 3), "finalize_ms": round(best[2] * 1e3, 3),
                 "total_ms": round(best[3] * 1e3, 3), "tokens": len(tokens)})
    e.close()

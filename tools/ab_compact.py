#!/usr/bin/env python3
"""A/B on one box: BASELINE config 3 (key-compacted panels) with the direct-to-LDS tile kernel
(FSK_COMPACT_DMA=1) and the register-staged one (default)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from conftest import load_golden, load_tokens
from fastsk_amd import _native
d = load_golden("f7_cfg3_ep47848_100combos")
tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
e = _native.Engine(d["g"], d["m"], t=d["t"], approx=True, max_iters=d["max_iters"], skip_variance=True, profile=True)
e.set_combo_order(d["order"])
best = 1e9
for _ in range(6):
    t0 = time.perf_counter(); e.compute(tokens, offsets, ntr, nte); best = min(best, time.perf_counter() - t0)
st = e.stats()
import hashlib
print(best * 1e3, st["ms_tile"] / 6, st["ms_count"] / 6, st["dense_macs"] / 6, hashlib.sha256(e.get_counts().tobytes()).hexdigest()[:16])
''' % (ROOT, ROOT)
for v in ("0", "1", "0", "1"):
    env = dict(os.environ, FSK_COMPACT_DMA=v)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("FSK_COMPACT_DMA=%s" % v, r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else "")

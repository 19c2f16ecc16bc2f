#!/usr/bin/env python3
"""Config 1 (variance mode, sparse dataflow) a few times with the tuning key trace=1: where its wall time goes."""
import os, sys, time
import numpy as np
os.environ["FSK_TUNING"] = "trace=1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
d = load_golden("f7_cfg1_prot11_approx_t1")
tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
e = _native.Engine(d["g"], d["m"], t=d["t"], approx=True, delta=d["delta"], max_iters=d["max_iters"])
e.set_combo_order(d["order"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(n):
    t0 = time.perf_counter(); e.compute(tokens, offsets, ntr, nte); t1 = time.perf_counter()
    print("run %.3f ms" % ((t1 - t0) * 1e3), file=sys.stderr)

#!/usr/bin/env python3
"""Run one BASELINE config once (PROFILE_CALLS=n: n times) through the C ABI (for rocprofv3 --pmc passes)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
name = sys.argv[1] if len(sys.argv) > 1 else "f7_cfg4_prot219_exact"
d = load_golden(name)
tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"],
                   skip_variance=bool(d["skip_variance"]))
if d["approx"]:
    e.set_combo_order(d["order"])
for _ in range(int(os.environ.get('PROFILE_CALLS', '1'))):
    e.compute(tokens, offsets, ntr, nte)
print(e.stats()["combos_done"])

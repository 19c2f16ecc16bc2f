#!/usr/bin/env python3
"""A/B of two builds of the engine on one box, one process: the in-tree library against a variant built with other
compile-time constants (tools/build_variant.sh), BASELINE configs 1 and 4, best of 7 whole fsk_compute calls, alternating."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
libs = {"base": _native.library()}
for p in sys.argv[1:]:
    libs[os.path.basename(p)] = _native.Library(p)
for name in os.environ.get("AB_CASES", "f7_cfg4_prot219_exact,f7_cfg1_prot11_approx_t1").split(","):
    if name == "large_g":  # EP300, k = 6, g = 20 (one call each; results of measurement builds are not checked)
        tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
        res = {}
        for k, lib in libs.items():
            e = _native.Engine(20, 14, lib=lib)
            e.compute(tokens, offsets, ntr, nte)
            t0 = time.perf_counter(); e.compute(tokens, offsets, ntr, nte); res[k] = round(time.perf_counter() - t0, 4)
            e.close()
        print(json.dumps({"case": name, **res}))
        continue
    d = load_golden(name)
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    eng = {}
    for k, lib in libs.items():
        e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"],
                           skip_variance=bool(d["skip_variance"]), lib=lib)
        if d["approx"]:
            e.set_combo_order(d["order"])
        e.compute(tokens, offsets, ntr, nte)
        eng[k] = e
    best = {k: 1e9 for k in libs}
    for _ in range(7):
        for k, e in eng.items():
            t0 = time.perf_counter(); e.compute(tokens, offsets, ntr, nte); best[k] = min(best[k], time.perf_counter() - t0)
    print(json.dumps({"case": name, **{k: round(v * 1e3, 3) for k, v in best.items()}}))
    for e in eng.values():
        e.close()

#!/usr/bin/env python3
"""Where k_sx_emit's cycles go, per phase, on BASELINE configs 1 and 4 (a measurement build: tools/build_variant.sh var_clk
-DFSK_EM_CLOCKS=1; the workgroups' first threads add clock64() deltas per phase into a device array).
tools/emit_phases.py fastsk_amd/lib/libvar_clk.so"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
lib = _native.Library(sys.argv[1])
raw = ctypes.CDLL(sys.argv[1])
names = ["load + classify", "fill slots + addresses", "word loop (short)", "list long", "long headers + half waves", "long entries", "zero + bin + scan", "", "", "", "", "", "", "", "", "workgroups"]
for name in os.environ.get("AB_CASES", "f7_cfg1_prot11_approx_t1,f7_cfg4_prot219_exact").split(","):
    if name == "large_g":  # the paper's large-g regime: EP300, k = 6, g = 20
        tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
        e = _native.Engine(20, 14, lib=lib)
    else:
        d = load_golden(name)
        tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
        e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"],
                           skip_variance=bool(d["skip_variance"]), lib=lib)
        if d["approx"]:
            e.set_combo_order(d["order"])
    e.compute(tokens, offsets, ntr, nte)
    out = (ctypes.c_ulonglong * 16)()
    raw.fsk_debug_emit_clocks(out)
    e.compute(tokens, offsets, ntr, nte)
    raw.fsk_debug_emit_clocks(out)
    v = np.array(list(out), dtype=np.float64)
    tot = v[:7].sum()
    print(name, "workgroups %d, cycles per workgroup %.0f" % (v[15], tot / max(1, v[15])))
    print("   longest workgroup %.0f cycles (%.1f x the mean); workgroups beyond 200 k cycles: %d, beyond 400 k: %d" % (v[14], v[14] / (tot / max(1, v[15])), v[13], v[12]))
    for i in (0, 6, 1, 2, 3, 4, 5):
        print("   %-28s %5.1f %%   %8.0f cycles a workgroup" % (names[i], 100 * v[i] / tot, v[i] / max(1, v[15])))
    e.close()

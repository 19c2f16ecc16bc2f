#!/usr/bin/env python3
"""A/B of environment settings on the paper's large-g regime (EP300, k = 6, g = 20: 38,760 combos, sparse dataflow, ~40
batches): best of 3 whole fsk_compute calls, one fresh process per setting.   tools/ab_large_g.py "K=V" "K=V K2=V2" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_tokens
from fastsk_amd import _native
tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
g, m = int(sys.argv[1]), int(sys.argv[2])
lib = _native.Library(os.environ['AB_LIB']) if os.environ.get('AB_LIB') else None
e = _native.Engine(g, m, lib=lib) if lib else _native.Engine(g, m)
e.compute(tokens, offsets, ntr, nte)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); e.compute(tokens, offsets, ntr, nte); best = min(best, time.perf_counter() - t0)
dg = e.counts_digest()
st = e.stats()
print(json.dumps({"g": g, "m": m, "s": round(best, 4), "launches": st["launches"] // 4, "redone": st["batches_redone"],
                  "digest": format(int(dg[0]), "x") + "." + format(int(dg[1]), "x")}))
'''.replace("ROOT", repr(ROOT))
gm = os.environ.get("AB_GM", "20,14").split(",")
for setting in [""] + sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("="); env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, gm[0], gm[1]], env=env, capture_output=True, text=True)
    print(setting or "(default)", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])

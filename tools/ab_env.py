#!/usr/bin/env python3
"""A/B of tuning keys of the engine on one box: best of 7 whole fsk_compute calls of BASELINE configs (or `large_g`: EP300,
k = 6, g = 20, best of 3), one fresh process per setting (FSK_TUNING is parsed by fsk_create).
tools/ab_env.py CASE[,CASE] "key=value,key2=value2" "key=value" ...   (the first run is always the defaults)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
out = {}
for name in sys.argv[1].split(","):
    if name == "large_g":   # the paper's large-g regime: EP300 (2000 + 2000 x 100 bp), k = 6, g = 20 -> 38,760 combos, sparse dataflow
        tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
        e = _native.Engine(20, 14)
        e.compute(tokens, offsets, ntr, nte)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); e.compute(tokens, offsets, ntr, nte); best = min(best, time.perf_counter() - t0)
        st = e.stats(); dg = e.counts_digest()
        out[name] = {"s": round(best, 4), "redone": st["batches_redone"], "desc": st["sparse_desc"], "digest": format(int(dg[0]), "x") + "." + format(int(dg[1]), "x")}
        e.close()
        continue
    d = load_golden(name)
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"], skip_variance=bool(d["skip_variance"]))
    if d["approx"]: e.set_combo_order(d["order"])
    e.compute(tokens, offsets, ntr, nte)
    best = 1e9
    for _ in range(7):
        t0 = time.perf_counter(); e.compute(tokens, offsets, ntr, nte); best = min(best, time.perf_counter() - t0)
    st = e.stats()
    dg = e.counts_digest() if not d["approx"] else (0, 0)
    out[name] = {"ms": round(best * 1e3, 3), "issued": st["combos_issued"], "done": st["combos_done"], "stdevs": len(e.get_stdevs()),
                 "launches": st["launches"] // 8, "redone": st["batches_redone"], "desc": st["sparse_desc"],
                 "stdev0": (float(e.get_stdevs()[-1]) if len(e.get_stdevs()) else 0.0), "digest": format(int(dg[0]), "x") + "." + format(int(dg[1]), "x")}
    e.close()
print(json.dumps(out))
''' % (ROOT, ROOT)
cases = sys.argv[1]
for setting in [""] + sys.argv[2:]:
    env = dict(os.environ)
    env.pop("FSK_TUNING", None)
    if setting:
        env["FSK_TUNING"] = setting
    r = subprocess.run([sys.executable, "-c", CHILD, cases], env=env, capture_output=True, text=True)
    print(setting or "(default)", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])

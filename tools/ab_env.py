#!/usr/bin/env python3
"""A/B of environment settings of the engine on one box: best of 7 whole fsk_compute calls of BASELINE configs, one fresh
process per setting (the engine reads its FSK_* variables at fsk_create).   tools/ab_env.py CASE[,CASE] "K=V K2=V2" "K=V" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
out = {}
for name in sys.argv[1].split(","):
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
                 "launches": st["launches"] // 8, "redone": st["batches_redone"], "digest": format(int(dg[0]), "x") + "." + format(int(dg[1]), "x")}
    e.close()
print(json.dumps(out))
''' % (ROOT, ROOT)
cases = sys.argv[1]
for setting in [""] + sys.argv[2:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("="); env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, cases], env=env, capture_output=True, text=True)
    print(setting or "(default)", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])

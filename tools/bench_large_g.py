#!/usr/bin/env python3
"""The reference paper's "large feature length" regime: EP300 DNA (2000+2000 x 100 bp), k = g-m = 6
kept positions, g up to 20 (C(20,14) = 38,760 combos, 4^6 = 4096 keys). Times the exact kernel on
one GPU and checks a random subset of combos against the oracle (sub-block of 600 sequences)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_tokens
from fastsk_amd import _native
from oracle import loader

tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
N = ntr + nte
for g, m in ((12, 6), (16, 10), (20, 14)):
    e = _native.Engine(g, m, profile=True)
    nc = e.lib.num_combos(g, m)
    t0 = time.perf_counter()
    e.compute(tokens, offsets, ntr, nte)
    dt_first = time.perf_counter() - t0   # includes the device allocations
    t0 = time.perf_counter()
    e.compute(tokens, offsets, ntr, nte)
    dt = time.perf_counter() - t0
    st = e.stats()
    # parity on a subset: 12 random combos, first 600 sequences, against the oracle
    rng = np.random.default_rng(g)
    sub = np.sort(rng.choice(nc, size=12, replace=False)).astype(np.int32)
    e2 = _native.Engine(g, m)
    e2.load_sequences(tokens[:offsets[600]], offsets[:601], 600, 0)
    e2.accumulate(sub); e2.finalize()
    want, _, _ = loader.port().raw_counts(tokens[:offsets[600]], offsets[:601], g, m, sub, threads=os.cpu_count())
    ok = bool(np.array_equal(e2.get_counts(), want))
    tr = e.get_block(0, 4, 0, 4)
    print(json.dumps(dict(g=g, m=m, combos=nc, N=N, seconds=dt, first_call_seconds=dt_first, combos_per_s=nc / dt, path="dense" if st["path_used"] == 1 else "sparse",
                          tile_ms=st["ms_tile"], count_ms=st["ms_count"], tile_launches=st["n_tile_launches"], subset_parity=ok,
                          diag_ok=bool(np.all(np.diag(tr) == 1.0)))), flush=True)
    e.close(); e2.close()

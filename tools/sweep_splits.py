#!/usr/bin/env python3
"""Tuning aid: time BASELINE configs 2 and 3 for several forced combo-split counts."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
for name in ("f7_cfg2_ep300_exact", "f7_cfg3_ep47848_100combos"):
    d = load_golden(name)
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    combos = np.sort(d["combos"]).astype(np.int32)
    for sp in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
        os.environ["FSK_TUNING"] = "tile_splits=%d" % sp
        e = _native.Engine(d["g"], d["m"], profile=True)
        e.load_sequences(tokens, offsets, ntr, nte)
        e.accumulate(combos); e.synchronize(); e.reset_counts()
        s0 = e.stats()
        for _ in range(3):
            e.accumulate(combos)
        e.synchronize()
        s1 = e.stats()
        print(json.dumps(dict(case=name, splits=sp, tile_ms=(s1["ms_tile"] - s0["ms_tile"]) / 3, count_ms=(s1["ms_count"] - s0["ms_count"]) / 3)), flush=True)
        e.close()

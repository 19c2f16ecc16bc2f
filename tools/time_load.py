import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens
from fastsk_amd import _native
for name in ["f7_cfg2_ep300_exact","f7_cfg3_ep47848_100combos","f7_cfg4_prot219_exact","f7_cfg1_prot11_approx_t1"]:
    d = load_golden(name)
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    e = _native.Engine(d["g"], d["m"])
    best=1e9
    for _ in range(20):
        t0=time.perf_counter(); e.load_sequences(tokens, offsets, ntr, nte); best=min(best,time.perf_counter()-t0)
    e.synchronize()
    print(name, "tokens", len(tokens), "load %.3f ms" % (best*1e3))
    e.close()

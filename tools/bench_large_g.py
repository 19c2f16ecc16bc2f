#!/usr/bin/env python3
"""The reference paper's "large feature length" regime: EP300 DNA (2000+2000 x 100 bp), k = g-m = 6
kept positions, g up to 20 (C(20,14) = 38,760 combos, 4^6 = 4096 keys). Times the exact kernel on
one GPU and checks a random subset of combos against the oracle (sub-block of 600 sequences).
    tools/bench_large_g.py [g,m ...]        (default: 12,6 16,10 20,14)
Prints one JSON line per setting; imported by bench.py for `also.large_g`."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fastsk_amd import _native

FORMS = {0: "bands", 1: "direct", 2: "blocks"}


def run(g, m, check=True):
    from conftest import load_tokens
    tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
    N = ntr + nte
    e = _native.Engine(g, m, profile=2)  # (2: the product dataflow, its kernels' times from events harvested by stats())
    nc = e.lib.num_combos(g, m)
    t0 = time.perf_counter()
    e.compute(tokens, offsets, ntr, nte)
    dt_first = time.perf_counter() - t0   # includes the device allocations
    t0 = time.perf_counter()
    e.compute(tokens, offsets, ntr, nte)
    dt = time.perf_counter() - t0
    st = e.stats()
    ok = None
    if check:  # parity on a subset: 12 random combos, first 600 sequences, against the oracle (the checker: never the thing timed)
        from oracle import loader
        rng = np.random.default_rng(g)
        sub = np.sort(rng.choice(nc, size=12, replace=False)).astype(np.int32)
        e2 = _native.Engine(g, m)
        e2.load_sequences(tokens[:offsets[600]], offsets[:601], 600, 0)
        e2.accumulate(sub); e2.finalize()
        want, _, _ = loader.port().raw_counts(tokens[:offsets[600]], offsets[:601], g, m, sub, threads=os.cpu_count())
        ok = bool(np.array_equal(e2.get_counts(), want))
        e2.close()
    tr = e.get_block(0, 4, 0, 4)
    # algorithmic bytes of the direct-atomic dataflow = 16 U + 16 P nfeat + packed input, per combo (two calls were counted)
    U = st["cell_updates"] / 2
    P = (max(1, int(np.ceil(np.log2(max(2, st["key_space"]))))) + 7) // 8
    alg = 16.0 * U + nc * (16.0 * P * st["n_feat"] + st["n_feat"] * st["bits_per_symbol"] / 8.0)
    dg = e.counts_digest()
    out = dict(g=g, m=m, combos=nc, N=N, seconds=dt, first_call_seconds=dt_first, combos_per_s=nc / dt, path="dense" if st["path_used"] == 1 else "sparse",
               U=int(U), U_per_combo=int(U / nc), pairs_per_record=round(U / (nc * st["n_feat"]), 2), algorithmic_GB=round(alg / 1e9, 1),
               algorithmic_GBs=round(alg / 1e9 / dt, 1), frac_of_hbm_peak=round(alg / 1e9 / dt / 8000.0, 3),
               sparse_form=FORMS.get(int(st["sparse_form"]), "-"), descriptors=bool(st["sparse_desc"]),
               share_positions_last_batch=int(st["share_positions"]), share_groups_last_batch=int(st["share_groups"]),
               ms={k: round(st[k] / 2, 1) for k in ("ms_extract", "ms_sort", "ms_segment", "ms_pairs", "ms_total")},
               digest=format(dg[0], "x") + "." + format(dg[1], "x"), subset_parity=ok,
               diag_ok=bool(np.all(np.diag(tr) == 1.0)))
    e.close()
    return out


if __name__ == "__main__":
    for g, m in [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or ((12, 6), (16, 10), (20, 14)):
        print(json.dumps(run(g, m)), flush=True)

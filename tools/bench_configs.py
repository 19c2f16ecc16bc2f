#!/usr/bin/env python3
"""Time BASELINE configs 1-4 (real FASTA inputs, committed as token fixtures) on one GPU through
the C ABI, print per-phase HIP-event times, and compare with the reference's CPU seconds recorded
in the fixtures when they were generated (tests/make_golden.py --full). Not the headline bench:
bench.py measures config 5."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, load_tokens, GOLD  # noqa: E402
from fastsk_amd import _native  # noqa: E402

CASES = ["f7_cfg1_prot11_approx_t1", "f7_cfg2_ep300_exact", "f7_cfg3_ep47848_100combos", "f7_cfg4_prot219_exact"]


def main():
    rows = []
    for name in CASES:
        if not os.path.exists(os.path.join(GOLD, name + ".npz")):
            continue
        d = load_golden(name)
        tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
        for profile in (False, True):
            e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"],
                               max_iters=d["max_iters"], skip_variance=bool(d["skip_variance"]), profile=profile)
            if d["approx"]:
                e.set_combo_order(d["order"])
            e.compute(tokens, offsets, ntr, nte)  # warm-up (allocations)
            dt = float("inf")
            for _ in range(1 if profile else 3):  # (best of three whole calls: a single call now and then catches a hiccup of the box)
                t0 = time.perf_counter()
                e.compute(tokens, offsets, ntr, nte)
                dt = min(dt, time.perf_counter() - t0)
            st = e.stats()
            e.close()
            if not profile:
                wall = dt
        # algorithmic bytes of the direct-atomic dataflow, 16*U + 16*P*nfeat + packed input per combo
        P = (max(1, int(np.ceil(np.log2(max(2, st["key_space"]))))) + 7) // 8  # 8-bit LSD passes over the packed k-mer
        useful = st["combos_done"] / st["combos_issued"] if st.get("combos_issued") else 1.0  # (variance mode drops what it ran ahead of its stop)
        alg = 16.0 * st["cell_updates"] * useful + st["combos_done"] * (16.0 * P * st["n_feat"] + st["n_feat"] * st["bits_per_symbol"] / 8.0)
        rows.append(dict(case=name, N=ntr + nte, combos=int(st["combos_done"]), gpu_seconds=wall,
                         algorithmic_GB=round(alg / 1e9, 2), algorithmic_GB_per_s=round(alg / 1e9 / wall, 1),
                         frac_of_8TBps=round(alg / 1e9 / wall / 8000.0, 3),
                         combos_per_s=st["combos_done"] / wall, ref_cpu_seconds=float(d["ref_seconds"]),
                         path="dense" if st["path_used"] == 1 else "sparse", U=int(st["cell_updates"]),
                         launches=st["launches"],
                         ms={k: round(st[k], 2) for k in ("ms_count", "ms_tile", "ms_extract", "ms_sort", "ms_segment",
                                                           "ms_pairs", "ms_total")}))
        print(json.dumps(rows[-1]))


if __name__ == "__main__":
    main()

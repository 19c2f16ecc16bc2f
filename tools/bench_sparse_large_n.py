#!/usr/bin/env python3
"""The sparse dataflow beyond the owner-band limit (N > ~23,000): protein-like inputs (sigma = 20, ragged L 60..220,
g = 10, m = 6) at N = 32k / 64k / 100k and DNA k = 8 (g = 12, m = 4, L = 300) at N = 32k, a bounded number of
combos each. Per workload: ms a combo, U (the reference's `+=` count, shared.cpp:316-320), SURVEY 8(d)'s
algorithmic bytes, the fraction of the 8 TB/s HBM roofline they make, the fraction of the guide's chip-wide
64-bit atomic ceiling (~1.3 TB/s of added bytes, 8 B an update), the dataflow that ran and which form of the
update stage it took (`sparse_form`: "bands" = LDS-summed owner bands, "blocks" = two-level 2-D blocks,
"direct" = one 64-bit atomic per +=).
    tools/bench_sparse_large_n.py [--quick] [--only NAME] [--combos n] [--tuning key=value,...]
Prints one JSON line per workload; imported by bench.py for `also.sparse_large_n`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsk_amd import _native  # noqa: E402

ATOMIC_CEILING_GBS = 1300.0   # MI355X_MICROARCH.md: chip-wide 64-bit atomic adds, bytes added a second
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (N, sigma, Lmin, Lmax, g, m)
    "protein_like_32k": (32000, 20, 60, 220, 10, 6),
    "protein_like_64k": (64000, 20, 60, 220, 10, 6),
    "protein_like_100k": (100000, 20, 60, 220, 10, 6),
    "dna_k8_32k": (32000, 4, 300, 300, 12, 4),
    # where the owner bands take several LDS rounds a band (the crossover to the two-level form): not part of the default list
    "protein_like_6k": (6000, 20, 60, 220, 10, 6),
    "protein_like_8k": (8000, 20, 60, 220, 10, 6),
    "protein_like_12k": (12000, 20, 60, 220, 10, 6),
    "protein_like_16k": (16000, 20, 60, 220, 10, 6),
    "protein_like_20k": (20000, 20, 60, 220, 10, 6),
    # long runs (descriptors) around the crossover
    "dna_k8_6k": (6000, 4, 300, 300, 12, 4),
    "dna_k8_8k": (8000, 4, 300, 300, 12, 4),
    "dna_k8_12k": (12000, 4, 300, 300, 12, 4),
    "dna_k8_16k": (16000, 4, 300, 300, 12, 4),
}
DEFAULT = ["protein_like_32k", "protein_like_64k", "protein_like_100k", "dna_k8_32k"]
FORMS = {0: "bands", 1: "direct", 2: "blocks"}


def make(name):
    N, sigma, lo, hi, g, m = WORKLOADS[name]
    rng = np.random.Generator(np.random.PCG64(20201214 + N + sigma))
    lens = rng.integers(lo, hi + 1, size=N, dtype=np.int64)
    offsets = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    tokens = rng.integers(1, sigma + 1, size=int(offsets[-1]), dtype=np.int32)
    return tokens, offsets, N, g, m


def run(name, n_combos=0, tuning=None, reps=2, lib=None):
    tokens, offsets, N, g, m = make(name)
    e = _native.Engine(g, m, path=_native.PATH_SPARSE, profile=2, tuning=tuning, lib=lib)  # (2: the product dataflow, events harvested by stats())
    nc = e.lib.num_combos(g, m)
    if not n_combos:
        n_combos = min(nc, 40)
    # evenly spread combos in natural order (an exact run walks them lexicographically; a prefix would share more positions)
    combos = np.unique(np.linspace(0, nc - 1, n_combos).astype(np.int32))
    e.load_sequences(tokens, offsets, N, 0)
    e.reset_counts()
    e.accumulate(combos)    # sizes every buffer
    e.synchronize()
    best = float("inf")
    for _ in range(reps):
        e.reset_counts()
        e.synchronize()
        t0 = time.perf_counter()
        e.accumulate(combos)
        e.synchronize()
        best = min(best, time.perf_counter() - t0)
    st = e.stats()
    form = FORMS.get(int(st["sparse_form"]), "?")
    per = best / len(combos)
    reps_total = reps + 1
    U = st["cell_updates"] / reps_total / len(combos)
    P = (max(1, int(np.ceil(np.log2(max(2, st["key_space"]))))) + 7) // 8
    alg = 16.0 * U + 16.0 * P * st["n_feat"] + st["n_feat"] * st["bits_per_symbol"] / 8.0
    dg = e.counts_digest()
    e.close()
    return dict(workload=name, N=N, sigma=WORKLOADS[name][1], g=g, m=m, combos_timed=int(len(combos)), combos_total=nc,
                n_feat=int(st["n_feat"]), key_space=int(st["key_space"]), ms_per_combo=round(per * 1e3, 3),
                U_per_combo=int(U), algorithmic_GB_per_combo=round(alg / 1e9, 3),
                algorithmic_GBs=round(alg / 1e9 / per, 1), frac_of_hbm_peak=round(alg / 1e9 / per / HBM_PEAK_GBS, 4),
                atomic_added_GBs=round(8.0 * U / 1e9 / per, 1), frac_of_atomic_ceiling=round(8.0 * U / 1e9 / per / ATOMIC_CEILING_GBS, 3),
                path_used="dense" if st["path_used"] == 1 else "sparse", sparse_form=form, descriptors=bool(st["sparse_desc"]),
                batches_redone=st["batches_redone"], passes=int(st["sparse_passes"]), full_kernel_seconds_estimate=round(per * nc, 2),
                ms={k: round(st[k] / reps_total, 2) for k in ("ms_extract", "ms_sort", "ms_segment", "ms_pairs", "ms_total")},
                digest=format(dg[0], "x") + "." + format(dg[1], "x"))


def main():
    names = list(DEFAULT)
    n_combos, tuning = 0, None
    args = sys.argv[1:]
    if "--quick" in args:
        names = ["protein_like_32k", "dna_k8_32k"]
    if "--only" in args:
        names = args[args.index("--only") + 1].split(",")
    if "--combos" in args:
        n_combos = int(args[args.index("--combos") + 1])
    if "--tuning" in args:
        tuning = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args[args.index("--tuning") + 1].split(",")}
    lib = _native.Library(args[args.index("--lib") + 1]) if "--lib" in args else None   # (a variant build: tools/build_variant.sh)
    for name in names:
        print(json.dumps(run(name, n_combos, tuning, lib=lib)), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Variance mode on the GPU against the CPU oracle, many random cases: chain count, max_iters, delta,
alphabet, ragged lengths; both dataflows; sparse batches sized exactly, enqueued ahead of their size, and
under a guard that overflows (redo). stdevs and the kernel must match bit for bit every time.
Not part of the pytest suite; run it after touching run_variance_mode or the Welford / sum kernels."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsk_amd import _native
from oracle import loader


def main(iters=200, seed=0):
    rng = np.random.default_rng(seed)
    port = loader.port()
    t0 = time.time()
    stops = set()
    for it in range(iters):
        sigma = int(rng.choice([3, 4, 5, 20]))
        k = int(rng.integers(2, 6 if sigma <= 5 else 4))
        m = int(rng.integers(1, 5))
        g = k + m
        N = int(rng.choice([1, 2, 3, 8, 40, 130, 300]))
        hi = int(rng.choice([g + 2, 30, 80]))
        X = [rng.integers(1, sigma + 1, size=int(L)).astype(np.int32) for L in rng.integers(g, max(hi, g) + 1, size=N)]
        if rng.random() < 0.3:
            X[int(rng.integers(N))][:] = 1  # a low-complexity row
        tokens, offsets = _native.flatten(X)
        ntr = int(rng.integers(1, N + 1))
        nc = port.num_combos(g, m)
        order = rng.permutation(nc).astype(np.int32)
        T = int(rng.choice([1, 1, 2, 3, 5, 25]))  # (25: more chains than some (g, m) have combos)
        max_iters = int(rng.choice([-1, 1, 2, 3, 5, 7, 12]))
        delta = float(rng.choice([0.01, 0.05, 0.2, 0.6, 2.0]))
        want, sd, _ = port.compute(tokens, offsets, ntr, N - ntr, g, m, t=T, approx=True, delta=delta, max_iters=max_iters, order=order)
        stops.add(len(sd))
        for path in (1, 2):
            if path == 1 and sigma ** k > 4096:
                continue
            mode, tuning = "default", {}
            if path == 2:
                mode = str(rng.choice(["default", "sync", "overflow", "ungrouped"]))
                if mode == "sync": tuning["sparse_sync"] = 1
                if mode == "overflow": tuning["guard_cap"] = int(rng.choice([1, 50, 2000]))
                if mode == "ungrouped": tuning["list_max_words"] = int(rng.choice([10, 1000]))
                # descriptors in the by-slot form of k_sx_consume (a slot's descriptors: a contiguous piece of every band's stream)
                tuning["sparse_desc"] = int(rng.choice([-1, 1, 1]))
                if tuning["sparse_desc"] == 1:
                    tuning["sparse_desc_min"] = int(rng.choice([1, 3, 16, 48]))
                    tuning["sparse_desc_cols"] = int(rng.integers(0, 4))
                    tuning["sparse_unpacked"] = int(rng.integers(0, 2))
            e = _native.Engine(g, m, t=T, approx=True, delta=delta, max_iters=max_iters, path=path, tuning=tuning)
            e.set_combo_order(order)
            e.compute(tokens, offsets, ntr, N - ntr)
            what = (it, path, mode, sigma, g, m, N, T, max_iters, delta)
            assert np.array_equal(e.get_stdevs(), sd), what
            assert np.array_equal(e.get_triangle(), want), what
            e.close()
        if it % 20 == 0:
            print("iter %d ok (%.0fs) sigma=%d g=%d m=%d N=%d T=%d max_iters=%d delta=%g stop=%d" % (it, time.time() - t0, sigma, g, m, N, T, max_iters, delta, len(sd)), flush=True)
    print("variance stress OK: %d random cases, %d distinct stopping iterations" % (iters, len(stops)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

#!/usr/bin/env python3
"""Race/edge screen on the GPU: many random shapes through both HIP dataflows (dense with and without
key compaction, sparse with LDS-owned and global pair accumulation), compared with each other every time and with the CPU oracle on a sample.
Not part of the pytest suite (minutes of GPU time); run it after touching a kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsk_amd import _native
from oracle import loader

def main(iters=150, seed=0):
    rng = np.random.default_rng(seed)
    port = loader.port()
    t0 = time.time()
    for it in range(iters):
        sigma = int(rng.choice([2, 3, 4, 4, 4, 5, 20]))
        k = int(rng.integers(1, 7 if sigma <= 5 else 4))
        m = int(rng.integers(0, 7))
        g = k + m
        N = int(rng.choice([1, 2, 63, 64, 65, 127, 129, 300, 700, 1500, 3000, 6500]))  # 6500: an owner band takes two LDS rounds
        lo = g
        hi = int(rng.choice([g, g + 1, 40, 120, 330, 700])) if N < 6000 else int(rng.choice([g + 1, 40, 90]))
        hi = max(hi, lo)
        lens = rng.integers(lo, hi + 1, size=N)
        X = [rng.integers(1, sigma + 1, size=int(L)).astype(np.int32) for L in lens]
        if N > 3 and rng.random() < 0.5:  # low-complexity rows: counts above 15 / 255
            X[int(rng.integers(N))][:] = 1
            j = int(rng.integers(N)); X[j][: len(X[j]) // 2] = 2
        if sigma >= 3 and N > 8 and rng.random() < 0.5:  # a rare symbol at a few places (the alphabet's last one, elsewhere remapped)
            for i in range(N):
                X[i][X[i] == sigma] = 1 + (i % (sigma - 1))
            for i in rng.choice(N, size=min(N, 5), replace=False):
                X[int(i)][int(rng.integers(len(X[int(i)])))] = sigma
        tokens, offsets = _native.flatten(X)
        nc = _native.library().num_combos(g, m)
        combos = np.unique(rng.integers(0, nc, size=int(rng.integers(1, 24)))).astype(np.int32)
        if rng.random() < 0.3:  # a long run of consecutive combos (with a few others): slots that share their leading positions
            first = int(rng.integers(0, nc))
            combos = np.concatenate([combos[:3], np.arange(first, min(nc, first + int(rng.integers(17, 60))), dtype=np.int32)]).astype(np.int32)
        ntr = int(rng.integers(1, N + 1))
        if os.environ.get("FSK_STRESS_VERBOSE"):  # (a crash loses the buffered line: say what is about to run)
            print("iter %d: sigma=%d g=%d m=%d N=%d Lmax=%d ntr=%d combos=%s" % (it, sigma, g, m, N, hi, ntr, combos.tolist()), flush=True)
        res = {}
        for name, path, env in (("dense", 1, "0"), ("dense_compact", 1, "1"), ("sparse", 2, "0"), ("sparse_global", 2, "1"), ("sparse_desc", 2, "0"),
                                ("sparse_blocks", 2, "0")):
            if path == 1 and (sigma ** k > 16384 or k > 16):
                continue
            if name == "dense_compact" and sigma ** k > 4096:
                continue
            tuning = {"compact": int(env) if path == 1 else 0, "sparse_global": int(env) if path == 2 else 0}
            if name == "sparse" and rng.random() < 0.3:
                tuning["guard_cap"] = int(rng.choice([1, 64, 5000]))
            # key compaction from the places of the rare symbols, forced on and off; a rare symbol planted in half of the cases;
            # the batches of an exact accumulate in two lanes, with batches of a few combos; the unpacked entry format
            if name == "dense_compact":
                tuning["compact_rare"] = int(rng.integers(0, 2))
            # descriptors (an entry of more than desc_min partners as one descriptor), forced or never: the owner bands with
            # several parts a band, the two-level blocks with small blocks (several passes, bands, sub-bands), every partner source
            if name in ("sparse", "sparse_global"):
                tuning["sparse_desc"] = -1
            if name in ("sparse_desc", "sparse_blocks"):
                tuning["sparse_desc"] = 1 if name == "sparse_desc" else int(rng.choice([-1, 1]))
                tuning["sparse_desc_min"] = int(rng.choice([1, 2, 5, 16, 48]))
                tuning["sparse_desc_cols"] = int(rng.integers(0, 4))
                if rng.random() < 0.5:
                    tuning["sparse_pairs"] = 0
                if name == "sparse_desc":
                    tuning["sparse_form"] = 1
                    if rng.random() < 0.5:
                        tuning["sparse_parts_target"] = int(rng.choice([16, 64, 1000]))
                        tuning["sparse_desc_parts"] = int(rng.choice([1, 64, 4096]))
                else:
                    tuning["sparse_form"] = 2
                    if rng.random() < 0.7:
                        tuning["blocks_sub_shift"] = int(rng.integers(5, 13))
                        tuning["blocks_max_bands"] = int(rng.choice([2, 7, 64, 512]))
                        tuning["blocks_band_shift_max"] = int(rng.integers(max(6, tuning["blocks_sub_shift"]), 16))
                        if rng.random() < 0.4:
                            tuning["blocks_pass_words"] = int(rng.choice([5000, 100000]))
            if path == 2 and rng.random() < 0.5:
                tuning["sparse_exact_lanes"] = 2
                tuning["sparse_batch_records"] = max(1, int(rng.integers(1, 5)) * int(sum(max(0, int(L) - g + 1) for L in lens)))
            if path == 2 and rng.random() < 0.3:
                tuning["sparse_unpacked"] = 1
            if path == 2 and k >= 2 and rng.random() < 0.6:  # windows presorted by the first 1 .. k - 1 kept positions per group of slots
                tuning["sparse_share"] = int(rng.integers(1, k))
            if os.environ.get("FSK_STRESS_VERBOSE"):
                print("   %s %s" % (name, tuning), flush=True)
            e = _native.Engine(g, m, path=path, tuning=tuning)
            e.load_sequences(tokens, offsets, ntr, N - ntr)
            if rng.random() < 0.5 or N < 256:
                # (several calls: the sparse dataflow enqueues all batches but the first ahead of their size;
                # now and then under a guard so small that they overflow and are redone)
                for part in np.array_split(combos, int(rng.integers(1, 4))):
                    if len(part):
                        e.accumulate(part)
            else:
                edges = sorted({0, N, *(int(x) // 128 * 128 for x in rng.integers(0, N, size=2))})
                for a, b in zip(edges[:-1], edges[1:]):
                    e.accumulate_rows(combos, a, b)
            e.finalize()
            res[name] = (e.get_counts(), e.get_triangle())
            e.close()
        if N - ntr > 0:  # skip_test_block on the sparse dataflow: exactly the test x test cells off the diagonal stay zero
            e = _native.Engine(g, m, path=2, skip_test_block=True, tuning={"sparse_global": it % 2, "sparse_unpacked": (it // 2) % 2,
                                                                            "sparse_desc": 1 if (it // 4) % 2 else -1, "sparse_desc_min": 1 + it % 7,
                                                                            "sparse_form": (0, 2)[(it // 8) % 2]})
            e.load_sequences(tokens, offsets, ntr, N - ntr)
            e.accumulate(combos)
            e.finalize()
            got = e.get_counts()
            e.close()
            i, j = np.tril_indices(N)
            masked = res["sparse"][0].copy()
            masked[(j >= ntr) & (i != j)] = 0
            assert np.array_equal(got, masked), (it, "skip_test_block", sigma, g, m, N, hi, ntr)
        names = list(res)
        for n in names[1:]:
            assert np.array_equal(res[names[0]][0], res[n][0]), (it, "counts", names[0], n, sigma, g, m, N, hi)
            assert np.array_equal(res[names[0]][1], res[n][1]), (it, "tri", names[0], n, sigma, g, m, N, hi)
        if it % 5 == 0 and N <= 700:
            want, _, _ = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
            assert np.array_equal(res[names[0]][0], want), (it, "oracle", sigma, g, m, N, hi)
        if it % 25 == 0:
            print("iter %d ok (%.0fs) sigma=%d g=%d m=%d N=%d Lmax=%d paths=%s" % (it, time.time() - t0, sigma, g, m, N, hi, names), flush=True)
    print("stress OK: %d random cases, all dataflows identical" % iters)

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

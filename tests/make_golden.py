#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (run in the build container only).

Every expected value written here comes out of the reference's own code:
  * tokens      <- the reference's Python reader, imported from /root/reference/src/fastsk/utils.py
  * train/test  <- FastSK::compute_kernel / compute_train + getters      (oracle/_ref, ref_compute)
  * tri, stdevs <- KernelFunction::compute_kernel (whole normalised triangle) (ref_full_triangle)
  * counts      <- the reference's extractFeatures/cntsrtna/countAndUpdateTri replayed per combo
                   (ref_raw_counts), i.e. the uint32 partial kernels the API never exposes
  * order       <- std::shuffle(default_random_engine(seed)) as fastsk_kernel.cpp:29-38 does it,
                   with time(0) pinned to `seed`
Only data is stored (inputs + expected outputs); no reference source travels.

Usage:  python tests/make_golden.py [--full]     (--full adds the full-size BASELINE configs 1-4,
                                                  ~30 min of reference CPU time)
"""
import argparse
import hashlib
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loader  # noqa: E402

REF_DATA = "/root/reference/data"
GOLD = os.path.join(ROOT, "tests", "golden")


def ref_reader():
    spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/src/fastsk/utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.FastaUtility()


def read_pair(name):
    rd = ref_reader()
    Xtr, Ytr = rd.read_data(os.path.join(REF_DATA, name + ".train.fasta"))
    Xte, Yte = rd.read_data(os.path.join(REF_DATA, name + ".test.fasta"))
    return Xtr, Ytr, Xte, Yte


def tri_to_square(tri, N):
    """Row-major lower triangle (index i(i+1)/2+j) -> symmetric N x N."""
    full = np.zeros((N, N), dtype=tri.dtype)
    il = np.tril_indices(N)
    full[il] = tri
    full.T[il] = tri
    return full


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def combos_used(order, n_combos, t, approx, max_iters, skip_variance, n_iters_t0=None):
    """Which combos enter an integer-valued result (exact or skip-variance modes)."""
    T = 20 if t == -1 else t
    T = max(1, min(T, n_combos))
    if not approx:
        return np.sort(order)
    used = []
    for tid in range(T):
        sl = order[tid::T]
        if max_iters != -1:
            sl = sl[:max_iters]
        used.extend(sl.tolist())
    return np.array(used, dtype=np.int32)


def make_case(name, Xtr, Xte, g, m, t=-1, approx=False, delta=0.025, max_iters=-1,
              skip_variance=False, seed=777, store_counts=True, store_tri=True, extra=None):
    r = loader.ref()
    p = loader.port()
    n_train, n_test = len(Xtr), len(Xte)
    tokens, offsets = loader.flatten(list(Xtr) + list(Xte))
    nc = p.num_combos(g, m)
    order = r.shuffle_order(seed, nc)
    t0 = time.time()
    train, test, sd = r.compute(tokens, offsets, n_train, n_test, g, m, t, approx, delta, max_iters,
                                skip_variance, seed)
    tri, sd2 = r.full_triangle(tokens, offsets, n_train, n_test, g, m, t, approx, delta, max_iters,
                               skip_variance, seed)
    out = dict(tokens=tokens, offsets=offsets, n_train=n_train, n_test=n_test, g=g, m=m, t=t,
               approx=int(approx), delta=delta, max_iters=max_iters,
               skip_variance=int(skip_variance), seed=seed, order=order, train=train, test=test,
               stdevs=sd2, tri_sha256=sha(tri))
    integer_mode = (not approx) or skip_variance
    # train/test blocks must be slices of the triangle (end-to-end == engine level)
    N = n_train + n_test
    full = tri_to_square(tri, N)
    assert np.array_equal(full[:n_train, :n_train], train), name
    if n_test:
        assert np.array_equal(full[n_train:, :n_train], test), name
    assert len(sd) == len(sd2) and np.array_equal(sd, sd2), name
    if store_tri:
        out["tri"] = tri
    if integer_mode and store_counts:
        used = combos_used(order, nc, t, approx, max_iters, skip_variance)
        counts, _ = r.raw_counts(tokens, offsets, g, m, used, threads=8)
        out["combos"] = used
        out["counts"] = counts
        out["counts_sha256"] = sha(counts)
        # raw counts + the normalisation expression must reproduce the reference's triangle
        chk = p.normalise(counts.astype(np.float64), N)
        assert np.array_equal(chk, tri), name + ": counts do not normalise to the reference output"
    if extra:
        out.update(extra)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s N=%d+%d g=%d m=%d  %.1fs  -> %s (%.1f KB)" % (
        name, n_train, n_test, g, m, time.time() - t0, os.path.relpath(path, ROOT),
        os.path.getsize(path) / 1024))


def small_cases():
    # F1: the reference's only toy input, data/small.*, g=3 m=1
    Xtr, _, Xte, _ = read_pair("small")
    make_case("f1_small_g3m1", Xtr, Xte, 3, 1, t=1)
    # F2: docs demo (docs/2demo/fastDemo.ipynb) — token 0 appears in the data
    make_case("f2_docsdemo_g3m2", [[1, 0, 1, 0, 1], [1, 1, 1, 0, 1]],
              [[1, 1, 1, 1, 1], [1, 0, 1, 0, 1]], 3, 2, t=1)
    # F3: edge cases
    make_case("f3_m0", [[1, 2, 3, 1, 2, 3, 1], [1, 2, 3, 3, 2, 1, 1]], [[3, 2, 1, 1, 2, 3, 1]], 3, 0, t=2)
    make_case("f3_zero_overlap", [[1, 1, 1, 1, 1, 1], [2, 2, 2, 2, 2, 2]], [[1, 1, 1, 2, 2, 2]], 4, 1, t=1)
    make_case("f3_varlen_g4m2", [[1, 2, 3, 4, 1, 2, 3, 4, 1, 2], [1, 2, 3, 4]],
              [[4, 3, 2, 1, 4, 3, 2, 1, 1, 2, 3, 4, 4]], 4, 2, t=3)
    make_case("f3_train_only", [[1, 2, 1, 2, 1, 2, 2, 1], [2, 1, 2, 1, 1, 1, 2], [1, 1, 2, 2, 1, 1]], [],
              4, 2, t=1)
    make_case("f3_g_equals_len", [[1, 2, 3, 4], [1, 2, 3, 1], [2, 2, 3, 4]], [[1, 2, 3, 4]], 4, 2, t=1)
    rng = np.random.default_rng(5)
    # sparse token ids in range (dict_size covers them): ids {0,1,2,3,4,5,6,7}
    X = [rng.integers(1, 8, size=int(n)).tolist() for n in rng.integers(9, 40, size=12)]
    make_case("f3_ragged_sigma7_g6m3", X[:8], X[8:], 6, 3, t=4)
    # low-complexity sequences: long runs, multiplicities > 1
    X = [[1] * 30 + [2] * 5, [1] * 12 + [2, 1] * 8, [2] * 40, [1, 2] * 17, [1] * 9]
    make_case("f3_lowcomplexity_g5m2", X[:3], X[3:], 5, 2, t=2)


def slice_cases():
    Xtr, _, Xte, _ = read_pair("EP300")
    Xtr, Xte = Xtr[:60], Xte[:40]
    make_case("f4_ep300_exact", Xtr, Xte, 10, 6, t=4)
    make_case("f4_ep300_skipvar_T1", Xtr, Xte, 10, 6, t=1, approx=True, max_iters=17, skip_variance=True)
    make_case("f4_ep300_skipvar_T3", Xtr, Xte, 10, 6, t=3, approx=True, max_iters=17, skip_variance=True)
    make_case("f4_ep300_variance_T1", Xtr, Xte, 10, 6, t=1, approx=True, max_iters=17)
    make_case("f4_ep300_variance_T1_conv", Xtr, Xte, 10, 6, t=1, approx=True, delta=0.5)
    Xtr, _, Xte, _ = read_pair("1.1")
    Xtr, Xte = Xtr[:70], Xte[:30]
    make_case("f5_prot11_exact", Xtr, Xte, 10, 6, t=4)
    make_case("f5_prot11_variance_T1", Xtr, Xte, 10, 6, t=1, approx=True)
    make_case("f5_prot11_variance_T1_it9", Xtr[:40], Xte[:20], 10, 6, t=1, approx=True, max_iters=9)
    Xtr, _, Xte, _ = read_pair("2.19")
    Xtr, Xte = Xtr[:50], Xte[:30]
    make_case("f6_prot219_exact", Xtr, Xte, 14, 10, t=8)
    make_case("f6_prot219_skipvar16", Xtr, Xte, 14, 10, t=1, approx=True, max_iters=16, skip_variance=True)
    Xtr, _, Xte, _ = read_pair("EP300_47848")
    make_case("f6_ep47848_slice_exact", Xtr[:40], Xte[:24], 10, 6, t=4)


def wide_cases():
    """F8: inputs beyond the packed fast paths — more than 256 distinct tokens (cntsrtna's radix is the dictionary size,
    shared.cpp:156-191) and a k-mer space beyond 2^62 (protein, g=20 m=4: 24^16)."""
    rng = np.random.default_rng(8)
    X = [rng.integers(1, 301, size=int(n)).tolist() for n in rng.integers(30, 61, size=80)]
    for x in X[:40]:                      # shared motifs, so that the kernel is not all zeros off the diagonal
        a = int(rng.integers(0, len(x) - 12))
        x[a:a + 12] = X[0][5:17]
    make_case("f8_sigma300_g5m2", X[:50], X[50:], 5, 2, t=2)
    Xtr, _, Xte, _ = read_pair("1.1")
    Xtr = [x for x in Xtr if len(x) >= 20][:40]
    Xte = [x for x in Xte if len(x) >= 20][:20]
    make_case("f8_prot11_g20m4_skipvar12", Xtr, Xte, 20, 4, t=1, approx=True, max_iters=12, skip_variance=True)


def token_fixtures():
    """Token arrays of the four FASTA configs, as produced by the reference's Python reader."""
    for name in ["EP300", "EP300_47848", "1.1", "2.19", "small"]:
        Xtr, Ytr, Xte, Yte = read_pair(name)
        tokens, offsets = loader.flatten(Xtr + Xte)
        assert tokens.max() < 256
        path = os.path.join(GOLD, "tokens_%s.npz" % name)
        np.savez_compressed(path, tokens=tokens.astype(np.uint8), offsets=offsets,
                            n_train=len(Xtr), n_test=len(Xte),
                            y_train=np.array(Ytr, dtype=np.int8), y_test=np.array(Yte, dtype=np.int8))
        print("tokens %-14s N=%d+%d  %.1f KB" % (name, len(Xtr), len(Xte), os.path.getsize(path) / 1024))


def full_case(name, data, g, m, t, approx=False, max_iters=-1, skip_variance=False, seed=777,
              ref_threads=None):
    """Full-size BASELINE config: digests + sampled cells only (inputs live in tokens_<data>.npz)."""
    r = loader.ref()
    p = loader.port()
    Xtr, _, Xte, _ = read_pair(data)
    n_train, n_test = len(Xtr), len(Xte)
    N = n_train + n_test
    tokens, offsets = loader.flatten(Xtr + Xte)
    nc = p.num_combos(g, m)
    order = r.shuffle_order(seed, nc)
    t0 = time.time()
    tri, sd = r.full_triangle(tokens, offsets, n_train, n_test, g, m,
                              t if ref_threads is None else ref_threads, approx, 0.025, max_iters,
                              skip_variance, seed)
    t_ref = time.time() - t0
    out = dict(data=data, n_train=n_train, n_test=n_test, g=g, m=m, t=t, approx=int(approx),
               delta=0.025, max_iters=max_iters, skip_variance=int(skip_variance), seed=seed,
               order=order, stdevs=sd, tri_sha256=sha(tri), ref_seconds=t_ref)
    rng = np.random.default_rng(1)
    cells = np.sort(rng.choice(N * (N + 1) // 2, size=4096, replace=False))
    out["sample_cells"] = cells
    out["sample_tri"] = tri[cells]
    # diagonal + first and last rows help localise a mismatch
    out["row_last"] = tri[N * (N - 1) // 2:]
    if (not approx) or skip_variance:
        used = combos_used(order, nc, t, approx, max_iters, skip_variance)
        counts, secs = r.raw_counts(tokens, offsets, g, m, used, threads=8)
        chk = p.normalise(counts.astype(np.float64), N)
        assert np.array_equal(chk, tri), name
        out.update(combos=used, counts_sha256=sha(counts), sample_counts=counts[cells],
                   diag_counts=counts[np.arange(N) * (np.arange(N) + 1) // 2 + np.arange(N)])
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s N=%d+%d g=%d m=%d  ref %.1fs -> %.1f KB" % (
        name, n_train, n_test, g, m, t_ref, os.path.getsize(path) / 1024))


def full_cases():
    # config 2: EP300 exact (result independent of thread count: run the reference with 8)
    full_case("f7_cfg2_ep300_exact", "EP300", 10, 6, t=8)
    # config 4: protein 2.19 exact, 1001 combos
    full_case("f7_cfg4_prot219_exact", "2.19", 14, 10, t=8)
    # config 1: protein 1.1 approx (variance mode), t=1, pinned seed
    full_case("f7_cfg1_prot11_approx_t1", "1.1", 10, 6, t=1, approx=True)
    # config 3: EP300_47848, fixed 100-combo sample = approx+skip_variance, max_iters=100, t=1
    full_case("f7_cfg3_ep47848_100combos", "EP300_47848", 10, 6, t=1, approx=True, max_iters=100,
              skip_variance=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only-full", action="store_true")
    ap.add_argument("--only-wide", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if args.only_wide:
        wide_cases()
        sys.exit(0)
    if not args.only_full:
        token_fixtures()
        small_cases()
        slice_cases()
        wide_cases()
    if args.full or args.only_full:
        full_cases()

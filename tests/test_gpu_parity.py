"""GPU parity tests proper: the HIP path, called through the C ABI (include/fastsk_amd.h), against
the golden vectors from the compiled reference, against the oracle on seeded inputs, and — at the
BASELINE's full sizes — through the sub-block property and structural invariants.
Bar: bit-exact (integer counts; fp64 normalised kernel via IEEE mul/sqrt/div in the same order).
Nothing here reads /root/reference."""
import hashlib
import os

import numpy as np
import pytest

from conftest import golden_names, load_golden, load_tokens, tri_to_square, synthetic_dna, set_tuning_env, GOLD, EDGE_CASES

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def native():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import __graft_entry__ as ge
    ge.build_engine()    # no-op when fastsk_amd/lib/libfastsk_amd.so is current
    ge.build_bindings()
    from fastsk_amd import _native
    lib = _native.library()  # raises if the HIP library is missing: no fallback
    assert lib.device_count() >= 1
    return _native


def hooks_library(native):
    """tests/hooks/libfastsk_amd_hooks.so: the engine compiled with -DFSK_TEST_HOOKS (fault injection); the product has none."""
    import __graft_entry__ as ge
    return native.Library(ge.build_engine(hooks=True))


def engine_for(native, d, path=0, **kw):
    e = native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"],
                      max_iters=d["max_iters"], skip_variance=bool(d["skip_variance"]), path=path, **kw)
    if d["approx"]:
        e.set_combo_order(d["order"])
    return e


def dense_ok(d):
    sigma = len(np.unique(d["tokens"]))
    return sigma ** (d["g"] - d["m"]) <= 1024


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("name", golden_names())
def test_golden_vectors(native, name, path):
    d = load_golden(name)
    if path == 1 and not dense_ok(d):
        pytest.skip("key space too large for the dense dataflow")
    e = engine_for(native, d, path)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert e.stats()["path_used"] == path
    tri = e.get_triangle()
    assert sha(tri) == d["tri_sha256"]
    assert np.array_equal(tri, d["tri"])
    assert np.array_equal(e.get_train(), d["train"])
    if d["n_test"]:
        assert np.array_equal(e.get_test(), d["test"])
    assert np.array_equal(e.get_stdevs(), d["stdevs"])
    if "counts" in d:
        assert np.array_equal(e.get_counts(), d["counts"])
    e.close()


def full_cases():
    return [n for n in ("f7_cfg2_ep300_exact", "f7_cfg4_prot219_exact", "f7_cfg1_prot11_approx_t1",
                        "f7_cfg3_ep47848_100combos") if os.path.exists(os.path.join(GOLD, n + ".npz"))]


@pytest.mark.parametrize("name", full_cases())
def test_baseline_configs_full_size(native, name):
    """BASELINE configs 1-4 at full size: digests of the reference's output."""
    d = load_golden(name)
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    e = engine_for(native, d)
    e.compute(tokens, offsets, ntr, nte)
    N = ntr + nte
    if "counts_sha256" in d:
        counts = e.get_counts()
        assert np.array_equal(counts[d["sample_cells"]], d["sample_counts"])
        assert sha(counts) == d["counts_sha256"]
    tri = e.get_triangle()
    assert np.array_equal(tri[d["sample_cells"]], d["sample_tri"])
    assert np.array_equal(tri[N * (N - 1) // 2:], d["row_last"])
    assert sha(tri) == d["tri_sha256"]
    assert np.array_equal(e.get_stdevs(), d["stdevs"])
    e.close()


def test_dense_equals_sparse_on_seeded_dna(native, port):
    """Two independent HIP dataflows and the oracle agree on config-5-shaped data (smaller N)."""
    tokens, offsets = synthetic_dna(700, 300)
    combos = np.arange(0, 495, 33, dtype=np.int32)
    out = []
    for path in (1, 2):
        e = native.Engine(12, 8, path=path, profile=True)  # (profile: the dense dataflow counts U as well)
        e.load_sequences(tokens, offsets, 700, 0)
        e.accumulate(combos)
        e.finalize()
        out.append((e.get_counts(), e.get_triangle(), e.stats()))
        e.close()
    want, _, U = port.raw_counts(tokens, offsets, 12, 8, combos, threads=8)
    assert np.array_equal(out[0][0], want) and np.array_equal(out[1][0], want)
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][1], port.normalise(want.astype(np.float64), 700))
    assert out[1][2]["cell_updates"] == U
    # the dense dataflow's U (k_dense_distinct over the count panels) is what bench.py's headline line builds
    # cell_updates_per_launch / useful_update_frac / algorithmic bytes from: the same exact count
    assert out[0][2]["path_used"] == 1 and out[0][2]["cell_updates"] == U
    # ... also over several accumulate calls, a repeated combo list (the cached-U shortcut) and key compaction
    e = native.Engine(12, 8, path=1, profile=True)
    e.load_sequences(tokens, offsets, 700, 0)
    e.accumulate(combos[:7]); e.accumulate(combos[7:]); e.accumulate(combos[7:])
    _, _, U2 = port.raw_counts(tokens, offsets, 12, 8, combos[7:], threads=8)
    assert e.stats()["cell_updates"] == U + U2
    e.close()


@pytest.mark.parametrize("sigma,g,m,n,lo,hi", [(4, 8, 4, 257, 8, 90), (5, 10, 6, 130, 10, 400), (20, 7, 3, 300, 7, 120),
                                               (24, 14, 10, 200, 16, 300), (3, 6, 5, 64, 6, 9), (2, 9, 3, 129, 9, 64)])
def test_ragged_random_inputs_vs_oracle(native, port, sigma, g, m, n, lo, hi):
    rng = np.random.default_rng(sigma * 1000 + g)
    X = [rng.integers(1, sigma + 1, size=int(L)).astype(np.int32) for L in rng.integers(lo, hi + 1, size=n)]
    tokens, offsets = native.flatten(X)
    ntr = n - n // 3
    want, _, _ = port.compute(tokens, offsets, ntr, n - ntr, g, m, t=1)
    for path in (1, 2):
        if path == 1 and sigma ** (g - m) > 1024:
            continue
        e = native.Engine(g, m, path=path)
        e.compute(tokens, offsets, ntr, n - ntr)
        assert np.array_equal(e.get_triangle(), want), "path %d" % path
        e.close()


def test_long_sequences_chunked_staging(native, port):
    """DNA sequences longer than one LDS staging pass (~1800 symbols at 256 keys)."""
    rng = np.random.default_rng(21)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(40, 5000, size=96)]
    tokens, offsets = native.flatten(X)
    combos = np.array([0, 100, 209], dtype=np.int32)
    want, _, _ = port.raw_counts(tokens, offsets, 10, 6, combos, threads=8)
    for path in (1, 2):
        e = native.Engine(10, 6, path=path)
        e.load_sequences(tokens, offsets, 96, 0)
        e.accumulate(combos)
        e.finalize()
        assert np.array_equal(e.get_counts(), want), "path %d" % path
        e.close()


@pytest.mark.parametrize("sigma,g,m,L", [(4, 10, 4, 100), (4, 11, 4, 300), (5, 8, 3, 150), (4, 9, 4, 60)])
def test_mid_size_key_spaces_dense_vs_sparse(native, port, sigma, g, m, L):
    """k = 5..7 (1024..16384 keys): count panels built in several LDS sweeps; both dataflows and
    the oracle agree; auto picks one of them by its cost model."""
    rng = np.random.default_rng(g * 100 + m)
    N = 500
    X = rng.integers(1, sigma + 1, size=(N, L), dtype=np.int32)
    X[7, :] = 2
    tokens, offsets = native.flatten(X)
    nc = port.num_combos(g, m)
    combos = np.unique(np.linspace(0, nc - 1, 9).astype(np.int32))
    want, _, _ = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    for path in (0, 1, 2):
        e = native.Engine(g, m, path=path)
        e.load_sequences(tokens, offsets, N, 0)
        e.accumulate(combos)
        e.finalize()
        assert np.array_equal(e.get_counts(), want), "path %d" % path
        e.close()


def test_wide_kmers_128_bit_sort_records(native, port):
    """k = 14 over 20 symbols (61 k-mer bits) and enough sequences that k-mer + sequence id pass 64 bits:
    the sparse dataflow sorts 128-bit records (and 64-bit ones for k = 9) — against the oracle."""
    rng = np.random.default_rng(31)
    for g, m, n in ((16, 2, 40), (12, 3, 300), (18, 4, 40)):  # (g = 18 x 8 bits: beyond the 128-bit window array)
        X = [rng.integers(1, 21, size=int(L)).astype(np.int32) for L in rng.integers(g + 1, 60, size=n)]
        for x in X[: n // 3]:
            x[1:g + 1] = X[n // 2][1:g + 1]
        for i in range(0, n // 3, 3):
            X[i][4] = X[i][4] % 20 + 1
        tokens, offsets = native.flatten(X)
        ntr = n - n // 4
        want, _, _ = port.compute(tokens, offsets, ntr, n - ntr, g, m, t=1)
        e = native.Engine(g, m)
        e.compute(tokens, offsets, ntr, n - ntr)
        assert e.stats()["path_used"] == 2 and e.stats()["key_space"] == 20 ** (g - m)
        assert np.array_equal(e.get_triangle(), want)
        e.close()


def test_low_complexity_counts_above_255(native, port):
    """A k-mer occurring > 255 times in one sequence does not fit the u8 count panels: the dense
    dataflow must notice and hand that batch to the general one."""
    X = [[1] * 400, [1] * 300 + [2] * 50, [2, 1] * 150, [1, 1, 2] * 100, [2] * 270]
    tokens, offsets = native.flatten(X)
    want, _, _ = port.compute(tokens, offsets, 3, 2, 6, 2, t=1)
    for path in (0, 1, 2):
        e = native.Engine(6, 2, path=path)
        e.compute(tokens, offsets, 3, 2)
        assert np.array_equal(e.get_triangle(), want)
        e.close()


@pytest.mark.parametrize("global_pairs", ["0", "1"])
def test_sparse_products_beyond_one_update_word(native, port, monkeypatch, global_pairs):
    """Sparse dataflow with multiplicities so large that a product does not fit the product field of a
    32-bit update word (a 1500-long homopolymer and a long dinucleotide repeat among 300 ordinary
    sequences: multiplicity x max windows ~ 2.2e6 > 2^18): such entries spend several words per pair."""
    set_tuning_env(monkeypatch, sparse_global=global_pairs)
    rng = np.random.default_rng(8)
    X = [rng.integers(1, 6, size=int(L)).astype(np.int32) for L in rng.integers(12, 90, size=300)]
    X[17] = np.full(1500, 3, dtype=np.int32)
    X[201] = np.array([1, 2] * 600, dtype=np.int32)
    X[202] = np.concatenate([np.full(700, 3, dtype=np.int32), rng.integers(1, 6, size=40).astype(np.int32)])
    tokens, offsets = native.flatten(X)
    g, m = 10, 6
    combos = np.arange(0, 210, 10, dtype=np.int32)
    want, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    for ntr in (300, 180):
        e = native.Engine(g, m, path=2)
        e.load_sequences(tokens, offsets, ntr, 300 - ntr)
        e.accumulate(combos)
        e.finalize()
        assert np.array_equal(e.get_counts(), want)
        assert e.stats()["cell_updates"] == U
        e.close()


def diag_by_definition(X, rows, g, m):
    """K_ii = sum over combos and k-mers v of cnt_i(combo, v)^2, straight from the definition
    (SURVEY section 0), for the given rows of a fixed-length token matrix with tokens 1..4."""
    import itertools
    S = (X[rows] - 1).astype(np.int64)
    n, L = S.shape
    W, k = L - g + 1, g - m
    out = np.zeros(n, dtype=np.uint64)
    base = np.arange(n, dtype=np.int64)[:, None] * (4 ** k)
    for pos in itertools.combinations(range(g), k):
        key = np.zeros((n, W), dtype=np.int64)
        for p in pos:
            key = key * 4 + S[:, p:p + W]
        cnt = np.bincount((key + base).ravel(), minlength=n * 4 ** k).reshape(n, 4 ** k)
        out += (cnt.astype(np.uint64) ** 2).sum(axis=1)
    return out


def check_random_subset(native, port, e, X, g, m, n_sub, seed, threads):
    """Sub-block property on a RANDOM subset of sequences: the oracle run on just those sequences
    must equal, cell for cell, the corresponding scattered cells of the big triangle — touches
    essentially every tile row and column."""
    N = X.shape[0]
    rng = np.random.Generator(np.random.PCG64(seed))
    idx = np.sort(rng.choice(N, size=n_sub, replace=False))
    st, so = native.flatten(X[idx])
    ncomb = native.library().num_combos(g, m)
    want, _, _ = port.raw_counts(st, so, g, m, np.arange(ncomb), threads=threads)
    a, b = np.tril_indices(n_sub)
    got = e.get_counts_cells(idx[a], idx[b])
    assert np.array_equal(got, want)
    # the same cells asked for transposed (column > row) are the same cells
    assert np.array_equal(e.get_counts_cells(idx[b][:5000], idx[a][:5000]), want[:5000])
    return idx


def test_config5_shape_mid_size_sub_blocks(native, port):
    """N = 16384 x 300 DNA, g=12 m=8, all 495 combos: sub-blocks and a random 1200-sequence subset
    against the oracle run on just those sequences (sub-block property), diagonals from the
    definition, plus structural invariants."""
    N, L, g, m = 16384, 300, 12, 8
    tokens, offsets = synthetic_dna(N, L)
    e = native.Engine(g, m)
    e.load_sequences(tokens, offsets, N, 0)
    e.accumulate(np.arange(495, dtype=np.int32))
    e.finalize()
    assert e.stats()["path_used"] == 1
    X = tokens.reshape(N, L)
    threads = min(32, os.cpu_count() or 8)
    for (a0, a1), (b0, b1) in [((0, 96), (0, 96)), ((16300, 16384), (37, 101)), ((8190, 8260), (8100, 8200))]:
        idx = np.concatenate([np.arange(b0, b1), np.arange(a0, a1)])
        idx = np.unique(idx)
        st, so = native.flatten(X[idx])
        want, _, _ = port.raw_counts(st, so, g, m, np.arange(495), threads=8)
        sq = tri_to_square(want, len(idx))
        ra = np.searchsorted(idx, np.arange(a0, a1))
        rb = np.searchsorted(idx, np.arange(b0, b1))
        got = e.get_counts_block(a0, a1, b0, b1)
        assert np.array_equal(got, sq[np.ix_(ra, rb)])
        dg = np.array([sq[i, i] for i in range(len(idx))], dtype=np.float64)
        wantn = sq.astype(np.float64) / np.sqrt(dg[:, None] * dg[None, :])
        gotn = e.get_block(a0, a1, b0, b1)
        off = ra[:, None] != rb[None, :]
        assert np.array_equal(gotn[off], wantn[np.ix_(ra, rb)][off])
    check_random_subset(native, port, e, X, g, m, 1200, seed=16384, threads=threads)
    rows = np.sort(np.random.Generator(np.random.PCG64(7)).choice(N, size=2000, replace=False))
    assert np.array_equal(e.get_counts_cells(rows, rows), diag_by_definition(X, rows, g, m))
    diag = e.get_counts_block(5000, 5001, 5000, 5001)[0, 0]
    assert diag >= 495 * (L - g + 1)
    blk = e.get_block(100, 228, 100, 228)
    assert np.array_equal(blk, blk.T) and np.all(np.diag(blk) == 1.0)
    e.close()


def test_config5_full_size_100k(native, port):
    """BASELINE config 5 at full size (100k x 300, 495 combos) on one GPU, in BOTH forms the product
    runs it: one un-banded launch after fsk_reset_counts (the storing flush bench.py times) and the
    row-banded form of the multi-GPU path. The two 5*10^9-cell triangles must be identical on the
    device; a random 1500-sequence subset (1.1 M scattered cells over every tile row and column,
    including rows beyond 46,340 where the reference's int index overflows, shared.cpp:97-117) must
    equal the oracle run on just those sequences; 4000 random diagonals must equal the definition."""
    import torch
    free, total = torch.cuda.mem_get_info()
    if free < 100e9:
        pytest.skip("needs ~85 GB of free HBM (two 40 GB triangles)")
    N, L, g, m = 100000, 300, 12, 8
    tokens, offsets = synthetic_dna(N, L)
    X = tokens.reshape(N, L)
    from fastsk_amd.distributed import band_edges
    pairs = N * (N + 1) // 2
    K = torch.zeros(pairs, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    e = native.Engine(g, m)
    e.bind_counts(K.data_ptr(), pairs, keepalive=K)
    e.load_sequences(tokens, offsets, N, 0)
    combos = np.arange(495, dtype=np.int32)
    # ---- form 1: exactly bench.py's step
    e.reset_counts()
    e.accumulate(combos)
    e.synchronize()
    e.finalize()
    st = e.stats()
    assert st["count_launches"] == 1 and st["n_tile_launches"] == 1 and st["combos_done"] == 495
    threads = min(64, os.cpu_count() or 8)
    idx = check_random_subset(native, port, e, X, g, m, 1500, seed=100000, threads=threads)
    assert idx.max() > 46340 and idx.min() < 2000
    rows = np.sort(np.random.Generator(np.random.PCG64(11)).choice(N, size=4000, replace=False))
    assert np.array_equal(e.get_counts_cells(rows, rows), diag_by_definition(X, rows, g, m))
    blk = e.get_block(99900, 100000, 99900, 100000)
    assert np.array_equal(blk, blk.T) and np.all(np.diag(blk) == 1.0)
    K1 = K.clone()
    torch.cuda.synchronize()
    # ---- form 2: row bands (panels counted once, one launch per band)
    edges = band_edges(N, 4)
    e.reset_counts()
    for lo, hi in zip(edges[:-1], edges[1:]):
        e.accumulate_rows(combos, lo, hi)
    e.synchronize()
    e.finalize()
    st = e.stats()
    assert st["count_launches"] == 2 and st["n_tile_launches"] == 1 + len(edges) - 1 and st["combos_done"] == 495
    step = 1 << 29
    for c0 in range(0, pairs, step):
        assert bool(torch.equal(K[c0:c0 + step], K1[c0:c0 + step])), "banded and un-banded triangles differ in cells %d.." % c0
    # a contiguous block that straddles a band edge, against the oracle
    (a0, a1), (b0, b1) = (edges[1] - 40, edges[1] + 40), (edges[1] - 60, edges[1] + 10)
    sub = np.unique(np.concatenate([np.arange(b0, b1), np.arange(a0, a1)]))
    st_, so_ = native.flatten(X[sub])
    want, _, _ = port.raw_counts(st_, so_, g, m, np.arange(495), threads=8)
    sq = tri_to_square(want, len(sub))
    ra, rb = np.searchsorted(sub, np.arange(a0, a1)), np.searchsorted(sub, np.arange(b0, b1))
    assert np.array_equal(e.get_counts_block(a0, a1, b0, b1), sq[np.ix_(ra, rb)])
    e.close()


def committed_digest(key):
    import json
    d = json.load(open(os.path.join(os.path.dirname(GOLD), "..", "profiles", "k_digests.json")))[key]
    return int(d["sum"], 16), int(d["xor"], 16)


def oracle_on_rows(native, port, X, rows, g, m, threads):
    """Raw integer square matrix and normalised matrix of the oracle run on just X[rows] (sub-block property:
    cells of the big kernel — raw AND normalised, the diagonals being the sequences' own — are the cells of the
    kernel of any subset of the sequences)."""
    st, so = native.flatten(X[rows])
    ncomb = native.library().num_combos(g, m)
    want, _, _ = port.raw_counts(st, so, g, m, np.arange(ncomb), threads=threads)
    n = len(rows)
    sq = tri_to_square(want, n)
    dg = np.diag(sq).astype(np.float64)
    norm = sq.astype(np.float64) / np.sqrt(dg[:, None] * dg[None, :])
    # the reference normalises a diagonal cell as K_ii / sqrt(K_ii * K_ii) (fastsk_kernel.cpp:99-102)
    norm[np.arange(n), np.arange(n)] = dg / np.sqrt(dg * dg)
    return sq, norm


def test_config5_full_size_100k_through_the_class(native, port):
    """The north-star call itself at the north-star size: fastsk.FastSK(g=12, m=8).compute_train(X) and
    .compute_kernel(X[:90000], X[90000:]) on the 100000 x 300 array of the config-5 generator, through
    the pybind11 class and the one-call fsk_compute (bindings.cpp:23-31, fastsk.cpp:30-188) — nothing staged, no
    caller-bound buffer. The integer triangle's digest must be the committed single-GPU one; normalised and raw
    blocks of >= 1000 randomly placed sequences must equal the oracle run on just those sequences; a device-resident
    block straddles row 46,341, where the reference's int triangle index overflows (shared.cpp:97-117)."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 110e9:
        pytest.skip("needs ~100 GB of free HBM (40 GB integer triangle + 40 GB normalised triangle + panels)")
    from fastsk import FastSK
    N, L, g, m = 100000, 300, 12, 8
    tokens, _ = synthetic_dna(N, L)
    X = tokens.reshape(N, L)
    want = committed_digest("config5:n_seq=100000,seq_len=300,g=12,m=8,combos=495")
    threads = min(64, os.cpu_count() or 8)
    rng = np.random.Generator(np.random.PCG64(4100000))

    # ---- compute_train
    f = FastSK(g=g, m=m)
    f.compute_train(X)
    st = f.stats()
    assert st["path_used"] == "dense" and st["combos_done"] == 495 and st["n_seq"] == N
    assert f.counts_digest() == want
    d_train = f.counts_digest(0, 90000)
    d_rest = f.counts_digest(90000, N)
    assert ((d_train[0] + d_rest[0]) % (1 << 64), d_train[1] ^ d_rest[1]) == want  # digests of row ranges combine
    # two randomly placed runs of 520 sequences (one beyond row 46,340) = a 1040-sequence subset
    a0 = int(rng.integers(0, 40000)); b0 = int(rng.integers(50000, N - 520))
    rows = np.concatenate([np.arange(a0, a0 + 520), np.arange(b0, b0 + 520)])
    sq, norm = oracle_on_rows(native, port, X, rows, g, m, threads)
    for (r0, r1, q0, q1) in [(0, 520, 0, 520), (520, 1040, 0, 520), (520, 1040, 520, 1040), (0, 520, 520, 1040)]:
        i0, i1, j0, j1 = rows[r0], rows[r1 - 1] + 1, rows[q0], rows[q1 - 1] + 1
        assert np.array_equal(f.get_counts_block(i0, i1, j0, j1), sq[r0:r1, q0:q1])
        assert np.array_equal(f.get_block(i0, i1, j0, j1), norm[r0:r1, q0:q1])
    # scattered cells of a random 1000-sequence subset, raw and (from them) normalised
    idx = np.sort(rng.choice(N, size=1000, replace=False))
    sq2, norm2 = oracle_on_rows(native, port, X, idx, g, m, threads)
    a, b = np.tril_indices(1000)
    got = f.get_counts_cells(idx[a], idx[b])
    assert np.array_equal(got, sq2[a, b])
    dg = f.get_counts_cells(idx, idx).astype(np.float64)
    off = a != b
    assert np.array_equal(got[off].astype(np.float64) / np.sqrt(dg[a[off]] * dg[b[off]]), norm2[a[off], b[off]])
    # a device-resident block straddling row 46,341
    blk = torch.from_dlpack(f.get_block_dlpack(46300, 46400, 46290, 46420))
    assert blk.is_cuda and blk.dtype == torch.float64 and tuple(blk.shape) == (100, 130)
    sub = np.arange(46290, 46420)
    _, norm3 = oracle_on_rows(native, port, X, sub, g, m, 8)
    assert np.array_equal(blk.cpu().numpy(), norm3[10:110, :])
    # the reference's K itself: the whole normalised triangle, device resident (fastsk_kernel.cpp:96-103 on 5e9 cells)
    tri = torch.from_dlpack(f.get_triangle_dlpack())
    assert tri.is_cuda and tri.dtype == torch.float64 and tri.numel() == N * (N + 1) // 2
    rr = torch.arange(N, device="cuda", dtype=torch.int64)
    assert bool((tri[rr * (rr + 1) // 2 + rr] == 1.0).all())                       # every diagonal
    ii, jj = torch.from_numpy(idx[a]).cuda(), torch.from_numpy(idx[b]).cuda()
    assert np.array_equal(tri[ii * (ii + 1) // 2 + jj].cpu().numpy()[off], norm2[a[off], b[off]])
    r0 = 46341 * 46342 // 2
    assert np.array_equal(tri[r0 + 46290:r0 + 46342].cpu().numpy()[:-1], norm3[46341 - 46290, :46341 - 46290])
    del tri, blk, f
    torch.cuda.empty_cache()

    # ---- compute_kernel, SURVEY 8(d)'s 90k / 10k split; the test x test block lazy (skip_test_block="lazy")
    f = FastSK(g=g, m=m, skip_test_block="lazy")
    f.compute_kernel(X[:90000], X[90000:])
    st = f.stats()
    assert st["combos_done"] == 495 and not st["test_block_computed"]
    assert f.counts_digest(0, 90000) == d_train            # train x train: same cells as compute_train's
    assert not f.stats()["test_block_computed"]
    tb = int(rng.integers(90000, N - 300)); ta = int(rng.integers(0, 90000 - 700))
    rows = np.concatenate([np.arange(ta, ta + 700), np.arange(tb, tb + 300)])
    sq, norm = oracle_on_rows(native, port, X, rows, g, m, threads)
    assert np.array_equal(f.get_block(tb, tb + 300, ta, ta + 700), norm[700:, :700])      # test x train
    te = torch.from_dlpack(f.get_test_kernel_dlpack())
    assert tuple(te.shape) == (10000, 90000)
    assert np.array_equal(te[tb - 90000:tb - 90000 + 300, ta:ta + 700].cpu().numpy(), norm[700:, :700])
    del te
    assert not f.stats()["test_block_computed"]
    assert np.array_equal(f.get_block(tb, tb + 300, tb, tb + 300), norm[700:, 700:])      # test x test: computed now
    assert f.stats()["test_block_computed"]
    assert f.counts_digest() == want


def protein_like(N, lo, hi, seed, sigma=20):
    """Ragged sequences over `sigma` letters with a skewed composition and shared motifs, so that the
    k-mer runs have the short-with-a-heavy-tail length distribution of real protein sets."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p = 1.0 / np.arange(1, sigma + 1) ** 0.7
    p /= p.sum()
    lens = rng.integers(lo, hi, size=N)
    motifs = [rng.choice(sigma, size=24, p=p) + 1 for _ in range(40)]
    X = []
    for L in lens:
        x = rng.choice(sigma, size=int(L), p=p) + 1
        if rng.random() < 0.5:
            mtf = motifs[int(rng.integers(len(motifs)))]
            a = int(rng.integers(0, L - 24))
            x[a:a + 24] = mtf
        X.append(x.astype(np.int32))
    return X


@pytest.mark.parametrize("N,lo,hi,ncombo,n_sub", [(12000, 60, 220, 24, 700), (24000, 30, 60, 6, 900)])
def test_sparse_dataflow_large_n(native, port, N, lo, hi, ncombo, n_sub):
    """The sparse (sort -> segments -> pair updates) dataflow beyond N = 8192: at 12,000 sequences a
    band of K takes several LDS rounds over its update stream; at 24,000 no band fits and every pair
    goes to K with a 64-bit atomic. Random-subset sub-block property against the oracle + U."""
    g, m = 10, 6
    X = protein_like(N, lo, hi, seed=N)
    tokens, offsets = native.flatten(X)
    combos = np.arange(0, 210, 210 // ncombo, dtype=np.int32)[:ncombo]
    e = native.Engine(g, m, path=2, profile=True)
    e.load_sequences(tokens, offsets, N, 0)
    e.accumulate(combos)
    e.finalize()
    st = e.stats()
    assert st["path_used"] == 2 and st["combos_done"] == ncombo
    rng = np.random.Generator(np.random.PCG64(5))
    idx = np.sort(rng.choice(N, size=n_sub, replace=False))
    stoks, soff = native.flatten([X[i] for i in idx])
    want, _, _ = port.raw_counts(stoks, soff, g, m, combos, threads=min(32, os.cpu_count() or 8))
    a, b = np.tril_indices(n_sub)
    assert np.array_equal(e.get_counts_cells(idx[a], idx[b]), want)
    # U of the whole run from the definition: sum over (combo, k-mer) of d(d+1)/2, on a slice the oracle can do
    sub = native.Engine(g, m, path=2)
    sub.load_sequences(stoks, soff, n_sub, 0)
    sub.accumulate(combos)
    sub.finalize()
    _, _, U = port.raw_counts(stoks, soff, g, m, combos, threads=8)
    assert sub.stats()["cell_updates"] == U
    blk = e.get_block(N - 100, N, N - 100, N)
    assert np.array_equal(blk, blk.T) and np.all(np.diag(blk) == 1.0)
    e.close()
    sub.close()


def _wave_primitives_check(lib):
    """fsk_test_wave_ops (test builds only): wave_incl_max_i32 (six hand-written v_max_i32_dpp), wave_incl_sum_u32 / _u64,
    and_xnor (v_bitop3), sbfe1, mbcnt ranking — against their definitions, on random, negative and packed (entry << 10 | count)
    values. The CPU emulation swaps these for shuffle loops, so only this runs the inline asm against a reference."""
    import ctypes as C
    rng = np.random.Generator(np.random.PCG64(64))
    n = 64 * 512
    v = rng.integers(-2 ** 31, 2 ** 31, size=n, dtype=np.int64).astype(np.int32)
    v[:64 * 128] = ((rng.integers(0, 2048, size=64 * 128) << 10) | rng.integers(0, 1024, size=64 * 128)).astype(np.int32)  # what k_sx_seg_write scans
    v[64 * 128:64 * 160] = -1
    v[64 * 160:64 * 192] = rng.integers(-5, 5, size=64 * 32).astype(np.int32)
    aux = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    out = [np.zeros(n, dtype=t) for t in (np.int32, np.uint32, np.uint64, np.uint32, np.int32, np.uint32)]
    fn = lib.L.fsk_test_wave_ops
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 2 + [C.c_int32] + [C.c_void_p] * 6
    assert fn(v.ctypes.data, aux.ctypes.data, n, *[o.ctypes.data for o in out]) == 0
    V, A = v.reshape(-1, 64), aux.reshape(-1, 64)
    assert np.array_equal(out[0].reshape(-1, 64), np.maximum.accumulate(V, axis=1))
    assert np.array_equal(out[1].reshape(-1, 64), np.cumsum(V.astype(np.uint32) & 0xffff, axis=1, dtype=np.uint32))
    w64 = (A.astype(np.uint64) << np.uint64(20)) | (V.astype(np.uint32).astype(np.uint64) & np.uint64(0xfffff))
    assert np.array_equal(out[2].reshape(-1, 64), np.cumsum(w64, axis=1, dtype=np.uint64))
    t = (A.astype(np.uint64) * np.uint64(2654435761)).astype(np.uint32)
    assert np.array_equal(out[3].reshape(-1, 64), V.astype(np.uint32) & ~(A ^ t))
    assert np.array_equal(out[4].reshape(-1, 64), -(((V.astype(np.uint32) >> (A & 31)) & 1).astype(np.int32)))
    bit = (A & 1).astype(np.uint32)
    assert np.array_equal(out[5].reshape(-1, 64), np.cumsum(bit, axis=1, dtype=np.uint32) - bit)


def test_dense_split_launch_through_staging_blocks(native):
    """tuning dense_small=1: the small-N tile launch (several workgroups a tile) leaves 32-bit staging blocks that k_dense_widen
    adds into K — BASELINE configs 2 and 3 at full size against the committed digests of their integer triangles."""
    for name in ("f7_cfg2_ep300_exact", "f7_cfg3_ep47848_100combos"):
        d = load_golden(name)
        tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
        base = None
        for tun in ({"dense_small": 0}, {"dense_small": 1}, {"dense_small": 1, "tile_splits": 5}):
            e = native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), max_iters=d["max_iters"], skip_variance=bool(d["skip_variance"]),
                              path=1, tuning=tun)
            if d["approx"]:
                e.set_combo_order(d["order"])
            e.compute(tokens, offsets, ntr, nte)
            import hashlib
            assert hashlib.sha256(e.get_counts().tobytes()).hexdigest() == d["counts_sha256"], (name, tun)
            e.close()


def test_wave_primitives(native):
    _wave_primitives_check(hooks_library(native))


# ---- the two-level form of the update stage (fsk_sparse_blocks.inc): what countAndUpdateTri (shared.cpp:268-333) takes where the
# owner bands end — bands binned by k_sx_emit, every band's stream split by sub-band, one workgroup a sub-band
@pytest.mark.parametrize("skip,desc", [(False, -1), (False, 1), (True, 1)])
def test_sparse_two_level_blocks_100k_protein_like(native, port, skip, desc):
    """N = 100,000 protein-like ragged sequences (20 letters, 60-220 long), g=10 m=6, four combos: 5 * 10^9 cells — more than
    one pass's 32-bit cell offsets cover, several passes by word count too — through the blocks form. Random 1400-sequence subset
    (rows on both sides of 65,535) against the oracle, U on that subset, the digest of the whole triangle against the same combos
    added with one 64-bit atomic per += (tuning sparse_form=3); with skip_test_block. desc = 1: the entries of more than 16
    partners as descriptor records, one per sub-band their partners fall into (8-byte entries: rows beyond 65,535)."""
    N, g, m = 100000, 10, 6
    X = protein_like(N, 60, 221, seed=100)
    tokens, offsets = native.flatten(X)
    combos = np.array([0, 71, 140, 209], dtype=np.int32)
    n_train = 60000
    e = native.Engine(g, m, path=2, skip_test_block=skip, tuning={"sparse_desc": desc})
    e.load_sequences(tokens, offsets, n_train if skip else N, N - n_train if skip else 0)
    e.accumulate(combos)
    e.finalize()
    st = e.stats()
    assert st["path_used"] == 2 and st["sparse_form"] == 2 and st["sparse_passes"] >= 2 and st["n_seq"] == N and st["sparse_desc"] == (desc > 0)
    rng = np.random.Generator(np.random.PCG64(9))
    idx = np.sort(np.concatenate([rng.choice(65535, size=800, replace=False), 65535 + rng.choice(N - 65535, size=600, replace=False)]))
    stoks, soff, U = _subset_against_oracle(native, port, e, X, idx, g, m, combos, n_train if skip else None)
    dg = e.counts_digest()
    e.close()
    if not skip:
        sub = native.Engine(g, m, path=2, tuning={"sparse_form": 2, "sparse_unpacked": 1})
        sub.load_sequences(stoks, soff, len(idx), 0)
        sub.accumulate(combos)
        sub.finalize()
        assert sub.stats()["cell_updates"] == U and sub.stats()["sparse_form"] == 2
        sub.close()
    d = native.Engine(g, m, path=2, skip_test_block=skip, tuning={"sparse_form": 3})
    d.load_sequences(tokens, offsets, n_train if skip else N, N - n_train if skip else 0)
    d.accumulate(combos)
    d.finalize()
    assert d.stats()["sparse_form"] == 1 and d.counts_digest() == dg
    assert skip or d.stats()["cell_updates"] == st["cell_updates"]
    d.close()


def test_sparse_two_level_blocks_forced_small(native, port):
    """The same form forced on inputs the owner bands hold (tuning sparse_form=2), with small blocks so that one call takes
    many passes, bands and sub-bands: whole triangles against the oracle, the band form and the atomics, exact U; DNA runs of
    hundreds of entries (long entries, multiplicities) and protein-like ones; row bands; three calls."""
    for X, g, m, combos in ((protein_like(1500, 40, 160, seed=31), 10, 6, np.arange(0, 210, 9, dtype=np.int32)),
                            ([x for x in synthetic_dna(1200, 90, seed=5)[0].reshape(1200, 90)], 9, 4, np.array([0, 50, 125], dtype=np.int32))):
        N = len(X)
        tokens, offsets = native.flatten(X)
        want, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=min(32, os.cpu_count() or 8))
        for tun in ({"sparse_form": 1}, {"sparse_form": 3}, {"sparse_form": 2},
                    {"sparse_form": 2, "blocks_sub_shift": 8, "blocks_max_bands": 7, "blocks_band_shift_max": 12},
                    {"sparse_form": 2, "blocks_sub_shift": 10, "blocks_max_bands": 64, "blocks_band_shift_max": 13, "blocks_pass_words": 400000,
                     "sparse_unpacked": 1},
                    # (passes of ONE row with more bands than blocks_max_bands: rows beyond 256 sequences at two bands of 2^7 cells a pass)
                    {"sparse_form": 2, "blocks_sub_shift": 6, "blocks_max_bands": 2, "blocks_band_shift_max": 7, "sparse_desc": 1, "sparse_desc_min": 7},
                    {"sparse_form": 2, "sparse_desc": 1, "sparse_desc_min": 3},
                    {"sparse_form": 2, "blocks_sub_shift": 8, "blocks_max_bands": 7, "blocks_band_shift_max": 12, "sparse_desc": 1},
                    {"sparse_form": 2, "blocks_sub_shift": 10, "blocks_max_bands": 64, "blocks_band_shift_max": 13, "blocks_pass_words": 400000,
                     "sparse_unpacked": 1, "sparse_desc": 1, "sparse_desc_min": 5}):
            for how in ("whole", "three calls", "row bands"):
                e = native.Engine(g, m, path=2, tuning=tun)
                e.load_sequences(tokens, offsets, N, 0)
                if how == "whole":
                    e.accumulate(combos)
                elif how == "three calls":
                    for part in np.array_split(combos, 3):
                        e.accumulate(part)
                else:
                    for lo, hi in ((0, 512), (512, 1024), (1024, N)):
                        e.accumulate_rows(combos, lo, hi)
                e.finalize()
                st = e.stats()
                assert st["sparse_form"] == {1: 0, 2: 2, 3: 1}[tun["sparse_form"]]
                assert "blocks_sub_shift" not in tun or st["sparse_passes"] >= 4
                assert st["sparse_desc"] == (1 if tun.get("sparse_desc") == 1 else st["sparse_desc"])
                assert np.array_equal(e.get_counts(), want), (g, m, tun, how)
                assert st["cell_updates"] == U
                e.close()


# ---- the unpacked entry format (8 + 4 (+ 4) bytes) of the sparse dataflow: N >= 65,535 sequences, or a sequence of >= 65,536
# windows — k_sx_seg_write<RecT, false> and k_sx_emit<DIRECT | lists, SKIP, false>, the path of countAndUpdateTri
# (shared.cpp:268-333) for inputs whose ids or multiplicities do not fit 16 bits
def _subset_against_oracle(native, port, e, X, idx, g, m, combos, n_train=None):
    """The oracle on just the sequences `idx`: its cells must equal the scattered cells (idx[a], idx[b]) of the big triangle —
    all of them, or with skip_test_block every cell whose column is a train sequence or that lies on the diagonal."""
    stoks, soff = native.flatten([X[i] for i in idx])
    want, _, U = port.raw_counts(stoks, soff, g, m, combos, threads=min(32, os.cpu_count() or 8))
    a, b = np.tril_indices(len(idx))
    got = e.get_counts_cells(idx[a], idx[b])
    if n_train is None:
        assert np.array_equal(got, want)
    else:
        keep = (idx[b] < n_train) | (a == b)
        assert np.array_equal(got[keep], want[keep]) and not got[~keep].any() and want[~keep].any()
    return stoks, soff, U


@pytest.mark.parametrize("skip", [False, True])
def test_sparse_unpacked_entries_beyond_65535_sequences(native, port, skip):
    """N = 70,000 protein-like ragged sequences (20 letters, 30-50 long), g=8 m=4, five combos, sparse dataflow: sequence ids
    no longer fit the packed entry format. Random 1500-sequence subset (rows on both sides of 65,535) against the oracle,
    the update count U on that subset, with and without skip_test_block."""
    N, g, m = 70000, 8, 4
    X = protein_like(N, 30, 51, seed=70)
    tokens, offsets = native.flatten(X)
    combos = np.array([0, 17, 33, 51, 69], dtype=np.int32)
    n_train = 40000
    e = native.Engine(g, m, path=2, skip_test_block=skip)
    e.load_sequences(tokens, offsets, n_train if skip else N, N - n_train if skip else 0)
    e.accumulate(combos)
    e.finalize()
    st = e.stats()
    assert st["path_used"] == 2 and st["combos_done"] == len(combos) and st["n_seq"] == N
    rng = np.random.Generator(np.random.PCG64(5))
    idx = np.sort(np.concatenate([rng.choice(65535, size=900, replace=False), 65535 + rng.choice(N - 65535, size=600, replace=False)]))
    stoks, soff, U = _subset_against_oracle(native, port, e, X, idx, g, m, combos, n_train if skip else None)
    if not skip:   # U from the definition, on the subset, through the same (forced) entry format
        sub = native.Engine(g, m, path=2, tuning={"sparse_unpacked": 1})
        sub.load_sequences(stoks, soff, len(idx), 0)
        sub.accumulate(combos)
        sub.finalize()
        assert sub.stats()["cell_updates"] == U
        sub.close()
    blk = e.get_block(N - 64, N, N - 64, N)
    assert np.all(np.diag(blk) == 1.0) and (skip or np.array_equal(blk, blk.T))
    e.close()


@pytest.mark.parametrize("global_pairs", ["0", "1"])
@pytest.mark.parametrize("skip", [False, True])
def test_sparse_unpacked_entries_one_very_long_sequence(native, port, monkeypatch, global_pairs, skip):
    """One sequence of 66,000 windows among 300 ordinary ones (N < 65,535): multiplicities and ranks need the unpacked entries
    while the update STREAMS are in use (k_sx_emit<false, SKIP, false>); and as 64-bit atomics (tuning sparse_global=1)."""
    set_tuning_env(monkeypatch, sparse_global=global_pairs)
    rng = np.random.default_rng(66)
    g, m = 9, 4
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(40, 120, size=300)]
    X.insert(137, rng.integers(1, 5, size=66000 + g - 1).astype(np.int32))
    N, n_train = len(X), 200
    tokens, offsets = native.flatten(X)
    combos = np.array([3, 40, 77, 125], dtype=np.int32)
    raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    e = native.Engine(g, m, path=2, skip_test_block=skip)
    e.load_sequences(tokens, offsets, n_train if skip else N, N - n_train if skip else 0)
    e.accumulate(combos)
    e.finalize()
    st = e.stats()
    assert st["path_used"] == 2 and st["max_windows"] == 66000
    got = e.get_counts()
    if not skip:
        assert np.array_equal(got, raw) and st["cell_updates"] == U
    else:
        a, b = np.tril_indices(N)
        keep = (b < n_train) | (a == b)
        assert np.array_equal(got[keep], raw[keep]) and not got[~keep].any() and raw[~keep].any()
        assert st["cell_updates"] < U
    e.close()


@pytest.mark.parametrize("name", ["f5_prot11_exact", "f6_prot219_skipvar16", "f3_ragged_sigma7_g6m3"])
@pytest.mark.parametrize("global_pairs", ["0", "1"])
def test_sparse_unpacked_entries_forced_on_the_goldens(native, monkeypatch, name, global_pairs):
    """The same kernel instantiations on the golden vectors (tuning sparse_unpacked=1): streams and atomics, exact counts, and
    variance mode's by-slot batches on top of unpacked entries."""
    set_tuning_env(monkeypatch, sparse_unpacked="1", sparse_global=global_pairs)
    d = load_golden(name)
    e = engine_for(native, d, path=2)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_counts(), d["counts"]) and np.array_equal(e.get_triangle(), d["tri"])
    e.close()
    if global_pairs == "0" and name == "f5_prot11_exact":
        v = load_golden("f5_prot11_variance_T1_it9")
        e = engine_for(native, v, path=2)
        e.compute(v["tokens"], v["offsets"], v["n_train"], v["n_test"])
        assert np.array_equal(e.get_stdevs(), v["stdevs"]) and np.array_equal(e.get_triangle(), v["tri"])
        e.close()


@pytest.mark.parametrize("skip", [False, True])
def test_sparse_paired_unit_words(native, port, monkeypatch, skip):
    """Update streams with PAIRS (a band of fewer than 32767 cells): the words of UNIT entries — multiplicity 1, every partner
    of multiplicity 1 — leave k_sx_emit as bare 15-bit cells, two to a 32-bit container, when the tile's short entries are
    binned in one pass; protein-like data (one-pass tiles, most words unit) and long runs over 400 keys (several passes: no
    pairs), pairs on and off (tuning sparse_pairs), in one call, in three, in row bands: the oracle's counts and U."""
    for X, g, m, combos in ((protein_like(1500, 40, 160, seed=31), 10, 6, np.arange(0, 210, 9, dtype=np.int32)),
                            ([np.random.default_rng(12 + i).integers(1, 21, size=30).astype(np.int32) for i in range(1400)], 4, 2, np.arange(6, dtype=np.int32))):
        N, ntr = len(X), 900
        X[7][:] = 3   # a low-complexity sequence: multiplicities above 1 inside otherwise clean runs
        tokens, offsets = native.flatten(X)
        raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=min(32, os.cpu_count() or 8))
        a, b = np.tril_indices(N)
        keep = (b < ntr) | (a == b) if skip else np.ones(len(a), dtype=bool)
        for pairs in ("1", "0"):
            set_tuning_env(monkeypatch, sparse_pairs=pairs)
            for how in ("whole", "three calls", "row bands"):
                e = native.Engine(g, m, path=2, skip_test_block=skip)
                e.load_sequences(tokens, offsets, ntr if skip else N, N - ntr if skip else 0)
                if how == "whole":
                    e.accumulate(combos)
                elif how == "three calls":
                    for part in np.array_split(combos, 3):
                        e.accumulate(part)
                else:
                    for lo, hi in ((0, 384), (384, 1024), (1024, N)):
                        e.accumulate_rows(combos, lo, hi)
                e.finalize()
                got = e.get_counts()
                assert np.array_equal(got[keep], raw[keep]), (pairs, how, g)
                assert skip or e.stats()["cell_updates"] == U
                e.close()


@pytest.mark.parametrize("skip", [False, True])
def test_sparse_descriptors(native, port, monkeypatch, skip):
    """Descriptors (tuning sparse_desc=1): entries of more partners than k_sx_emit bins leave as ONE descriptor each, which
    k_sx_consume expands in LDS (countAndUpdateTri's += for a whole entry, shared.cpp:316-327). Long runs over 256 keys at
    N = 5000 (bands of two LDS rounds: every descriptor is walked twice, each round keeping its own cells) and protein-like
    data at N = 1500, thresholds 48 / 6 partners, a low-complexity sequence (own cells of multiplicities above 1), in one
    call, in three, in row bands, with skip_test_block: the oracle's counts and U."""
    for X, g, m, combos, desc_mins in (
            ([np.random.default_rng(40 + i).integers(1, 5, size=26).astype(np.int32) for i in range(5000)], 6, 2, np.arange(0, 15, 5, dtype=np.int32), ("48", "6")),
            (protein_like(1500, 40, 160, seed=33), 10, 6, np.arange(0, 210, 9, dtype=np.int32), ("6",))):
        N, ntr = len(X), (2 * len(X)) // 3
        X[7][:] = 3
        tokens, offsets = native.flatten(X)
        raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=min(32, os.cpu_count() or 8))
        a, b = np.tril_indices(N)
        keep = (b < ntr) | (a == b) if skip else np.ones(len(a), dtype=bool)
        for desc_min in desc_mins:
            set_tuning_env(monkeypatch, sparse_desc="1", sparse_desc_min=desc_min, sparse_form="1")
            for how in ("whole", "three calls", "row bands"):
                e = native.Engine(g, m, path=2, skip_test_block=skip)
                e.load_sequences(tokens, offsets, ntr if skip else N, N - ntr if skip else 0)
                if how == "whole":
                    e.accumulate(combos)
                elif how == "three calls":
                    for part in np.array_split(combos, 3):
                        e.accumulate(part)
                else:
                    for lo, hi in ((0, 128 * (N // 512)), (128 * (N // 512), 128 * (N // 192)), (128 * (N // 192), N)):  # (multiples of 128)
                        e.accumulate_rows(combos, lo, hi)
                e.finalize()
                got = e.get_counts()
                st = e.stats()
                assert st["sparse_desc"] == 1 and st["sparse_form"] == 0
                assert np.array_equal(got[keep], raw[keep]), (desc_min, how, g)
                assert skip or st["cell_updates"] == U
                e.close()


@pytest.mark.parametrize("name", ["f7_cfg4_prot219_exact", "f7_cfg1_prot11_approx_t1", "f5_prot11_variance_T1_it9", "f6_prot219_skipvar16"])
@pytest.mark.parametrize("tune", [{"sparse_desc": "1"}, {"sparse_desc": "1", "sparse_desc_min": "4", "sparse_unpacked": "1"}])
def test_sparse_descriptors_on_the_goldens(native, monkeypatch, name, tune):
    """Descriptors forced on BASELINE configs 4 and 1 at full size (digests of the reference's output) and on the golden
    slices: exact, skip-variance, variance mode (the by-slot form of k_sx_consume, u16 slot triangles), both entry formats."""
    if not os.path.exists(os.path.join(GOLD, name + ".npz")):
        pytest.skip("golden not present")
    set_tuning_env(monkeypatch, **tune)
    d = load_golden(name)
    if "data" in d and "tokens" not in d:
        tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    else:
        tokens, offsets, ntr, nte = d["tokens"], d["offsets"], d["n_train"], d["n_test"]
    e = engine_for(native, d, path=2)
    e.compute(tokens, offsets, ntr, nte)
    assert e.stats()["sparse_desc"] == 1
    tri = e.get_triangle()
    if "sample_cells" in d:
        assert np.array_equal(tri[d["sample_cells"]], d["sample_tri"]) and sha(tri) == d["tri_sha256"]
        if "counts_sha256" in d:
            assert sha(e.get_counts()) == d["counts_sha256"]
    else:
        assert np.array_equal(tri, d["tri"])
        if "counts" in d:
            assert np.array_equal(e.get_counts(), d["counts"])
    assert np.array_equal(e.get_stdevs(), d["stdevs"])
    e.close()


@pytest.mark.parametrize("share", ["0", "1", "2", "3", "-1"])
def test_sparse_shared_leading_positions_config4(native, monkeypatch, share):
    """Shared prefixes: BASELINE config 4 (1001 consecutive combos, k = 4 of 20 symbols — the batches are large enough for the
    product to share by itself: share 0) with the windows presorted by the first 1, 2, 3 kept positions per group of slots
    (4- and 8-byte presort records) and never: the reference's own triangle."""
    set_tuning_env(monkeypatch, sparse_share=share)
    d = load_golden("f7_cfg4_prot219_exact")
    tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
    e = native.Engine(d["g"], d["m"], path=2)
    e.compute(tokens, offsets, ntr, nte)
    counts = e.get_counts()
    assert np.array_equal(counts[d["sample_cells"]], d["sample_counts"]) and sha(counts) == d["counts_sha256"], share
    e.close()


def test_sparse_shared_leading_positions_lists(native, port, monkeypatch):
    """The same on arbitrary combo lists (non-consecutive ids, repeats, jumps back: groups of one slot), 64-bit records (300
    symbols), skip_test_block, split calls and row bands — the oracle's counts."""
    rng = np.random.default_rng(77)
    for sigma, g, m, N, L, shares in ((20, 7, 3, 900, 60, ("1", "2", "3")), (300, 7, 3, 500, 40, ("1", "3"))):
        X = [rng.integers(1, sigma + 1, size=int(n)).astype(np.int32) for n in rng.integers(g + 4, L, size=N)]
        X[3][:] = 2
        tokens, offsets = native.flatten(X)
        ntr = N * 2 // 3
        combos = np.concatenate([np.arange(35, dtype=np.int32), np.asarray([7, 7, 1, 30, 2, 34, 33], dtype=np.int32)])
        raw, _, _ = port.raw_counts(tokens, offsets, g, m, combos, threads=min(32, os.cpu_count() or 8))
        a, b = np.tril_indices(N)
        for skip in (False, True):
            keep = (b < ntr) | (a == b) if skip else np.ones(len(a), dtype=bool)
            for share in shares:
                set_tuning_env(monkeypatch, sparse_share=share)
                for how in ("whole", "two calls", "row bands"):
                    e = native.Engine(g, m, path=2, skip_test_block=skip)
                    e.load_sequences(tokens, offsets, ntr, N - ntr)
                    if how == "whole":
                        e.accumulate(combos)
                    elif how == "two calls":
                        e.accumulate(combos[:20]); e.accumulate(combos[20:])
                    else:
                        for lo, hi in ((0, 256), (256, N)):
                            e.accumulate_rows(combos, lo, hi)
                    e.finalize()
                    assert np.array_equal(e.get_counts()[keep], raw[keep]), (sigma, skip, share, how)
                    e.close()


def test_pybind_surface_on_gpu(native):
    """The drop-in class: same calls as the reference's users make (test/run_check.py:45-49)."""
    from fastsk import FastSK
    d = load_golden("f4_ep300_exact")
    X = [d["tokens"][d["offsets"][i]:d["offsets"][i + 1]].tolist() for i in range(d["n_train"] + d["n_test"])]
    f = FastSK(g=10, m=6, t=4)
    f.compute_kernel(X[:d["n_train"]], X[d["n_train"]:])
    assert np.array_equal(np.array(f.get_train_kernel()), d["train"])
    assert np.array_equal(np.array(f.get_test_kernel()), d["test"])
    assert f.get_stdevs() == []
    assert np.array_equal(f.get_counts_np(), d["counts"])
    d = load_golden("f4_ep300_variance_T1")
    f = FastSK(10, 6, 1, True, 0.025, 17)
    f.set_combo_order(d["order"].tolist())
    f.compute_kernel(X[:60], X[60:])
    assert np.array_equal(np.array(f.get_stdevs()), d["stdevs"])
    assert np.array_equal(f.get_test_kernel_np(), d["test"])
    # the same WITHOUT an injected order: seed= is the reference's seed (its std::shuffle with time(0) == 777 drew d["order"])
    f = FastSK(10, 6, 1, True, 0.025, 17, seed=int(d["seed"]))
    f.compute_kernel(X[:60], X[60:])
    assert np.array_equal(np.array(f.get_stdevs()), d["stdevs"])
    assert np.array_equal(f.get_test_kernel_np(), d["test"]) and np.array_equal(f.get_train_kernel_np(), d["train"])
    f = FastSK(g=10, m=6)
    f.compute_train(X[:60])
    assert f.get_test_kernel() == []
    f = FastSK(g=10, m=6)
    f.compute_kernel_flat(d["tokens"].astype(np.int32), d["offsets"].astype(np.int64), 60)
    assert np.array_equal(f.get_test_kernel_np(), load_golden("f4_ep300_exact")["test"])
    A = np.array(X, dtype=np.int32)  # EP300 rows all have length 100: the 2-D array fast path
    f = FastSK(g=10, m=6)
    f.compute_kernel(A[:60], A[60:])
    assert np.array_equal(f.get_train_kernel_np(), load_golden("f4_ep300_exact")["train"])
    with pytest.raises(ValueError):
        FastSK(g=101, m=99).compute_train(X[:10])  # g > shortest (100): the reference exit(1)s
    with pytest.raises(NotImplementedError):
        f.fit()


def test_edge_cases_and_error_convention(native, port):
    """Degenerate shapes the reference accepts (one sequence, sequences of exactly g symbols, all
    sequences equal, m = g - 1, a single combo) and the situations where it exits: status codes."""
    for X, ntr, nte, g, m in EDGE_CASES:
        tokens, offsets = native.flatten(X)
        want, _, _ = port.compute(tokens, offsets, ntr, nte, g, m, t=1)
        for path in (0, 1, 2):
            sigma = len(np.unique(tokens))
            if path == 1 and sigma ** (g - m) > 16384:
                continue
            e = native.Engine(g, m, path=path)
            e.compute(tokens, offsets, ntr, nte)
            assert np.array_equal(e.get_triangle(), want), (X, g, m, path)
            assert e.get_test().shape == (nte, ntr)
            e.close()
    with pytest.raises(native.FskError) as ei:      # m must be < g
        native.Engine(3, 3)
    assert ei.value.code == -1
    e = native.Engine(6, 2)
    tok, off = native.flatten([[1, 2, 3, 4, 1, 2, 3], [1, 2, 3, 4, 1]])
    with pytest.raises(native.FskError) as ei:      # reference: printf + exit(1), fastsk.cpp:53-58
        e.compute(tok, off, 1, 1)
    assert ei.value.code == -2 and "shortest test sequence has length 5" in str(ei.value)
    with pytest.raises(native.FskError) as ei:      # nothing computed yet
        e.get_train()
    assert ei.value.code == -3
    with pytest.raises(native.FskError) as ei:      # no sequences at all
        e.compute(np.zeros(0, np.int32), np.zeros(1, np.int64), 0, 0)
    assert ei.value.code == -1
    tok, off = native.flatten([[1, 2, 3, 4, 1, 2, 3], [1, 2, 3, 4, 1, 4]])
    e.load_sequences(tok, off, 2, 0)
    with pytest.raises(native.FskError) as ei:      # combo id out of range (C(6,2) = 15)
        e.accumulate([15])
    assert ei.value.code == -1
    e.accumulate(np.arange(15))
    e.finalize()
    assert e.get_test().shape == (0, 2) and e.get_train()[0, 0] == 1.0
    e.close()


@pytest.mark.parametrize("shard", ["combos", "rows"])
def test_bench_two_ranks_sharing_the_gpu(native, shard):
    """bench.py's multi-rank flow end to end, started the way a user would — `python bench.py --gpus 2`,
    no external launcher: bench.py spawns its own ranks before touching the GPU — both decompositions
    with full steps, the JSON contract, two ranks on cuda:0 over gloo: everything but RCCL."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, FSK_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--n-seq", "8000", "--shard", shard], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])   # the JSON line is the last thing printed
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "comm"):
        assert key in d
    assert d["n_gpus"] == 2 and d["unit"] == "combos/s" and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["combos"] == 495 and d["config"]["n_seq"] == 8000
    # BASELINE.md's gate "result identical at 1/2/4/8 GPUs": the digest of the reduced triangle, folded on the
    # device by every rank, equals the committed single-GPU digest of this workload (profiles/k_digests.json)
    assert d["bit_identical_to_1gpu"] is True and d["k_digest"]["ranks_agree"] and d["k_digest"]["committed"] is not None
    assert d["alt"]["bit_identical_to_1gpu"] is True and d["alt"]["k_digest"]["sum"] == d["k_digest"]["sum"]
    assert ("row-band" in d["config"]["parallelism"]) == (shard == "rows")
    assert ("combo-sharded" in d["config"]["parallelism"]) == (shard == "combos")
    assert d["alt"]["value"] > 0 and ("row-band" in d["alt"]["parallelism"]) == (shard == "combos")
    assert d["alt"]["steps"] == 2 and d["alt"]["warmup"] == 1          # both decompositions at equal weight
    assert d["comm"]["rccl_ranks"] == 0 and "gloo" in d["comm"]["backend"]  # (a shared GPU cannot run RCCL)
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] <= 1.0


def _bench_env(**extra):
    env = dict(os.environ, **extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FSK_BENCH_FORCE_DIST"):
        env.pop(k, None)
    return env


def test_bench_fails_fast_with_a_json_line_when_a_rank_hangs(native):
    """First multi-GPU contact must not burn the lease silently: a rank that stops answering (here rank 1 of two ranks
    sharing the GPU over gloo, stalled for 90 s inside the timed steps — FSK_BENCH_STALL, test-only) makes rank 0's
    collective wait; the watchdog's bound for the stage (8 s per step here) expires, rank 0 prints ONE JSON line with
    "error", "stage", "n_gpus" and the timings so far, and the run ends non-zero well inside the bound — no hang,
    no re-exec. Before that the same job runs clean with the preflight (N = 16000, committed digest) in its line."""
    import json
    import subprocess
    import sys
    import time
    from conftest import ROOT
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n-seq", "16000",
            "--no-alt", "--no-inproc-leg"]
    r = subprocess.run(base, env=_bench_env(FSK_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["bit_identical_to_1gpu"] is True and d["preflight"]["bit_identical_to_1gpu"] is True and d["preflight"]["every_rank_agrees"]
    assert d["fail_fast"]["watchdog"] and d["memory_plan_GB"]["triangle_u64"] > 1.0
    t0 = time.perf_counter()
    r = subprocess.run(base + ["--step-bound", "8"], env=_bench_env(FSK_BENCH_SHARE_GPU="1", FSK_BENCH_STALL="1:timed steps:90"),
                       capture_output=True, text=True, timeout=300)
    dt = time.perf_counter() - t0
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-1500:] + r.stderr[-1500:]
    e = json.loads(lines[-1])
    assert "error" in e and e["n_gpus"] == 2 and e["value"] is None and "timed steps" in e["stage"], e
    assert e["timings"] and "preflight" in e["partial"]
    assert dt < 85, dt   # (well before the stalled rank would have come back)


def test_bench_inproc_reports_a_stuck_exchange(native):
    """The same for the in-process engine (fsk_create_multi): engine 1's exchange stream is held for 6 s in front of
    band 0's all-reduce (tests/hooks' build of the engine: fault_* tuning keys, a bounded spin kernel), the engine's deadline is the step bound (2 s): fsk_finalize
    returns FSK_EDEVICE naming the band, bench.py prints the JSON error line and exits 2."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--inproc", "--steps", "1", "--warmup", "0",
                        "--n-seq", "8000", "--no-cpu-baseline", "--no-also", "--step-bound", "2", "--engine-lib", hooks_library(native).path],
                       env=_bench_env(FSK_BENCH_SHARE_GPU="1", FSK_TUNING="fault_kind=2,fault_rank=1,fault_band=0,fault_ms=6000"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, r.stdout[-1500:] + r.stderr[-1500:]
    e = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "band 0" in e["error"] and "2000 ms" in e["error"] and e["n_gpus"] == 2 and e["value"] is None


def test_bench_rccl_leg_on_one_rank(native):
    """The RCCL leg itself (backend nccl, device_id, band-wise int32 all-reduce ordered by stream events)
    with a world of one — the most this single-GPU box can run of it — and config 4 through the
    sparse dataflow."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, FSK_BENCH_FORCE_DIST="1", MASTER_PORT="29741")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for extra, path in ((["--n-seq", "8000"], "dense"), (["--config", "4"], "sparse")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                            "--no-cpu-baseline", "--no-also"] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["comm"]["rccl_ranks"] == 1 and "RCCL" in d["comm"]["backend"] and d["comm"]["allreduce_dtype"] == "int32"
        assert d["config"]["path"] == path and d["value"] > 0
        assert d["bit_identical_to_1gpu"] is True   # int32 narrowing + RCCL left every cell what one engine computes
        if path == "sparse":
            assert d["config"]["combos"] == 1001 and d["roofline"]["bound"] == "hbm" and d["roofline"]["cell_updates_per_step"] > 1e9


@pytest.mark.parametrize("gpus,share", [(1, False), (3, True)])
def test_bench_inproc(native, gpus, share):
    """`bench.py --gpus N --inproc`: the job from ONE process through fsk_create_multi (no torch in it). One
    GPU: RCCL from the host C++ over a world of one; three engines sharing the GPU: the P2P kernels."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FSK_BENCH_FORCE_DIST"):
        env.pop(k, None)
    if share:
        env["FSK_BENCH_SHARE_GPU"] = "1"
    for extra, path in ((["--n-seq", "8000"], "dense"), (["--config", "4"], "sparse")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--inproc", "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline", "--no-also"] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["n_gpus"] == gpus and d["value"] > 0 and "in-process" in d["config"]["parallelism"] and d["config"]["path"] == path
        assert d["bit_identical_to_1gpu"] is True
        assert d["comm"]["ranks"] == gpus and d["comm"]["rccl_ranks"] == (0 if share else 1)
        assert d["comm"]["allreduce_dtype"] == "int32" and sum(d["comm"]["combos_per_engine"]) == d["config"]["combos"]
        assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


@pytest.mark.parametrize("surface", ["ctypes", "pybind"])
def test_engine_before_torch_in_one_process(native, surface):
    """A PyTorch-ROCm wheel carries its own HIP/HSA runtime; two runtimes in one process leave the
    second without a GPU. Importing the engine FIRST and torch afterwards must still give both a
    working device (fastsk_amd binds to torch's copies when torch is installed)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = """
import sys
sys.path.insert(0, %r)
X = [[1, 2, 3, 1, 2, 3, 1], [2, 3, 1, 2, 3, 1, 2], [1, 1, 2, 2, 3, 3, 1]]
if %r == "ctypes":
    from fastsk_amd import _native
    e = _native.Engine(3, 1)
else:
    from fastsk import FastSK
    e = FastSK(3, 1)
assert "torch" not in sys.modules
import torch
assert torch.cuda.is_available()
assert float(torch.ones(8, device="cuda").sum()) == 8.0
if %r == "ctypes":
    tok, off = _native.flatten(X)
    e.compute(tok, off, 3, 0)
    K = e.get_train()
else:
    e.compute_train(X)
    K = e.get_train_kernel()
assert K[0][0] == 1.0 and K[1][1] == 1.0 and 0.0 < K[1][0] <= 1.0
print("both fine")
""" % (ROOT, surface, surface)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "both fine" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_diag_exchange_on_a_triangle_beyond_2_31_cells():
    """The row-sharded multi-GPU path gathers and scatters the N diagonal cells of the bound
    triangle with torch indexing; at config 5 that tensor has 5e9 cells. Same indexing here on
    2.45e9 cells (N = 70000): offsets above 2^31 must address the right cells."""
    import torch
    from fastsk_amd.distributed import diag_index, cell
    N = 70000
    K = torch.zeros(cell(N), dtype=torch.int64, device="cuda")
    idx = diag_index(N, K.device)
    assert int(idx[-1]) == cell(N) - 1 and int(idx[-1]) > 2 ** 31
    vals = torch.arange(1, N + 1, dtype=torch.int64, device="cuda")
    K[idx] = vals
    assert torch.equal(K[idx], vals)
    for i in (0, 1, 46340, 46341, 65535, 65536, N - 1):   # around the int32 / uint16 edges
        assert int(K[cell(i) + i]) == i + 1
        if i:
            assert int(K[cell(i) + i - 1]) == 0
    assert int(K.sum()) == N * (N + 1) // 2
    d = K[idx]
    d[:1000] = 0
    d[N - 1000:] = 0
    K[idx] = d
    assert int(K.sum()) == N * (N + 1) // 2 - sum(range(1, 1001)) - sum(range(N - 999, N + 1))


def test_reset_then_storing_launch(native, port, monkeypatch):
    """fsk_reset_counts leaves the zeros to the next tile launch when that launch can STORE its
    sums (dense dataflow, one workgroup per tile, rows starting at the reset range's lower edge);
    every other consumer of K gets the zeros filled in first."""
    rng = np.random.default_rng(123)
    N = 520
    X = rng.integers(1, 5, size=(N, 48), dtype=np.int32)
    X[::11, 5:40] = 2   # rows with counts above 15
    tokens, offsets = native.flatten(X)
    ca, cb = np.arange(0, 70, 3, dtype=np.int32), np.arange(1, 70, 4, dtype=np.int32)
    wa, _, _ = port.raw_counts(tokens, offsets, 8, 4, ca, threads=4)
    wb, _, _ = port.raw_counts(tokens, offsets, 8, 4, cb, threads=4)
    cell = lambda r: r * (r + 1) // 2
    for splits in ("1", "0"):   # "1": one workgroup per tile -> the storing launch; "0": automatic splits -> zero fill + atomics
        set_tuning_env(monkeypatch, tile_splits=splits)
        e = native.Engine(8, 4, path=1)
        e.load_sequences(tokens, offsets, N, 0)
        e.accumulate(ca)                        # K now holds data that every reset below must erase
        e.reset_counts(); e.accumulate(cb); e.finalize()
        assert np.array_equal(e.get_counts(), wb), splits
        e.reset_counts()
        assert not e.get_counts().any()         # a getter right after the reset sees zeros
        e.accumulate(ca); e.accumulate(cb); e.finalize()
        assert np.array_equal(e.get_counts(), wa + wb)
        e.reset_counts()                        # ascending row bands: the first stores, the rest of the reset range follows
        for lo, hi in ((0, 128), (128, 384), (384, N)):
            e.accumulate_rows(cb, lo, hi)
        e.finalize()
        assert np.array_equal(e.get_counts(), wb)
        e.reset_counts()                        # bands out of order: the fill must happen before the first of them
        for lo, hi in ((256, N), (0, 256)):
            e.accumulate_rows(ca, lo, hi)
        e.finalize()
        assert np.array_equal(e.get_counts(), wa)
        e.reset_counts_rows(128, 384)           # only these rows are reset; the others keep `wa`
        e.accumulate_rows(cb, 128, 384)
        e.synchronize()
        want = wa.copy()
        want[cell(128):cell(384)] = wb[cell(128):cell(384)]
        assert np.array_equal(e.get_counts(), want)
        e.reset_counts_rows(128, 384)           # reset rows, then touch OTHER rows first
        e.accumulate_rows(cb, 0, 128)
        e.accumulate_rows(cb, 128, 384)
        e.finalize()
        want[cell(0):cell(128)] += wb[cell(0):cell(128)]
        assert np.array_equal(e.get_counts(), want)
        e.close()


def test_sequential_sum_random_inputs(native):
    """The device's left-to-right fp64 sum on 300 random vectors of five kinds (wide magnitude ranges,
    squares scaled by 1 - 1/i as the Welford products are, dyadic values that tie at every addition, mostly
    zeros, one huge value among small ones; 1 to 60,000 values: whole blocks, group records, tails) — bit
    for bit against the host loop."""
    from test_sequential_sum import sequential
    e = native.Engine(4, 2)
    rng = np.random.default_rng(9)
    for t in range(300):
        n = int(rng.integers(1, 60000))
        kind = t % 5
        if kind == 0:
            x = rng.random(n) * 10.0 ** rng.integers(-8, 8)
        elif kind == 1:
            x = (rng.integers(0, 40, n) ** 2) * (1.0 - 1.0 / rng.integers(2, 60))
        elif kind == 2:
            x = np.ldexp(rng.integers(1, 1 << 20, n).astype(np.float64), rng.integers(-30, 30, n))
        elif kind == 3:
            x = np.where(rng.random(n) < 0.7, 0.0, rng.random(n))
        else:
            x = np.full(n, 0.5 * 2.0 ** rng.integers(-5, 5))
            x[rng.integers(0, n)] = 2.0 ** 40
        assert np.float64(e.sequential_sum(x)).tobytes() == np.float64(sequential(x)).tobytes(), (t, kind, n)
    e.close()


@pytest.mark.parametrize("path", [0, 2, -1])
def test_variance_mode_stops_anywhere(native, port, path, monkeypatch):
    """Variance mode runs ahead of its stop test (batches of 4 iterations, two in flight, second
    stream): whatever the chain count, max_iters and delta, stdevs and the kernel are
    the oracle's, bit for bit — on both dataflows the counts of a batch's iterations sit in triangles of
    their own and the Welford state is written once per batch (a stop inside a batch runs its prefix again);
    path -1: the dense dataflow with its per-iteration triangles switched off."""
    if path == -1:
        set_tuning_env(monkeypatch, variance_dense_slots="0")
        path = 1
    rng = np.random.default_rng(5)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(12, 40, size=30)]
    tok, off = native.flatten(X)
    g, m = 7, 3
    order = rng.permutation(port.num_combos(g, m)).astype(np.int32)
    lengths = set()
    for T in (1, 2, 3):
        for max_iters in (-1, 1, 2, 4, 5, 6, 9):
            for delta in (0.025, 0.2, 0.5, 1.0, 3.0):
                want, sd, _ = port.compute(tok, off, 22, 8, g, m, t=T, approx=True, delta=delta, max_iters=max_iters, order=order)
                e = native.Engine(g, m, t=T, approx=True, delta=delta, max_iters=max_iters, path=path)
                e.set_combo_order(order)
                e.compute(tok, off, 22, 8)
                assert np.array_equal(e.get_stdevs(), sd), (T, max_iters, delta)
                assert np.array_equal(e.get_triangle(), want), (T, max_iters, delta)
                lengths.add(len(sd))
                e.close()
    assert len(lengths) >= 6  # stops landed at many different places inside the batches


@pytest.mark.parametrize("form", ["dense_slots", "dense_fill", "sparse", "sparse_u32"])
def test_variance_mode_count_above_255(native, port, form, monkeypatch):
    """Variance mode, dense dataflow, a sequence with more than 255 equal windows: the iteration is
    diverted to the general dataflow, which adds into the iteration's own triangle — zeroed first
    (the storing tile launch that normally writes every cell of it is not coming)."""
    if form == "dense_fill":
        set_tuning_env(monkeypatch, variance_dense_slots="0")
    rng = np.random.default_rng(11)
    X = [rng.integers(1, 5, size=int(l)).astype(np.int32) for l in rng.integers(280, 320, size=24)]
    for i in (2, 9, 17):
        X[i][5:295] = 1
    tok, off = native.flatten(X)
    g, m, T = 5, 2, 2
    order = np.random.default_rng(3).permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tok, off, 16, 8, g, m, t=T, approx=True, delta=0.025, max_iters=-1, order=order)
    if form == "sparse_u32":
        set_tuning_env(monkeypatch, var_slots16="0")
    e = native.Engine(g, m, t=T, approx=True, path=2 if form.startswith("sparse") else 1)
    e.set_combo_order(order)
    e.compute(tok, off, 16, 8)
    assert e.stats()["max_windows"] > 255
    assert np.array_equal(e.get_stdevs(), sd)
    assert np.array_equal(e.get_triangle(), want)
    # (sparse dataflow: u16 slot triangles until a sum does not fit one — two 286-window runs of one k-mer meet in a cell of
    # 81,796 — then that batch once more with u32 triangles, which the sequences keep)
    if form == "sparse":
        assert e.stats()["batches_redone"] == 1
    if form == "sparse_u32":
        assert e.stats()["batches_redone"] == 0
    e.close()


def test_sequential_sum_on_the_device(native):
    """The device replacement of get_variance's sequential fp64 sum (fastsk_kernel.cpp:116-131): the
    adversarial cases of tests/test_sequential_sum.py on the real kernels, plus a 2.7 M-value sum of
    the size config 1 produces per iteration."""
    from test_sequential_sum import cases, sequential
    e = native.Engine(4, 2)
    for name, x in cases().items():
        assert np.float64(e.sequential_sum(x)).tobytes() == np.float64(sequential(x)).tobytes(), name
    rng = np.random.default_rng(3)
    x = (rng.integers(0, 30, 2_736_630) ** 2) * (1.0 - 1.0 / 11.0)
    x[rng.random(len(x)) < 0.6] = 0.0
    want = float(np.add.accumulate(x)[-1])  # accumulate is the plain left-to-right loop
    assert np.float64(e.sequential_sum(x)).tobytes() == np.float64(want).tobytes()
    e.close()


@pytest.mark.parametrize("env", [{}, {"sparse_sync": "1"}, {"guard_cap": "5000"}])
def test_sparse_batches_enqueued_ahead_of_their_size(native, port, monkeypatch, env):
    """Sparse dataflow, several accumulate calls: batches after the first are enqueued before their
    update-word count is known; the same counts and U as batches sized one by one, also when every
    such batch overflows the (shrunk) guard and is redone. (Without descriptors: the batch that switches them on forgets the
    words per record seen so far, and the next one is sized exactly instead of under a guard.)"""
    set_tuning_env(monkeypatch, sparse_desc="-1", **env)
    tokens, offsets = synthetic_dna(900, 120, seed=21)
    g, m = 12, 6
    combos = np.arange(0, port.num_combos(g, m), 5, dtype=np.int32)
    want, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    e = native.Engine(g, m, path=2)
    e.load_sequences(tokens, offsets, 600, 300)
    parts = np.array_split(combos, 5)
    for part in parts:
        e.accumulate(part)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    st = e.stats()
    assert st["cell_updates"] == U
    assert st["batches_redone"] == (len(parts) - 1 if "guard_cap" in env else 0)
    e.close()


@pytest.mark.parametrize("env", [{}, {"sparse_exact_lanes": "2"}, {"sparse_exact_lanes": "1"}, {"guard_cap": "5000"}, {"sparse_global": "1"}])
def test_sparse_exact_accumulate_in_two_lanes(native, port, monkeypatch, env):
    """Sparse dataflow: the batches of ONE exact accumulate alternate between two lanes (a scratch set and a stream each);
    their consume passes — plain read-modify-writes of K — are ordered by events. Same counts and U as on one stream,
    with every guarded batch redone, and with atomics instead of streams; repeated, so that a race would show."""
    set_tuning_env(monkeypatch, **env)
    tokens, offsets = synthetic_dna(900, 120, seed=33)
    g, m = 12, 6
    nfeat = 900 * (120 - g + 1)
    set_tuning_env(monkeypatch, sparse_batch_records=str(6 * nfeat))  # (six combos a batch)
    combos = np.arange(0, port.num_combos(g, m), 7, dtype=np.int32)
    want, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    e = native.Engine(g, m, path=2)
    e.load_sequences(tokens, offsets, 600, 300)
    for rep in range(3):
        e.reset_counts()
        e.accumulate(combos)
        e.finalize()
        assert np.array_equal(e.get_counts(), want), rep
    st = e.stats()
    assert st["cell_updates"] == 3 * U
    if "guard_cap" in env:
        assert st["batches_redone"] > 0
    e.close()


@pytest.mark.parametrize("env", [{"sparse_global": "1"}, {"list_max_words": "200000"}, {}, {"sparse_sync": "1"},
                                 {"guard_cap": "5000"}])
def test_variance_mode_sparse_forms(native, monkeypatch, env):
    """Variance mode through the sparse dataflow in its forms (grouped batches with a u32 triangle
    per slot; atomics; ungrouped after a batch too large for one stream; batches sized one by one;
    batches that overflow their guard and are redone): the reference's stdevs and
    triangle, bit for bit, on the protein slice and on BASELINE config 1 at full size."""
    set_tuning_env(monkeypatch, **env)
    for name in ("f5_prot11_variance_T1", "f7_cfg1_prot11_approx_t1"):
        d = load_golden(name)
        if "tokens" in d:
            tokens, offsets, ntr, nte = d["tokens"], d["offsets"], d["n_train"], d["n_test"]
        else:
            tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
        e = engine_for(native, d, path=2)
        e.compute(tokens, offsets, ntr, nte)
        assert np.array_equal(e.get_stdevs(), d["stdevs"]), name
        if "tri" in d:
            assert np.array_equal(e.get_triangle(), d["tri"]), name
        else:
            assert sha(e.get_triangle()) == d["tri_sha256"], name
        e.close()


def test_skip_test_block(native, port):
    """skip_test_block=1: everything a getter of the reference exposes (train x train, test x train,
    hence every diagonal entry) is unchanged; tiles made of test x test cells only are not computed."""
    N, ntr = 1500, 600   # first all-test tile column: ceil(600/128) = 5 -> cells with column >= 640 are skipped
    tokens, offsets = synthetic_dna(N, 80, seed=3)
    want, _, _ = port.compute(tokens, offsets, ntr, N - ntr, 9, 5, t=1)
    sq = tri_to_square(want, N)
    raw, _, _ = port.raw_counts(tokens, offsets, 9, 5, np.arange(port.num_combos(9, 5), dtype=np.int32), threads=8)
    for bands in (False, True):
        e = native.Engine(9, 5, path=1, skip_test_block=True)
        if bands:
            e.load_sequences(tokens, offsets, ntr, N - ntr)
            for lo, hi in ((0, 512), (512, 1024), (1024, N)):
                e.accumulate_rows(np.arange(126, dtype=np.int32), lo, hi)
            e.finalize()
        else:
            e.compute(tokens, offsets, ntr, N - ntr)
        assert np.array_equal(e.get_train(), sq[:ntr, :ntr])
        assert np.array_equal(e.get_test(), sq[ntr:, :ntr])
        got = tri_to_square(e.get_counts(), N).astype(np.int64)
        ref = tri_to_square(raw, N).astype(np.int64)
        assert np.array_equal(np.diag(got), np.diag(ref))
        i, j = np.tril_indices(N, -1)
        skipped = (j // 128 >= 5) & (i // 128 != j // 128)
        assert not got[i[skipped], j[skipped]].any() and skipped.sum() > 200000
        assert np.array_equal(got[i[~skipped], j[~skipped]], ref[i[~skipped], j[~skipped]])
        e.close()


@pytest.mark.parametrize("global_pairs", ["0", "1"])
def test_skip_test_block_sparse(native, port, monkeypatch, global_pairs):
    """skip_test_block=1 on the sparse dataflow: a test row pairs only with the train entries of its k-mer
    runs and with itself, so exactly the test x test cells off the diagonal stay zero and everything a
    getter of the reference exposes is unchanged — update streams and atomics, whole and in row bands."""
    set_tuning_env(monkeypatch, sparse_global=global_pairs)
    X = protein_like(1400, 40, 160, seed=21)
    N, ntr, g, m = len(X), 500, 10, 6
    tokens, offsets = native.flatten(X)
    combos = np.arange(0, 210, 7, dtype=np.int32)
    raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    ref = tri_to_square(raw, N).astype(np.int64)
    i, j = np.tril_indices(N, -1)
    tt = j >= ntr
    for bands in (False, True):
        e = native.Engine(g, m, path=2, skip_test_block=True)
        e.load_sequences(tokens, offsets, ntr, N - ntr)
        if bands:
            for lo, hi in ((0, 384), (384, 1024), (1024, N)):
                e.accumulate_rows(combos, lo, hi)
        else:
            e.accumulate(combos)
        e.finalize()
        got = tri_to_square(e.get_counts(), N).astype(np.int64)
        assert np.array_equal(np.diag(got), np.diag(ref))
        assert not got[i[tt], j[tt]].any() and ref[i[tt], j[tt]].any()
        assert np.array_equal(got[i[~tt], j[~tt]], ref[i[~tt], j[~tt]])
        assert e.stats()["cell_updates"] < U   # fewer `+=` than the reference issues
        dg = np.diag(ref).astype(np.float64)
        want_test = ref[ntr:, :ntr] / np.sqrt(dg[ntr:, None] * dg[None, :ntr])
        assert np.array_equal(e.get_test(), want_test)
        e.close()


def test_sparse_emit_quarter_tile_passes(native, port):
    """k_sx_emit bins a tile's short entries a quarter at a time when their update words exceed the LDS
    slots: 16 long sequences over 4^5 keys make every run ~16 entries of up to 16 partners (~17 k words per
    2048-entry tile). In one call and in two (the second enqueued ahead of its size), against the oracle."""
    rng = np.random.default_rng(44)
    X = rng.integers(1, 5, size=(16, 6000), dtype=np.int32)
    tokens, offsets = native.flatten(X)
    g, m = 8, 3
    combos = np.arange(0, 56, 5, dtype=np.int32)
    want, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    for parts in (1, 2):
        e = native.Engine(g, m, path=2)
        e.load_sequences(tokens, offsets, 12, 4)
        for part in np.array_split(combos, parts):
            e.accumulate(part)
        e.finalize()
        assert np.array_equal(e.get_counts(), want)
        assert e.stats()["cell_updates"] == U
        e.close()


def test_sparse_guarded_batch_beyond_32_bit_word_total(native):
    """A sparse batch enqueued ahead of its size whose update words pass 2^32 (long runs: 729 keys, 3000
    ragged sequences): the 32-bit stream offsets wrap, so the kernels must judge the batch by the 64-bit
    total — it does not fit its guard, is left alone and redone with atomics. Checked against the dense
    dataflow (this shape found the 32-bit comparison as a GPU memory fault in tools/stress_parity.py)."""
    rng = np.random.default_rng(1)
    g, m, N = 11, 5, 3000
    X = [rng.integers(1, 4, size=int(L)).astype(np.int32) for L in rng.integers(g, 701, size=N)]
    tokens, offsets = native.flatten(X)
    combos = np.array([81, 203, 210, 230, 290, 311, 318, 321, 339, 348, 358], dtype=np.int32)
    d = native.Engine(g, m, path=1)
    d.load_sequences(tokens, offsets, N, 0)
    d.accumulate(combos)
    d.finalize()
    want = d.get_counts()
    d.close()
    for desc in (-1, 1):  # (with descriptors the same pairs are a fraction of the words: nothing overflows, the same counts)
        e = native.Engine(g, m, path=2, tuning={"sparse_desc": desc})
        e.load_sequences(tokens, offsets, N, 0)
        e.accumulate_rows(combos, 0, 768)     # sized exactly: ~0.3 G words, the guard of the next batch is 1.5x that
        e.accumulate_rows(combos, 768, N)     # ~4.3 G words
        e.finalize()
        st = e.stats()
        assert st["cell_updates"] > 2 ** 32 + 2 ** 28 and (desc > 0 or st["batches_redone"] == 1)
        assert np.array_equal(e.get_counts(), want), desc
        e.close()


@pytest.mark.parametrize("forced", ["0", "1"])
def test_sparse_segment_scan_in_chunks(native, port, monkeypatch, forced):
    """A sparse batch of more than 4096 entry tiles scans its tile records in three launches (chunk
    totals, the chunks, the tiles with their carries); forced=1 takes that form for the small batches of
    the row bands too. Counts against the oracle, with and without skip_test_block."""
    set_tuning_env(monkeypatch, seg_scan_chunked=forced)
    X = protein_like(2600, 60, 260, seed=33)
    N, ntr, g, m = len(X), 1700, 8, 4
    tokens, offsets = native.flatten(X)
    combos = np.arange(0, 70, 2, dtype=np.int32)   # 35 combos x ~200 tiles each: one batch of ~7000 tiles
    raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    i, j = np.tril_indices(N)
    for skip in (False, True):
        e = native.Engine(g, m, path=2, skip_test_block=skip)
        e.load_sequences(tokens, offsets, ntr, N - ntr)
        if forced == "1":
            for lo, hi in ((0, 1024), (1024, N)):
                e.accumulate_rows(combos, lo, hi)
        else:
            e.accumulate(combos)
        e.finalize()
        ref = raw.copy()
        if skip:
            ref[(j >= ntr) & (i != j)] = 0
        assert np.array_equal(e.get_counts(), ref), skip
        if not skip:
            assert e.stats()["cell_updates"] == U
        e.close()


def test_device_resident_block_getter(native):
    """fsk_get_block_device: the normalised block straight into a torch tensor on the GPU."""
    d = load_golden("f4_ep300_exact")
    e = engine_for(native, d)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    t = e.get_block_torch(0, d["n_train"], 0, d["n_train"])
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), d["train"])
    t = e.get_block_torch(d["n_train"], d["n_train"] + d["n_test"], 0, d["n_train"])
    assert np.array_equal(t.cpu().numpy(), d["test"])
    # fsk_get_triangle_device: the reference's K itself (fastsk_kernel.cpp:96-103 over every cell), written on the device
    tri = e.get_triangle_torch()
    assert tri.is_cuda and np.array_equal(tri.cpu().numpy(), d["tri"]) and np.array_equal(e.get_triangle(), d["tri"])
    e.close()


SAVE_CASES = ["f3_ragged_sigma7_g6m3", "f4_ep300_exact", "f3_train_only", "f4_ep300_variance_T1", "f6_prot219_exact"]


@pytest.mark.parametrize("name", SAVE_CASES)
def test_save_kernel_whole_file_against_the_reference(native, tmp_path, name):
    """fastsk.FastSK(...).save_kernel(path) (pybind11 class) writes, byte for byte, the file the REFERENCE's own
    FastSK::save_kernel (fastsk.cpp:223-237) wrote for the same input — tests/golden/save_kernel.npz holds those
    bytes (tests/make_golden_save_kernel.py: compiled reference, compute then save_kernel). Covers the lazy
    test x test block (save_kernel dumps the whole N x N matrix), ragged input, train only, and variance mode."""
    import hashlib
    from fastsk import FastSK
    z = np.load(os.path.join(GOLD, "save_kernel.npz"))
    want = z[name].tobytes()
    assert hashlib.sha256(want).hexdigest() == str(z[name + "__sha256"])
    d = load_golden(name)
    N = d["n_train"] + d["n_test"]
    X = [d["tokens"][d["offsets"][i]:d["offsets"][i + 1]].tolist() for i in range(N)]
    f = FastSK(d["g"], d["m"], d["t"], bool(d["approx"]), d["delta"], d["max_iters"], bool(d["skip_variance"]))
    if d["approx"]:
        f.set_combo_order(d["order"].tolist())
    if d["n_test"]:
        f.compute_kernel(X[:d["n_train"]], X[d["n_train"]:])
    else:
        f.compute_train(X)
    p = tmp_path / "k.txt"
    f.save_kernel(str(p))
    got = p.read_bytes()
    assert got == want
    # and the ctypes view of the same C-ABI call
    e = engine_for(native, d)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    p2 = tmp_path / "k2.txt"
    e.save_kernel(str(p2))
    assert p2.read_bytes() == want
    f.save_kernel("")   # the reference silently does nothing for an empty name


def test_run_check_style_auc(native):
    """The reference's only CI test (test/run_check.py:37-64): EP300, g=10 m=6 approx t=1,
    LinearSVC + 5-fold calibration, test AUC >= 0.9 — with our module in place of fastsk."""
    from sklearn.svm import LinearSVC
    from sklearn.calibration import CalibratedClassifierCV
    from sklearn.metrics import roc_auc_score
    tokens, offsets, ntr, nte, ytr, yte = load_tokens("EP300")
    e = native.Engine(10, 6, t=1, approx=True)
    e.set_seed(1234)
    e.compute(tokens, offsets, ntr, nte)
    Xtr, Xte = e.get_train(), e.get_test()
    assert len(e.get_stdevs()) >= 2
    clf = CalibratedClassifierCV(LinearSVC(C=1), cv=5).fit(Xtr, ytr)
    auc = roc_auc_score(yte, clf.predict_proba(Xte)[:, 1])
    assert auc >= 0.9, auc


@pytest.mark.parametrize("narrow", [0, 1])
def test_two_ranks_sharing_the_gpu_banded_reduce(native, port, tmp_path, narrow):
    """The multi-GPU host path with the REAL engine: two processes (gloo, both on cuda:0) shard
    the combos, accumulate row bands on their own HIP streams and all-reduce each finished band
    (uint64 and int32-narrowed payloads) while the next band runs. Everything but RCCL itself."""
    import subprocess
    import sys
    from conftest import ROOT
    N, L, g, m = 1500, 120, 10, 6
    tokens, offsets = synthetic_dna(N, L, seed=7)
    combos = np.arange(0, 210, 7, dtype=np.int32)
    fx = tmp_path / "in.npz"
    np.savez(fx, tokens=tokens, offsets=offsets, n_train=N, n_test=0, g=g, m=m, combos=combos)
    want, _, _ = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29701 + narrow), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(fx), str(tmp_path), "4",
                               str(narrow), "cuda:0"], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    done = 0
    for r in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(z["counts"], want)
        assert np.array_equal(z["tri"], port.normalise(want.astype(np.float64), N))
        done += int(z["done"])
    assert done == len(combos)


@pytest.mark.parametrize("replicate", [0, 1])
def test_two_ranks_sharing_the_gpu_row_sharded(native, port, tmp_path, replicate):
    """shard_by="rows" with the REAL engine: two processes on cuda:0, each runs all combos over its
    own band of rows; only the diagonal is exchanged (or finished bands are broadcast)."""
    import subprocess
    import sys
    from conftest import ROOT
    from fastsk_amd.distributed import owner_edges, cell
    N, L, g, m = 1500, 120, 10, 6
    tokens, offsets = synthetic_dna(N, L, seed=11)
    combos = np.arange(0, 210, 7, dtype=np.int32)
    fx = tmp_path / "in.npz"
    np.savez(fx, tokens=tokens, offsets=offsets, n_train=N, n_test=0, g=g, m=m, combos=combos)
    want, _, _ = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    tri = port.normalise(want.astype(np.float64), N)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29711 + replicate), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(fx), str(tmp_path), "3",
                               "1", "cuda:0", "rows", str(replicate)], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    edges = owner_edges(N, 2)
    il = np.tril_indices(N)
    for r in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(z["full"][il], tri)          # assembled from the owners: bit-identical
        assert np.array_equal(z["full"], z["full"].T)
        lo, hi = cell(edges[r]), cell(edges[r + 1])
        assert np.array_equal(z["counts"][lo:hi], want[lo:hi])
        if replicate:
            assert np.array_equal(z["counts"], want)


@pytest.mark.parametrize("T", [2, 4])
def test_two_ranks_sharing_the_gpu_variance_chains(native, port, tmp_path, T):
    """Variance mode over ranks with the REAL engine (two processes on cuda:0, gloo): chain c on rank
    c mod 2, one fp64 all-reduce of the K_hat sums, stdevs from chain 0's rank — protein-like input
    through the sparse dataflow. Two terms add up the same in either order: bit-identical to the oracle
    for T = 2; to rounding for T = 4 (as between the reference's own threads)."""
    import subprocess
    import sys
    from conftest import ROOT
    X = protein_like(400, 30, 120, seed=5)
    tokens, offsets = native.flatten(X)
    g, m, delta, max_iters = 8, 4, 0.05, 11
    order = np.random.default_rng(3).permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tokens, offsets, 300, 100, g, m, t=T, approx=True, delta=delta, max_iters=max_iters, order=order)
    fx = tmp_path / "in.npz"
    np.savez(fx, tokens=tokens, offsets=offsets, n_train=300, n_test=100, g=g, m=m, t=T, delta=delta, max_iters=max_iters, order=order)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29720 + T), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_variance_worker.py"), str(fx), str(tmp_path),
                               "cuda:0"], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for r in range(2):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(z["stdevs"], sd)
        if T == 2:
            assert np.array_equal(z["tri"], want)
        else:
            assert np.allclose(z["tri"], want, rtol=1e-14, atol=0)


@pytest.mark.parametrize("global_pairs", ["0", "1"])
def test_sparse_pair_accumulation_variants(native, port, monkeypatch, global_pairs):
    """Protein-like input through the sparse dataflow with owner-slice LDS accumulation and with
    direct per-pair atomics, whole and in row bands; U equals the oracle's count of `+=`."""
    set_tuning_env(monkeypatch, sparse_global=global_pairs)
    rng = np.random.default_rng(31)
    N = 900
    X = [rng.integers(1, 21, size=int(L)).astype(np.int32) for L in rng.integers(20, 400, size=N)]
    for i in range(0, N, 7):
        X[i][:18] = X[0][:18]  # shared prefixes: runs with many sequences
    tokens, offsets = native.flatten(X)
    combos = np.arange(0, 210, 11, dtype=np.int32)
    want, _, U = port.raw_counts(tokens, offsets, 10, 6, combos, threads=8)
    e = native.Engine(10, 6, path=2)
    e.load_sequences(tokens, offsets, 600, 300)
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    assert e.stats()["cell_updates"] == U
    e.reset_counts()
    for lo, hi in [(0, 256), (256, 768), (768, 900)]:
        e.accumulate_rows(combos, lo, hi)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    e.close()


@pytest.mark.parametrize("hint", [None, "0"])
def test_sparse_words_per_record_hint_across_loads(native, port, monkeypatch, hint):
    """Sparse dataflow: a second set of sequences of the same shape starts from the first set's words per record as a hint
    (its first batch goes out under a guard, at full size); a hint that is far too low costs a redone batch, never a count."""
    set_tuning_env(monkeypatch, sparse_desc="-1")  # (the batch that switches descriptors on forgets the words per record seen so far)
    if hint is not None:
        set_tuning_env(monkeypatch, sparse_hint=hint)
    rng = np.random.default_rng(5)
    N, L, g, m = 700, 90, 9, 4
    A = rng.integers(1, 5, size=(N, L), dtype=np.int32)
    B = rng.integers(1, 3, size=(N, L), dtype=np.int32)   # two symbols: far more update words per record —
    B[0, :4] = [1, 2, 3, 4]                                 # (the alphabet stays the same)
    combos = np.arange(0, port.num_combos(g, m), 4, dtype=np.int32)
    e = native.Engine(g, m, path=2)
    redone = []
    for X in (A, B, B):
        tok, off = native.flatten(X)
        want, _, _ = port.raw_counts(tok, off, g, m, combos, threads=8)
        e.load_sequences(tok, off, 500, 200)
        e.accumulate(combos)
        e.finalize()
        assert np.array_equal(e.get_counts(), want)
        redone.append(e.stats()["batches_redone"])
    assert redone == ([0, 1, 1] if hint is None else [0, 0, 0])
    e.close()


def test_key_compaction_two_rare_symbols(native, port):
    """Key compaction from the places of the rare symbols with TWO of them (an 'n' and an 'r' in DNA: sigma = 6), single
    places and runs, next to each other and at the sequences' ends; the marking pass over every window gives the same."""
    rng = np.random.default_rng(78)
    N, L = 900, 140
    X = rng.integers(1, 5, size=(N, L), dtype=np.int32)
    for i in rng.choice(N, size=30, replace=False):
        X[i, rng.integers(0, L, size=2)] = 5
    for i in rng.choice(N, size=12, replace=False):
        X[i, rng.integers(0, L, size=2)] = 6
    X[3, :5] = 5; X[3, 5:9] = 6; X[4, -6:] = 6; X[7, 60] = 5; X[7, 61] = 6
    tokens, offsets = native.flatten(X)
    g, m = 8, 4
    combos = np.arange(0, port.num_combos(g, m), 3, dtype=np.int32)
    want, _, _ = port.raw_counts(tokens, offsets, g, m, combos, threads=8)
    got = {}
    for rare in ("1", "0"):
        e = native.Engine(g, m, path=1, tuning={"compact_rare": rare})
        e.load_sequences(tokens, offsets, 600, 300)
        e.accumulate(combos)
        e.finalize()
        got[rare] = (e.get_counts(), e.stats()["compact_keys_avg"])
        e.close()
    assert np.array_equal(got["1"][0], want) and np.array_equal(got["0"][0], want)
    assert 256 <= got["0"][1] <= got["1"][1] < 6 ** 4   # (the places' form takes every common key as present: a superset)


@pytest.mark.parametrize("force", [None, "0", "1", "mark"])
def test_key_compaction_rare_symbol(native, port, monkeypatch, force):
    """DNA with a few 'n' (config-3-like): key compaction on (auto / forced) and off give the oracle's counts; with
    compaction the tile kernel multiplies far fewer keys than 5^4."""
    if force == "mark":  # (the keys that occur from a marking pass over every window, not from the places of the rare symbol)
        set_tuning_env(monkeypatch, compact_rare="0")
    elif force is not None:
        set_tuning_env(monkeypatch, compact=force)
    rng = np.random.default_rng(77)
    N, L = 1100, 160
    X = rng.integers(1, 5, size=(N, L), dtype=np.int32)
    for i in rng.choice(N, size=40, replace=False):
        X[i, rng.integers(0, L, size=3)] = 5
    X[5, :30] = 5
    X[9, :] = 3
    tokens, offsets = native.flatten(X)
    combos = np.arange(0, 210, 9, dtype=np.int32)
    want, _, _ = port.raw_counts(tokens, offsets, 10, 6, combos, threads=8)
    e = native.Engine(10, 6, path=1)
    e.load_sequences(tokens, offsets, 800, 300)
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    st = e.stats()
    if force == "0":
        assert st["compact_keys_avg"] == 0
    else:
        assert 256 <= st["compact_keys_avg"] < 625
    e.close()


def test_many_high_count_kmers(native, port):
    """Low-complexity sequences of many kinds in the same panels: more rows with counts above 15
    per 32-row stage than the 4 whose hi plane rides along with the prefetch."""
    rng = np.random.default_rng(3)
    L, N = 180, 700
    X = rng.integers(1, 5, size=(N, L), dtype=np.int32)
    pats = [[1], [2], [3], [4], [1, 2], [3, 4], [1, 3], [2, 4], [1, 2, 3], [4, 3, 2, 1], [1, 1, 2], [3, 3, 4, 4]]
    for i, pat in enumerate(pats):
        for rep in range(4):
            X[7 + 53 * i + rep * 3] = np.array((pat * L)[:L], dtype=np.int32)
    tokens, offsets = native.flatten(X)
    combos = np.arange(0, 210, 13, dtype=np.int32)
    want, _, _ = port.raw_counts(tokens, offsets, 10, 6, combos, threads=8)
    e = native.Engine(10, 6, path=1)
    e.load_sequences(tokens, offsets, 500, 200)
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    e.close()


# ---- one engine over several devices of the process (fsk_create_multi) ---------------------------------
def group_cases():
    return [("rccl_world_of_one", [0], "auto"), ("p2p_two_engines", [0, 0], "auto"), ("p2p_four_engines", [0, 0, 0, 0], "p2p")]


@pytest.mark.parametrize("label,devices,collective", group_cases())
@pytest.mark.parametrize("name", ["f4_ep300_exact", "f5_prot11_exact", "f4_ep300_skipvar_T3", "f5_prot11_variance_T1_it9"])
def test_group_golden_vectors(native, label, devices, collective, name):
    """FastSK(devices=[...])'s engine on the GPU box: a world of one goes through the SAME banded exchange
    with RCCL called from the host C++ (ncclCommInitAll + ncclAllReduce); a device listed twice or four
    times runs the group's worker threads, streams and events for real, with the engine's peer-to-peer
    all-reduce kernels as the collective (RCCL cannot put two ranks on one GPU)."""
    d = load_golden(name)
    e = engine_for(native, d, devices=devices, collective={"auto": 0, "rccl": 1, "p2p": 2}[collective])
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    info = e.multi_info()
    assert info["ndev"] == len(devices) and info["comm_ranks"] == len(devices)
    assert info["collective"] == ("rccl" if label.startswith("rccl") else "p2p"), info
    assert np.array_equal(e.get_triangle(), d["tri"])
    assert np.array_equal(e.get_train(), d["train"]) and np.array_equal(e.get_test(), d["test"])
    assert np.array_equal(e.get_stdevs(), d["stdevs"])
    if "counts" in d:
        assert np.array_equal(e.get_counts(), d["counts"])
    e.close()


@pytest.mark.parametrize("label,devices,collective", group_cases())
def test_group_banded_exchange_against_one_engine(native, label, devices, collective):
    """N = 9000 DNA sequences: the dense dataflow in 8 equal-area row bands, every band's int32 all-reduce on
    the exchange stream under the next band's tile kernels; two passes (the second starts from lazily
    reset triangles), then an additive third accumulate. Digest and sampled blocks against ONE engine."""
    tokens, offsets = synthetic_dna(9000, 60, seed=77)
    g, m = 8, 4
    combos = np.arange(0, 70, 2, dtype=np.int32)
    one = native.Engine(g, m)
    one.load_sequences(tokens, offsets, 9000, 0)
    one.accumulate(combos)
    one.finalize()
    want = one.counts_digest()
    e = native.Engine(g, m, devices=devices, collective={"auto": 0, "rccl": 1, "p2p": 2}[collective])
    e.load_sequences(tokens, offsets, 9000, 0)
    for _ in range(2):
        e.reset_counts()
        e.accumulate(combos)
        e.finalize()
        info = e.multi_info()
        assert info["bands"] == 8 and info["narrow"] and info["reduce_bytes"] == 4 * e.pairs
        assert e.counts_digest() == want
    assert np.array_equal(e.get_counts_block(8000, 8200, 100, 400), one.get_counts_block(8000, 8200, 100, 400))
    assert np.array_equal(e.get_block(0, 300, 0, 300), one.get_block(0, 300, 0, 300))
    extra = np.array([1, 3, 5], dtype=np.int32)
    one.accumulate(extra); one.finalize()
    e.accumulate(extra); e.finalize()
    assert e.counts_digest() == one.counts_digest()
    assert e.stats()["combos_done"] == len(combos) + len(extra)
    one.close(); e.close()


def test_group_sparse_and_wide_cells(native, port):
    """The sparse dataflow in a group (one band, every engine sorts its own combos) and the uint64 exchange
    (one very long sequence: C(g,m) * max_windows^2 >= 2^31)."""
    d = load_golden("f6_prot219_exact")
    for devices in ([0, 0], [0, 0, 0]):
        e = engine_for(native, d, path=2, devices=devices)
        e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
        assert np.array_equal(e.get_counts(), d["counts"]) and np.array_equal(e.get_triangle(), d["tri"])
        assert e.multi_info()["bands"] == 1
        e.close()
    rng = np.random.default_rng(21)
    X = [rng.integers(1, 5, size=n).astype(np.int32) for n in (20000, 50, 64, 41, 77, 58)]
    tok, off = native.flatten(X)
    want, _, _ = port.compute(tok, off, 4, 2, 6, 2, t=1)
    for devices in ([0], [0, 0]):
        e = native.Engine(6, 2, devices=devices)
        e.compute(tok, off, 4, 2)
        assert not e.multi_info()["narrow"]
        assert np.array_equal(e.get_triangle(), want)
        e.close()


@pytest.mark.parametrize("T", [2, 5])
def test_group_variance_chains(native, port, T):
    rng = np.random.default_rng(5)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(12, 40, size=30)]
    tok, off = native.flatten(X)
    g, m = 7, 3
    order = rng.permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tok, off, 22, 8, g, m, t=T, approx=True, delta=0.5, max_iters=6, order=order)
    for devices in ([0], [0, 0], [0, 0, 0]):
        e = native.Engine(g, m, t=T, approx=True, delta=0.5, max_iters=6, devices=devices)
        e.set_combo_order(order)
        e.compute(tok, off, 22, 8)
        assert np.array_equal(e.get_stdevs(), sd)
        if T <= 2:
            assert np.array_equal(e.get_triangle(), want)
        else:   # a sum of T fp64 terms: the reference adds them in thread-arrival order
            assert np.allclose(e.get_triangle(), want, rtol=1e-14, atol=0)
        e.close()


def test_group_deadline_on_a_late_exchange_kernel(native, monkeypatch):
    """Fail fast on the device side: engine 1's exchange stream is held by a (bounded, 3 s) spinning kernel in front
    of band 0's all-reduce (the fault_* tuning keys of tests/hooks' build of the engine), the deadline is 400 ms: finalize returns FSK_EDEVICE naming the
    band within about the deadline instead of sitting in hipStreamSynchronize, the group is dead afterwards, and
    closing it (which waits for the spin to end by itself) does not hang. Then the same job without the fault."""
    import time
    d = load_golden("f4_ep300_exact")
    combos = np.asarray(d["combos"], dtype=np.int32)
    hooks = hooks_library(native)
    e = native.Engine(d["g"], d["m"], devices=[0, 0], collective=native.COLL_P2P, deadline_ms=400, lib=hooks,
                      tuning={"fault_kind": 2, "fault_rank": 1, "fault_band": 0, "fault_ms": 3000})
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    e.accumulate(combos)           # asynchronous: nothing waits here
    t0 = time.perf_counter()
    with pytest.raises(native.FskError) as ei:
        e.finalize()
    dt = time.perf_counter() - t0
    assert ei.value.code == -4 and "band 0" in str(ei.value) and "400 ms" in str(ei.value), str(ei.value)
    assert 0.3 < dt < 2.0, dt
    with pytest.raises(native.FskError) as ei2:
        e.accumulate(combos)
    assert "dead after an earlier failure" in str(ei2.value)
    e.close()
    e = native.Engine(d["g"], d["m"], devices=[0, 0], collective=native.COLL_P2P, deadline_ms=400, lib=hooks)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_counts(), d["counts"])
    e.close()


def test_counts_digest_definition(native):
    tokens, offsets = synthetic_dna(700, 50, seed=3)
    e = native.Engine(7, 3)
    e.compute(tokens, offsets, 700, 0)
    K = e.get_counts()
    idx = np.arange(K.size, dtype=np.uint64)
    with np.errstate(over="ignore"):
        want = (int(K.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(K * (idx | np.uint64(1)))))
    assert e.counts_digest() == want
    cell = lambda r: r * (r + 1) // 2
    with np.errstate(over="ignore"):
        part = (int(K[cell(128):cell(500)].sum(dtype=np.uint64)),
                int(np.bitwise_xor.reduce(K[cell(128):cell(500)] * (idx[cell(128):cell(500)] | np.uint64(1)))))
    assert e.counts_digest(128, 500) == part
    e.close()


def test_pybind_devices_dlpack_and_lazy_test_block(native, port):
    """The drop-in class: devices=[...] (bindings.cpp:14-22 kwargs unchanged, additive), the device-resident
    DLPack hand-off to the SVM stage, and the test x test block that no getter of the reference exposes
    (fastsk.cpp:190-217) left out until something asks for it."""
    import torch
    from fastsk_amd import FastSK
    d = load_golden("f4_ep300_exact")
    tokens, offsets = d["tokens"], d["offsets"]
    X = [tokens[offsets[i]:offsets[i + 1]].tolist() for i in range(len(offsets) - 1)]
    N, ntr = len(X), d["n_train"]
    sq = tri_to_square(d["tri"], N)
    for kw in ({"skip_test_block": "lazy"}, {"devices": [0], "skip_test_block": None}, {"devices": [0, 0], "collective": "p2p", "skip_test_block": "lazy"}, {}):
        lazy = "skip_test_block" in kw      # (the default is the reference's: everything computed by compute_kernel)
        f = FastSK(g=d["g"], m=d["m"], **kw)
        f.compute_kernel(X[:ntr], X[ntr:])
        st = f.stats()
        assert st["test_block_computed"] == (not lazy)
        assert st["devices"] == kw.get("devices", [0])
        if "devices" in kw:
            assert st["collective"] == ("p2p" if len(kw["devices"]) > 1 else "rccl") and st["comm_ranks"] == len(kw["devices"])
        assert np.array_equal(np.array(f.get_train_kernel()), d["train"])
        assert np.array_equal(np.array(f.get_test_kernel()), d["test"])
        Ktr = torch.from_dlpack(f.get_train_kernel_dlpack())
        Kte = torch.from_dlpack(f.get_test_kernel_dlpack())
        assert Ktr.is_cuda and Ktr.dtype == torch.float64 and tuple(Kte.shape) == (N - ntr, ntr)
        assert np.array_equal(Ktr.cpu().numpy(), d["train"]) and np.array_equal(Kte.cpu().numpy(), d["test"])
        del f                                # the blocks outlive the engine
        assert np.array_equal(Ktr.cpu().numpy(), d["train"])
        f = FastSK(g=d["g"], m=d["m"], **kw)
        f.compute_kernel(X[:ntr], X[ntr:])
        assert np.array_equal(f.get_block(0, N, 0, ntr), sq[:, :ntr])          # no test x test cell: nothing recomputed
        assert f.stats()["test_block_computed"] == (not lazy)
        assert np.array_equal(f.get_block(ntr, N, ntr, N), sq[ntr:, ntr:])      # now they are needed
        assert f.stats()["test_block_computed"]
        assert np.array_equal(f.get_counts_np(), d["counts"])
        unused = f.get_block_dlpack(0, 5, 0, 5)   # a capsule nobody consumes frees its block itself
        del unused


def test_tokeniser_to_kernel_on_the_gpu_box(native, port):
    """SURVEY 8f-2 end to end on the box: FASTA file -> FastaUtility (reference reader semantics,
    src/fastsk/utils.py:50-96) -> tokens equal to what the reference's reader produced -> compute_kernel ->
    equal to the oracle on the same tokens; and read_packed -> compute_kernel_flat (test/run_check.py:40-46)."""
    from fastsk import FastSK, FastaUtility   # the reference's import path
    fasta = os.path.join(GOLD, "fasta")
    want = np.load(os.path.join(fasta, "expected.npz"))
    rd = FastaUtility()
    Xtr, Ytr = rd.read_data(os.path.join(fasta, "messy.train.fasta"))
    Xte, Yte = rd.read_data(os.path.join(fasta, "messy.test.fasta"))
    assert [t for x in Xtr for t in x] == want["messy.train.fasta:tokens"].tolist()
    assert [t for x in Xte for t in x] == want["messy.test.fasta:tokens"].tolist()
    assert Ytr == want["messy.train.fasta:labels"].tolist() and Yte == want["messy.test.fasta:labels"].tolist()
    g, m = 4, 2
    with pytest.raises(ValueError, match="shortest train sequence has length 0"):   # the reference exit(1)s
        FastSK(g=g, m=m).compute_kernel(Xtr, Xte)
    Xtr = [x for x in Xtr if len(x) >= g]
    tok, off = native.flatten(Xtr + Xte)
    tri, _, _ = port.compute(tok, off, len(Xtr), len(Xte), g, m, t=1)
    sq = tri_to_square(tri, len(Xtr) + len(Xte))
    f = FastSK(g=g, m=m)
    f.compute_kernel(Xtr, Xte)
    assert np.array_equal(np.array(f.get_train_kernel()), sq[:len(Xtr), :len(Xtr)])
    assert np.array_equal(np.array(f.get_test_kernel()), sq[len(Xtr):, :len(Xtr)])
    toks, offs, labels = FastaUtility().read_packed(os.path.join(fasta, "protein.train.fasta"))
    assert np.array_equal(toks, want["protein.train.fasta:tokens"]) and np.array_equal(labels, want["protein.train.fasta:labels"])
    tri, _, _ = port.compute(toks, offs, 1, 1, 5, 2, t=1)
    f = FastSK(g=5, m=2)
    f.compute_kernel_flat(toks, offs, 1)
    assert np.array_equal(f.get_train_kernel_np(), tri_to_square(tri, 2)[:1, :1])
    assert np.array_equal(f.get_test_kernel_np(), tri_to_square(tri, 2)[1:, :1])


def test_variance_mode_descriptors_two_lds_rounds(native, port, monkeypatch):
    """Variance mode (the by-slot form of k_sx_consume: one u16 / u32 triangle a slot) on owner bands of TWO LDS rounds with
    descriptors forced: N = 4500 sequences over 400 keys (runs of ~100 entries), six iterations — kernel and stdevs against the
    oracle, with and without descriptors."""
    rng = np.random.default_rng(8)
    N, g, m = 4500, 4, 2
    X = [rng.integers(1, 21, size=int(L)).astype(np.int32) for L in rng.integers(8, 12, size=N)]
    X[3][:] = 2
    tokens, offsets = native.flatten(X)
    order = rng.permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tokens, offsets, 3000, 1500, g, m, t=1, approx=True, delta=1e-9, max_iters=6, order=order)
    for desc in (1, -1):
        set_tuning_env(monkeypatch, sparse_desc=desc, sparse_desc_min=5, sparse_form=1)
        e = native.Engine(g, m, t=1, approx=True, delta=1e-9, max_iters=6, path=2)
        e.set_combo_order(order)
        e.compute(tokens, offsets, 3000, 1500)
        assert e.stats()["sparse_desc"] == (desc > 0)
        assert np.array_equal(e.get_stdevs(), sd) and np.array_equal(e.get_triangle(), want), desc
        e.close()

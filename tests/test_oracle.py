"""The oracle (oracle/fastsk_oracle.c) against the golden vectors generated from the compiled
reference, against the compiled reference itself when present, and against an independent
brute-force statement of the kernel definition. CPU only."""
import hashlib
import itertools
from collections import Counter

import numpy as np
import pytest

from conftest import golden_names, load_golden, tri_to_square, load_tokens


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def brute_counts(tokens, offsets, g, m, combos):
    """K_int[a][b] = sum_combo sum_key cnt_a[key]*cnt_b[key]; no reference code involved."""
    N = len(offsets) - 1
    k = g - m
    all_pos = list(itertools.combinations(range(g), k))
    K = np.zeros((N, N), dtype=np.uint64)
    for c in combos:
        pos = all_pos[c]
        cnts = []
        for i in range(N):
            x = tokens[offsets[i]:offsets[i + 1]]
            cnts.append(Counter(tuple(int(x[j + p]) for p in pos) for j in range(len(x) - g + 1)))
        for a in range(N):
            for b in range(a + 1):
                ca, cb = cnts[a], cnts[b]
                if len(ca) > len(cb):
                    ca, cb = cb, ca
                s = sum(v * cb.get(key, 0) for key, v in ca.items())
                K[a, b] = K[b, a] = K[a, b] + np.uint64(s)
    return K


@pytest.mark.parametrize("name", golden_names())
def test_port_matches_golden(port, name):
    d = load_golden(name)
    N = d["n_train"] + d["n_test"]
    tri, sd, iters = port.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"], d["g"], d["m"],
                                  t=d["t"], approx=bool(d["approx"]), delta=d["delta"],
                                  max_iters=d["max_iters"], skip_variance=bool(d["skip_variance"]),
                                  order=d["order"])
    assert sha(tri) == d["tri_sha256"]
    assert np.array_equal(tri, d["tri"])
    sq = tri_to_square(tri, N)
    assert np.array_equal(sq[:d["n_train"], :d["n_train"]], d["train"])
    if d["n_test"]:
        assert np.array_equal(sq[d["n_train"]:, :d["n_train"]], d["test"])
    assert np.array_equal(sd, d["stdevs"])
    if "counts" in d:
        counts, _, U = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], d["combos"], threads=2)
        assert np.array_equal(counts, d["counts"])
        assert sha(counts) == d["counts_sha256"]
        assert U >= N


def test_known_answers_from_survey():
    """data/small.* g=3 m=1 raw counts; docs demo values; sqrt(9999999) sentinel."""
    d = load_golden("f1_small_g3m1")
    sq = tri_to_square(d["counts"], 4)
    assert sq.tolist() == [[15, 9, 15, 5], [9, 13, 9, 7], [15, 9, 15, 5], [5, 7, 5, 11]]
    d = load_golden("f2_docsdemo_g3m2")
    np.testing.assert_allclose(d["train"], [[1, 0.88852332], [0.88852332, 1]], atol=1e-8)
    np.testing.assert_allclose(d["test"], [[0.74535599, 0.92717265], [1, 0.88852332]], atol=1e-8)
    d = load_golden("f3_zero_overlap")
    assert d["train"][0, 1] == 0.0
    d = load_golden("f3_train_only")
    assert d["test"].shape[0] == 0
    d = load_golden("f4_ep300_variance_T1")
    assert d["stdevs"][0] == 3162.2775020544923 and len(d["stdevs"]) == 17


@pytest.mark.parametrize("name", ["f1_small_g3m1", "f2_docsdemo_g3m2", "f3_varlen_g4m2",
                                  "f3_ragged_sigma7_g6m3", "f3_lowcomplexity_g5m2"])
def test_port_matches_bruteforce_definition(port, name):
    d = load_golden(name)
    N = d["n_train"] + d["n_test"]
    nc = port.num_combos(d["g"], d["m"])
    K = brute_counts(d["tokens"], d["offsets"], d["g"], d["m"], range(nc))
    counts, _, _ = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], np.arange(nc))
    assert np.array_equal(tri_to_square(counts, N), K)
    diag = np.diag(K).astype(np.float64)
    expect = K.astype(np.float64) / np.sqrt(diag[:, None] * diag[None, :])
    got = tri_to_square(port.normalise(counts.astype(np.float64), N), N)
    assert np.array_equal(np.tril(got, -1), np.tril(expect, -1))


def test_combo_table_is_lexicographic(port):
    for g, k in [(3, 1), (5, 2), (10, 4), (12, 4), (14, 4), (7, 7)]:
        want = list(itertools.combinations(range(g), k))
        assert port.num_combos(g, g - k) == len(want)
        for c in {0, min(1, len(want) - 1), len(want) // 2, len(want) - 1}:
            assert tuple(port.combo_positions(g, k, c)) == want[c]


def test_out_of_range_tokens_are_rank_remapped(port):
    """Equality-preserving relabelling must not change the kernel (hard part 6)."""
    d = load_golden("f3_ragged_sigma7_g6m3")
    remap = np.array([0, 1000, 7, 300000, 12, 99, 5, 65536], dtype=np.int32)
    t2 = remap[d["tokens"]]
    a, _, _ = port.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"], d["g"], d["m"], t=1)
    b, _, _ = port.compute(t2, d["offsets"], d["n_train"], d["n_test"], d["g"], d["m"], t=1)
    assert np.array_equal(a, b)


def test_sub_block_property(port):
    """K_ij depends only on sequences i, j and the combo set — the basis of the
    full-size parity check: a subset run equals the sub-block of the full run, bit for bit."""
    tokens, offsets, ntr, nte, _, _ = load_tokens("EP300")
    rng = np.random.default_rng(3)
    big = np.sort(rng.choice(ntr + nte, size=90, replace=False))
    sub = np.sort(rng.choice(90, size=25, replace=False))

    def take(idx):
        toks = np.concatenate([tokens[offsets[i]:offsets[i + 1]] for i in idx]).astype(np.int32)
        offs = np.zeros(len(idx) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([offsets[i + 1] - offsets[i] for i in idx])
        return toks, offs

    combos = np.arange(0, 210, 7)
    tb, ob = take(big)
    ts, os_ = take(big[sub])
    cb, _, _ = port.raw_counts(tb, ob, 10, 6, combos, threads=4)
    cs, _, _ = port.raw_counts(ts, os_, 10, 6, combos, threads=2)
    assert np.array_equal(tri_to_square(cb, 90)[np.ix_(sub, sub)], tri_to_square(cs, 25))


def test_port_matches_compiled_reference_fresh_inputs(port, ref):
    """Beyond the stored vectors: random ragged inputs, straight against the compiled reference."""
    rng = np.random.default_rng(11)
    for sigma, g, m, n in [(4, 8, 4, 30), (20, 7, 3, 25), (3, 6, 5, 12)]:
        X = [rng.integers(1, sigma + 1, size=int(L)).astype(np.int32) for L in rng.integers(g, 60, size=n)]
        offsets = np.zeros(n + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([len(x) for x in X])
        tokens = np.concatenate(X)
        ntr = n - 7
        tri_r, sd_r = ref.full_triangle(tokens, offsets, ntr, 7, g, m, t=3)
        tri_p, sd_p, _ = port.compute(tokens, offsets, ntr, 7, g, m, t=3)
        assert np.array_equal(tri_r, tri_p)
        nc = port.num_combos(g, m)
        order = ref.shuffle_order(42, nc)
        tri_r, sd_r = ref.full_triangle(tokens, offsets, ntr, 7, g, m, t=1, approx=True, seed=42)
        tri_p, sd_p, _ = port.compute(tokens, offsets, ntr, 7, g, m, t=1, approx=True, order=order)
        assert np.array_equal(tri_r, tri_p) and np.array_equal(sd_r, sd_p)


def test_port_matches_compiled_reference_edge_cases(port, ref):
    """The degenerate shapes the engine tests use (conftest.EDGE_CASES): one sequence, lengths == g,
    identical sequences, k = 1, nothing shared, m = 0."""
    from conftest import EDGE_CASES
    from fastsk_amd import _native
    for X, ntr, nte, g, m in EDGE_CASES:
        tokens, offsets = _native.flatten(X)
        tri_r, _ = ref.full_triangle(tokens, offsets, ntr, nte, g, m, t=1)
        tri_p, _, _ = port.compute(tokens, offsets, ntr, nte, g, m, t=1)
        assert np.array_equal(tri_r, tri_p), (X, g, m)


def test_fasta_reader_matches_reference_tokens(tmp_path):
    """fastsk_amd.utils.FastaUtility reproduces the reference reader's ids (shared vocab,
    lower-casing, first-seen order from 1) — checked on a file that exercises each rule."""
    from fastsk_amd.utils import FastaUtility
    tr = tmp_path / "a.train.fasta"
    te = tmp_path / "a.test.fasta"
    tr.write_text(">1\nACgt\n>0\n  ttGA \n>-1\nnACG\n")
    te.write_text(">0\nGNAx\n")
    rd = FastaUtility()
    Xtr, Ytr = rd.read_data(str(tr))
    Xte, Yte = rd.read_data(str(te))
    assert Xtr == [[1, 2, 3, 4], [4, 4, 3, 1], [5, 1, 2, 3]] and Ytr == [1, 0, -1]
    assert Xte == [[3, 5, 1, 6]] and Yte == [0]


def test_fast_fasta_reader_equals_slow_reader(tmp_path):
    """FastaUtility.read_packed (vectorised) == read_data (the reference's tokenisation), incl.
    the shared vocabulary across train and test files."""
    import os
    from fastsk_amd.utils import FastaUtility
    rng = np.random.default_rng(0)
    alpha = "ACGTNacgtn"

    def write(path, n):
        with open(path, "w") as f:
            for i in range(n):
                f.write(">%d\n" % int(rng.integers(0, 2)))
                f.write("".join(rng.choice(list(alpha), size=int(rng.integers(5, 40)))) + "  \n")

    files = [str(tmp_path / "a.train.fasta"), str(tmp_path / "a.test.fasta")]
    write(files[0], 30)
    write(files[1], 11)
    if os.path.isdir("/root/reference/data"):
        files += ["/root/reference/data/1.1.train.fasta", "/root/reference/data/1.1.test.fasta"]
    slow, fast = FastaUtility(), FastaUtility()
    for fpath in files:
        X, Y = slow.read_data(fpath)
        toks, offs, labels = fast.read_packed(fpath)
        assert labels.tolist() == Y
        assert offs.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in X])]).tolist()
        assert toks.tolist() == [t for x in X for t in x]
    assert str(slow._vocab) == str(fast._vocab)


def test_seed_order_is_the_references_shuffle(ref):
    """fsk_set_seed(S) stands for the order the reference draws when time(0) == S: libstdc++'s std::shuffle over
    std::default_random_engine (fastsk_kernel.cpp:31-38), restated in the engine (fsk_seed_order) — against the compiled
    reference's own std::shuffle for lengths on both sides of its two-swaps-a-draw limit (n * n <= 2^31 - 3) and seeds
    incl. the ones minstd_rand0 folds (0 and 2^31 - 1 both seed as 1)."""
    from fastsk_amd import _native
    lib = _native.library()
    for n in (0, 1, 2, 3, 4, 210, 495, 1001, 38760, 46340, 46341, 50000, 184756):
        for seed in (0, 1, 42, 777, 2 ** 31 - 2, 2 ** 31 - 1, 2 ** 31, 1759622400, 2 ** 40 + 12345):
            want = ref.shuffle_order(seed, n) if n else np.zeros(0, dtype=np.int32)
            got = lib.seed_order(seed, n)
            assert np.array_equal(got, want), (n, seed)


def test_seed_order_reproduces_every_golden_order():
    """... and the orders stored in the committed fixtures (written by the compiled reference with time() pinned to `seed`)."""
    from conftest import golden_names
    from fastsk_amd import _native
    lib = _native.library()
    seen = 0
    for name in golden_names(("f1", "f2", "f3", "f4", "f5", "f6", "f7", "f8")):
        d = load_golden(name)
        if "order" in d and "seed" in d and len(d["order"]):
            assert np.array_equal(lib.seed_order(d["seed"], len(d["order"])), d["order"]), name
            seen += 1
    assert seen >= 5

"""Kernel LOGIC on the CPU: the engine's HIP source compiled against tests/emu/hip_emu.h
(fibers + wave64 collectives) must reproduce the golden vectors bit for bit through the C ABI.
This does not replace the GPU parity tests (tests/test_gpu_parity.py); it catches indexing,
barrier and ranking mistakes without spending GPU minutes."""
import os
import sys

import numpy as np
import pytest

from conftest import golden_names, load_golden, tri_to_square, set_tuning_env, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


@pytest.fixture(scope="session")
def emu_lib():
    import build_emu
    from fastsk_amd import _native
    return _native.Library(build_emu.build())


def run_case(emu_lib, d, path):
    from fastsk_amd import _native
    e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"],
                       max_iters=d["max_iters"], skip_variance=bool(d["skip_variance"]), path=path, lib=emu_lib)
    if d["approx"]:
        e.set_combo_order(d["order"])
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    return e


def test_emu_dense_split_launch_through_staging_blocks(emu_lib):
    """The dense dataflow's split tile launch with its sums in 32-bit staging blocks (k_dense_tile_small + k_dense_widen, tuning
    dense_small=1, its own translation unit) instead of 64-bit atomics: goldens incl. counts above 15 (the hi plane) and a
    rare symbol (the compact kernel), three splits a tile."""
    from fastsk_amd import _native
    for name in ("f3_lowcomplexity_g5m2", "f3_ragged_sigma7_g6m3", "f2_docsdemo_g3m2"):
        d = load_golden(name)
        for tun in ({"dense_small": 1, "tile_splits": 3}, {"dense_small": 1, "tile_splits": 2, "compact": 1}):
            e = _native.Engine(d["g"], d["m"], path=1, lib=emu_lib, tuning=tun)
            e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
            assert np.array_equal(e.get_counts(), d["counts"]), (name, tun)
            e.close()


def test_emu_wave_primitives_stand_ins(emu_lib):
    """The emulator's shuffle-loop stand-ins for the wave primitives meet the same definitions the GPU's inline asm is checked
    against (tests/test_gpu_parity.py::test_wave_primitives)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_parity import _wave_primitives_check
    _wave_primitives_check(emu_lib)


def test_emu_seed_is_the_references_seed(emu_lib):
    """approx mode with fsk_set_seed(S) and no injected order == the reference run with time(0) == S (fastsk_kernel.cpp:31-38):
    variance mode (stdevs included) and skip-variance, both dataflows; the older splitmix order stays behind a tuning key."""
    from fastsk_amd import _native
    for name in ("f4_ep300_variance_T1", "f6_prot219_skipvar16"):
        d = load_golden(name)
        for path in (1, 2):
            if path == 1 and name.startswith("f6"):
                continue  # (protein: no dense dataflow)
            e = _native.Engine(d["g"], d["m"], t=d["t"], approx=True, delta=d["delta"], max_iters=d["max_iters"],
                               skip_variance=bool(d["skip_variance"]), path=path, lib=emu_lib)
            e.set_seed(d["seed"])
            e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
            assert np.array_equal(e.get_triangle(), d["tri"]), (name, path)
            if not d["skip_variance"]:
                assert np.array_equal(e.get_stdevs(), d["stdevs"])
            e.close()
    d = load_golden("f6_prot219_skipvar16")
    e = _native.Engine(d["g"], d["m"], t=d["t"], approx=True, max_iters=d["max_iters"], skip_variance=True, lib=emu_lib,
                       tuning={"seed_splitmix": 1})
    e.set_seed(d["seed"])
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert not np.array_equal(e.get_triangle(), d["tri"])  # (another sample of 16 combos)
    e.close()


SMALL = [n for n in golden_names() if n.split("_")[0] in ("f1", "f2", "f3")]


@pytest.mark.parametrize("name", ["f3_ragged_sigma7_g6m3", "f3_train_only", "f4_ep300_exact"])
def test_emu_save_kernel_against_the_reference_file(emu_lib, tmp_path, name):
    """fsk_save_kernel writes, byte for byte, what the reference's FastSK::save_kernel (fastsk.cpp:223-237) wrote
    for the same input (tests/golden/save_kernel.npz, tests/make_golden_save_kernel.py)."""
    import os
    from conftest import GOLD
    z = np.load(os.path.join(GOLD, "save_kernel.npz"))
    d = load_golden(name)
    e = run_case(emu_lib, d, 0)
    p = tmp_path / "k.txt"
    e.save_kernel(str(p))
    assert p.read_bytes() == z[name].tobytes()


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("name", SMALL)
def test_emu_small_cases_both_paths(emu_lib, name, path):
    d = load_golden(name)
    e = run_case(emu_lib, d, path)
    N = d["n_train"] + d["n_test"]
    assert np.array_equal(e.get_counts(), d["counts"])
    assert np.array_equal(e.get_triangle(), d["tri"])
    assert np.array_equal(e.get_train(), d["train"])
    if d["n_test"]:
        assert np.array_equal(e.get_test(), d["test"])
    sq = tri_to_square(d["tri"], N)
    assert np.array_equal(e.get_block(0, N, 0, N), sq)
    st = e.stats()
    assert st["path_used"] == path and st["combos_done"] == len(d["combos"])
    e.close()


def test_emu_ep300_slice_exact_dense(emu_lib):
    d = load_golden("f4_ep300_exact")
    e = run_case(emu_lib, d, 0)
    assert e.stats()["path_used"] == 1  # DNA, 4^4 keys -> dense
    assert np.array_equal(e.get_counts(), d["counts"])
    assert np.array_equal(e.get_triangle(), d["tri"])
    assert np.array_equal(e.get_test(), d["test"])


@pytest.mark.parametrize("name,path,ncombo", [("f4_ep300_exact", 2, 9), ("f5_prot11_exact", 0, 14),
                                               ("f6_prot219_exact", 0, 40), ("f6_ep47848_slice_exact", 1, 210),
                                               ("f6_ep47848_slice_exact", 2, 6)])
def test_emu_staged_accumulate_vs_port(emu_lib, port, name, path, ncombo):
    """load -> accumulate (two calls) -> finalize, against the oracle on the same combos."""
    from fastsk_amd import _native
    d = load_golden(name)
    nc = port.num_combos(d["g"], d["m"])
    combos = np.linspace(0, nc - 1, ncombo).astype(np.int32)
    combos = np.unique(combos)
    want, _, U = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], combos)
    e = _native.Engine(d["g"], d["m"], path=path, lib=emu_lib, profile=True)  # (profile: the dense dataflow counts U too)
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    half = len(combos) // 2
    e.accumulate(combos[:half])
    e.accumulate(combos[half:])
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    N = d["n_train"] + d["n_test"]
    assert np.array_equal(e.get_triangle(), port.normalise(want.astype(np.float64), N))
    st = e.stats()
    # sparse: the engine issues exactly the reference's `+=` count; dense: k_dense_distinct counts the same U from
    # the count panels (what bench.py's algorithmic bytes and useful_update_frac are built from)
    assert st["path_used"] in (1, 2) and st["cell_updates"] == U
    blk = e.get_counts_block(3, 17, 1, 9)
    assert np.array_equal(blk, tri_to_square(want, N)[3:17, 1:9])


@pytest.mark.parametrize("name", ["f4_ep300_skipvar_T1", "f4_ep300_skipvar_T3", "f4_ep300_variance_T1",
                                  "f4_ep300_variance_T1_conv", "f5_prot11_variance_T1_it9", "f6_prot219_skipvar16"])
def test_emu_approx_modes(emu_lib, name):
    d = load_golden(name)
    e = run_case(emu_lib, d, 0)
    assert np.array_equal(e.get_triangle(), d["tri"])
    assert np.array_equal(e.get_stdevs(), d["stdevs"])
    assert np.array_equal(e.get_train(), d["train"])
    assert np.array_equal(e.get_test(), d["test"])
    if "counts" in d:
        assert np.array_equal(e.get_counts(), d["counts"])


@pytest.mark.parametrize("path", [0, 2, -1])
def test_emu_variance_mode_stops_anywhere(emu_lib, port, path, monkeypatch):
    """Variance mode runs ahead of its stop test (batches of 4 iterations, two in flight): whatever
    the chain count, max_iters and delta, the stop must land on the reference's iteration and the
    state of the dropped iterations must not leak into the result. (Sparse dataflow: the Welford
    state is written once per batch; a stop inside a batch runs the accepted prefix again. So it is on
    the dense dataflow, where every iteration's tile launch stores into a triangle of its own; path -1 =
    the dense dataflow with that switched off: zero fill and one Welford kernel per iteration.)"""
    from fastsk_amd import _native
    slots_off = path == -1
    if path == -1:
        set_tuning_env(monkeypatch, variance_dense_slots="0")
        path = 1
    elif path == 0:
        path = 1
    rng = np.random.default_rng(5)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(12, 40, size=30)]
    tok, off = _native.flatten(X)
    g, m = 7, 3
    order = rng.permutation(port.num_combos(g, m)).astype(np.int32)
    lengths = set()
    full = path == 1 and not slots_off
    for T in (1, 2, 3) if full else (1, 2):
        for max_iters in (-1, 1, 2, 4, 5, 6, 9) if full else (-1, 3, 6):
            for delta in (0.025, 0.5, 1.0, 3.0) if full else (0.2, 1.0, 3.0):
                want, sd, _ = port.compute(tok, off, 22, 8, g, m, t=T, approx=True, delta=delta, max_iters=max_iters, order=order)
                e = _native.Engine(g, m, t=T, approx=True, delta=delta, max_iters=max_iters, path=path, lib=emu_lib)
                e.set_combo_order(order)
                e.compute(tok, off, 22, 8)
                assert np.array_equal(e.get_stdevs(), sd), (T, max_iters, delta)
                assert np.array_equal(e.get_triangle(), want), (T, max_iters, delta)
                lengths.add(len(sd))
                e.close()
    assert len(lengths) >= (6 if full else 4)  # stops landed at many different places inside the batches


def poly_a_sequences(n=24, L=300, seed=11):
    """DNA with a few 290-long runs of one symbol: a k-mer count above 255 in one sequence, which the
    dense dataflow's u8 panels cannot hold (the batch goes to the general dataflow)."""
    rng = np.random.default_rng(seed)
    X = [rng.integers(1, 5, size=int(l)).astype(np.int32) for l in rng.integers(L - 20, L + 20, size=n)]
    for i in (2, 9, 17):
        X[i][5:295] = 1
    return X


@pytest.mark.parametrize("form", ["dense_slots", "dense_fill", "sparse", "sparse_u32"])
def test_emu_variance_mode_count_above_255(emu_lib, port, form, monkeypatch):
    """Variance mode on the dense dataflow keeps one triangle per iteration in flight and lets the
    tile launch STORE into it; an iteration whose counts overflow the panels is diverted to the
    general dataflow, which adds — into a triangle that must have been zeroed first."""
    from fastsk_amd import _native
    if form == "dense_fill":
        set_tuning_env(monkeypatch, variance_dense_slots="0")
    tok, off = _native.flatten(poly_a_sequences())
    g, m, T = 5, 2, 2
    order = np.random.default_rng(3).permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tok, off, 16, 8, g, m, t=T, approx=True, delta=0.025, max_iters=-1, order=order)
    if form == "sparse_u32":
        set_tuning_env(monkeypatch, var_slots16="0")
    e = _native.Engine(g, m, t=T, approx=True, path=2 if form.startswith("sparse") else 1, lib=emu_lib)
    e.set_combo_order(order)
    e.compute(tok, off, 16, 8)
    assert e.stats()["max_windows"] > 255
    assert np.array_equal(e.get_stdevs(), sd)
    assert np.array_equal(e.get_triangle(), want)
    # (sparse dataflow: the slot triangles are u16 until a sum does not fit — two 286-window runs of one k-mer meet in a
    # cell of 81,796 — and that batch is redone with u32 triangles, which the sequences then keep)
    if form == "sparse":
        assert e.stats()["batches_redone"] == 1
    if form == "sparse_u32":
        assert e.stats()["batches_redone"] == 0
    e.close()


def test_emu_reset_then_storing_launch(emu_lib, port, monkeypatch):
    """fsk_reset_counts leaves the zeros to the next tile launch when that launch can STORE its
    sums (dense dataflow, one workgroup per tile, rows starting at the reset range's lower edge);
    every other consumer of K gets the zeros filled in first."""
    from fastsk_amd import _native
    rng = np.random.default_rng(123)
    N = 520
    X = rng.integers(1, 5, size=(N, 48), dtype=np.int32)
    X[::11, 5:40] = 2   # rows with counts above 15
    tokens, offsets = _native.flatten(X)
    ca, cb = np.arange(0, 70, 6, dtype=np.int32), np.arange(1, 70, 8, dtype=np.int32)
    wa, _, _ = port.raw_counts(tokens, offsets, 8, 4, ca, threads=4)
    wb, _, _ = port.raw_counts(tokens, offsets, 8, 4, cb, threads=4)
    cell = lambda r: r * (r + 1) // 2
    for splits in ("1", "0"):   # "1": one workgroup per tile -> the storing launch; "0": automatic splits -> zero fill + atomics
        set_tuning_env(monkeypatch, tile_splits=splits)
        e = _native.Engine(8, 4, path=1, lib=emu_lib)
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


def test_emu_bound_counts_buffer_and_reset(emu_lib, port):
    """fsk_bind_counts: the integer triangle lives in caller-provided memory (on the GPU: a torch
    tensor that RCCL all-reduces). Under emulation device memory is host memory."""
    from fastsk_amd import _native
    d = load_golden("f3_ragged_sigma7_g6m3")
    N = d["n_train"] + d["n_test"]
    buf = np.full(N * (N + 1) // 2, 7, dtype=np.uint64)
    e = _native.Engine(d["g"], d["m"], lib=emu_lib)
    e.bind_counts(buf.ctypes.data, buf.size, keepalive=buf)
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])  # zeroes the bound buffer
    assert e.counts_device_ptr() == buf.ctypes.data
    e.accumulate(d["combos"][:5])
    e.synchronize()
    want5, _, _ = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], d["combos"][:5])
    assert np.array_equal(buf, want5)
    e.reset_counts()
    e.accumulate(d["combos"])
    e.finalize()
    assert np.array_equal(buf, d["counts"])
    assert np.array_equal(e.get_triangle(), d["tri"])


def test_emu_error_convention(emu_lib):
    from fastsk_amd import _native
    with pytest.raises(_native.FskError) as ei:
        _native.Engine(3, 3, lib=emu_lib)
    assert ei.value.code == -1
    e = _native.Engine(6, 2, lib=emu_lib)
    tok, off = _native.flatten([[1, 2, 3, 4, 1, 2, 3], [1, 2, 3, 4, 1]])
    with pytest.raises(_native.FskError) as ei:  # reference: printf + exit(1), fastsk.cpp:53-58
        e.compute(tok, off, 1, 1)
    assert ei.value.code == -2 and "shortest test sequence has length 5" in str(ei.value)
    with pytest.raises(_native.FskError) as ei:
        e.get_train()
    assert ei.value.code == -3
    tok, off = _native.flatten([[1, 2, 3, 4, 1, 2, 3], [1, 2, 3, 4, 1, 4]])
    e.load_sequences(tok, off, 2, 0)
    with pytest.raises(_native.FskError) as ei:
        e.accumulate([15])
    assert ei.value.code == -1
    e.accumulate(np.arange(15))
    e.finalize()
    assert e.get_test().shape == (0, 2)
    assert e.get_train()[0, 0] == 1.0
    # the chain entry points belong to the variance mode
    with pytest.raises(_native.FskError) as ei:
        e.run_chains(0, 1)
    assert ei.value.code == -3 and "variance mode" in str(ei.value)
    buf = np.zeros(3, dtype=np.float64)
    with pytest.raises(_native.FskError) as ei:
        e.get_kernel_sum_device(buf.ctypes.data)
    assert ei.value.code == -3
    v = _native.Engine(6, 2, t=2, approx=True, max_iters=3, lib=emu_lib)
    with pytest.raises(_native.FskError) as ei:
        v.run_chains(0, 1)           # nothing loaded
    assert ei.value.code == -3
    v.load_sequences(tok, off, 2, 0)
    with pytest.raises(_native.FskError) as ei:
        v.run_chains(-1, 1)
    assert ei.value.code == -1
    with pytest.raises(_native.FskError) as ei:
        v.set_kernel_sum_device(buf.ctypes.data)   # no chains have run yet
    assert ei.value.code == -3
    v.run_chains(1, 2)               # chain 1 only: no stdevs on this engine
    v.get_kernel_sum_device(buf.ctypes.data)
    v.set_kernel_sum_device(buf.ctypes.data)
    v.finalize()
    assert len(v.get_stdevs()) == 0 and v.get_train()[1, 1] == 1.0


def test_emu_edge_cases(emu_lib, port):
    from fastsk_amd import _native
    from conftest import EDGE_CASES
    for X, ntr, nte, g, m in EDGE_CASES:
        tokens, offsets = _native.flatten(X)
        want, _, _ = port.compute(tokens, offsets, ntr, nte, g, m, t=1)
        for path in (0, 1, 2):
            if path == 1 and len(np.unique(tokens)) ** (g - m) > 16384:
                continue
            e = _native.Engine(g, m, path=path, lib=emu_lib)
            e.compute(tokens, offsets, ntr, nte)
            assert np.array_equal(e.get_triangle(), want), (X, g, m, path)
            assert e.get_test().shape == (nte, ntr)
            e.close()
    with pytest.raises(_native.FskError) as ei:
        _native.Engine(6, 2, lib=emu_lib).compute(np.zeros(0, np.int32), np.zeros(1, np.int64), 0, 0)
    assert ei.value.code == -1


def test_emu_skip_test_block(emu_lib, port):
    from fastsk_amd import _native
    rng = np.random.default_rng(4)
    N, ntr = 400, 100   # first all-test tile column = 1: cells with column >= 128 off the diagonal tiles
    X = rng.integers(1, 5, size=(N, 24), dtype=np.int32)
    tokens, offsets = X.reshape(-1), np.arange(N + 1, dtype=np.int64) * 24
    want, _, _ = port.compute(tokens, offsets, ntr, N - ntr, 7, 4, t=1)
    sq = tri_to_square(want, N)
    e = _native.Engine(7, 4, path=1, lib=emu_lib, skip_test_block=True)
    e.compute(tokens, offsets, ntr, N - ntr)
    assert np.array_equal(e.get_train(), sq[:ntr, :ntr])
    assert np.array_equal(e.get_test(), sq[ntr:, :ntr])
    got = tri_to_square(e.get_counts(), N)
    assert not got[300, 130] and not got[399, 255] and got[300, 290] and got[399, 399]
    e.close()
    # sparse dataflow: a test row pairs only with the train entries of its runs and with itself —
    # exactly the test x test cells off the diagonal stay zero (also in row bands, also with atomics)
    some = np.arange(0, 35, 2, dtype=np.int32)
    raw, _, _ = port.raw_counts(tokens, offsets, 7, 4, some, threads=4)
    ref = tri_to_square(raw, N).astype(np.int64)
    i, j = np.tril_indices(N, -1)
    tt = j >= ntr
    for bands in (False, True):
        e = _native.Engine(7, 4, path=2, lib=emu_lib, skip_test_block=True)
        e.load_sequences(tokens, offsets, ntr, N - ntr)
        if bands:
            for lo, hi in ((0, 128), (128, 384), (384, N)):
                e.accumulate_rows(some, lo, hi)
        else:
            e.accumulate(some)
        e.finalize()
        got = tri_to_square(e.get_counts(), N).astype(np.int64)
        assert np.array_equal(np.diag(got), np.diag(ref))
        assert not got[i[tt], j[tt]].any() and ref[i[tt], j[tt]].any()
        assert np.array_equal(got[i[~tt], j[~tt]], ref[i[~tt], j[~tt]])
        e.close()


def test_emu_relabelled_and_wide_alphabets(emu_lib, port):
    from fastsk_amd import _native
    d = load_golden("f3_ragged_sigma7_g6m3")
    remap = np.array([0, 1000, 7, 300000, -12, 99, 5, 2 ** 30], dtype=np.int32)
    e = _native.Engine(d["g"], d["m"], lib=emu_lib)
    e.compute(remap[d["tokens"]], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_triangle(), d["tri"])
    # 20-symbol alphabet, k = 9: 20^9 keys -> 64-bit composite keys on the sparse path
    rng = np.random.default_rng(2)
    X = [rng.integers(1, 21, size=int(L)) for L in rng.integers(12, 50, size=9)]
    for x in X[:4]:
        x[3:15] = X[5][3:15]  # shared 12-mers so that off-diagonals are non-trivial
    tok, off = _native.flatten(X)
    combos = np.arange(0, port.num_combos(12, 3), 5, dtype=np.int32)  # (a subset: the emulator pays per workgroup)
    e = _native.Engine(12, 3, lib=emu_lib)
    e.load_sequences(tok, off, 6, 3)
    e.accumulate(combos)
    e.finalize()
    want, _, _ = port.raw_counts(tok, off, 12, 3, combos)
    assert np.array_equal(e.get_counts(), want) and want[1] > 0
    assert np.array_equal(e.get_triangle(), port.normalise(want.astype(np.float64), 9))
    assert e.stats()["path_used"] == 2
    e.close()
    # k = 14 over 20 symbols: 61 k-mer bits + 5 sequence bits = 66 -> 128-bit sort records
    X = [rng.integers(1, 21, size=int(L)) for L in rng.integers(17, 40, size=20)]
    for x in X[:9]:
        x[1:17] = X[12][1:17]
    X[3][5] = X[3][5] % 20 + 1  # a near copy: shares some gapped 14-mers only
    tok, off = _native.flatten(X)
    combos = np.arange(0, port.num_combos(16, 2), 4, dtype=np.int32)
    e = _native.Engine(16, 2, lib=emu_lib)
    e.load_sequences(tok, off, 14, 6)
    e.accumulate(combos)
    e.finalize()
    want, _, _ = port.raw_counts(tok, off, 16, 2, combos)
    assert np.array_equal(e.get_counts(), want)
    assert np.array_equal(e.get_triangle(), port.normalise(want.astype(np.float64), 20))
    assert e.stats()["path_used"] == 2 and e.stats()["key_space"] == 20 ** 14
    e.close()
    # k = 6 over 20 symbols: 26 k-mer bits + 5 sequence bits still travel in 32-bit records, but the k-mer no longer fits
    # the 24-bit multiply-add of the window extraction
    X = [rng.integers(1, 21, size=int(L)) for L in rng.integers(9, 40, size=20)]
    for x in X[:8]:
        x[0:9] = X[11][0:9]
    tok, off = _native.flatten(X)
    combos = np.arange(0, port.num_combos(8, 2), 3, dtype=np.int32)
    e = _native.Engine(8, 2, path=2, lib=emu_lib)
    e.load_sequences(tok, off, 13, 7)
    e.accumulate(combos)
    e.finalize()
    want, _, _ = port.raw_counts(tok, off, 8, 2, combos)
    assert np.array_equal(e.get_counts(), want) and want[1] > 0
    assert e.stats()["key_space"] == 20 ** 6
    e.close()
    # g = 17 symbols of 8 bits pass the 128-bit window array: the records' symbols are gathered from the sequences
    X = [rng.integers(1, 21, size=int(L)) for L in rng.integers(18, 40, size=12)]
    for x in X[:5]:
        x[0:18] = X[7][0:18]
    tok, off = _native.flatten(X)
    combos = np.arange(0, port.num_combos(17, 13), 90, dtype=np.int32)
    e = _native.Engine(17, 13, path=2, lib=emu_lib)
    e.load_sequences(tok, off, 8, 4)
    e.accumulate(combos)
    e.finalize()
    want, _, _ = port.raw_counts(tok, off, 17, 13, combos)
    assert np.array_equal(e.get_counts(), want) and want[1] > 0
    e.close()


def test_emu_mixed_4bit_and_8bit_panels(emu_lib, port):
    """Only the (panel, combo) pairs holding a count above 15 leave the 4-bit/dot8 form: mix
    random DNA with a few low-complexity sequences in different panels and combos."""
    from fastsk_amd import _native
    rng = np.random.default_rng(9)
    X = [rng.integers(1, 5, size=70).astype(np.int32) for _ in range(200)]
    X[5][:40] = 1                      # poly-A: counts of 'aaaa' far above 15 in panel 0
    X[130][10:50] = [2, 3] * 20        # dinucleotide repeat in panel 2
    X[199][:] = 4                      # whole sequence one letter, last (partial) panel
    tok, off = _native.flatten(X)
    combos = np.arange(0, 70, 3, dtype=np.int32)
    want, _, _ = port.raw_counts(tok, off, 8, 4, combos, threads=4)
    e = _native.Engine(8, 4, path=1, lib=emu_lib)
    e.load_sequences(tok, off, 150, 50)
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    assert e.stats()["u4_tile_launches"] == 1


@pytest.mark.parametrize("path", [1, 2])
def test_emu_row_bands(emu_lib, port, path):
    """fsk_accumulate_rows: bands of 128 rows add up to the full partial kernels; the count
    panels are computed once per pass (dense) and each finished band is already final."""
    from fastsk_amd import _native
    rng = np.random.default_rng(4)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(20, 50, size=300)]
    tok, off = _native.flatten(X)
    combos = np.array([0, 5, 11, 33, 69], dtype=np.int32)
    want, _, U = port.raw_counts(tok, off, 8, 4, combos, threads=4)
    sq = tri_to_square(want, 300)
    e = _native.Engine(8, 4, path=path, lib=emu_lib)
    e.load_sequences(tok, off, 300, 0)
    for lo, hi in [(0, 128), (128, 256), (256, 300)]:
        e.accumulate_rows(combos, lo, hi)
        e.synchronize()
        got = e.get_counts_block(lo, hi, 0, 300)
        assert np.array_equal(np.tril(got, lo), np.tril(sq[lo:hi], lo))  # this band is complete
    assert np.array_equal(e.get_counts(), want)
    st = e.stats()
    assert st["combos_done"] == len(combos)
    if path == 1:
        assert st["count_launches"] == 1 and st["n_tile_launches"] == 3
    else:
        assert st["cell_updates"] == U
    with pytest.raises(_native.FskError):
        e.accumulate_rows(combos, 64, 128)
    # a new pass starting at row 0 recounts (no stale cache across passes)
    e.reset_counts()
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    if path == 1:
        assert e.stats()["count_launches"] == 2


@pytest.mark.parametrize("chunk", ["7", "64"])
def test_emu_dense_count_chunked_staging(emu_lib, port, monkeypatch, chunk):
    """Sequences too long for one LDS staging pass are counted chunk by chunk (forced small here)."""
    from fastsk_amd import _native
    set_tuning_env(monkeypatch, dense_chunk=chunk)
    rng = np.random.default_rng(12)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(9, 200, size=70)]
    tok, off = _native.flatten(X)
    combos = np.array([0, 17, 125], dtype=np.int32)
    want, _, _ = port.raw_counts(tok, off, 9, 5, combos, threads=2)
    e = _native.Engine(9, 5, path=1, lib=emu_lib)
    e.load_sequences(tok, off, 50, 20)
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)


@pytest.mark.parametrize("sigma,g,m", [(4, 8, 2), (5, 7, 2), (3, 9, 2), (2, 12, 2), (2, 14, 1)])
def test_emu_dense_large_key_space_sweeps(emu_lib, port, sigma, g, m):
    """Key spaces above one LDS histogram (640 keys) are counted in several sweeps: DNA k = 6
    (4096 keys), 5 symbols k = 5 (3125), 3 symbols k = 7 (2187)."""
    from fastsk_amd import _native
    rng = np.random.default_rng(sigma + g)
    X = [rng.integers(1, sigma + 1, size=int(L)).astype(np.int32) for L in rng.integers(g, 90, size=70)]
    X[3][:] = 1  # low complexity: exercises the u8 fallback in a large key space
    tok, off = _native.flatten(X)
    nc = port.num_combos(g, m)
    combos = np.array([0, nc // 2, nc - 1], dtype=np.int32)
    want, _, _ = port.raw_counts(tok, off, g, m, combos, threads=2)
    e = _native.Engine(g, m, path=1, lib=emu_lib)
    e.load_sequences(tok, off, 40, 30)
    e.accumulate(combos)
    e.finalize()
    assert e.stats()["key_space"] == sigma ** (g - m)
    assert np.array_equal(e.get_counts(), want)


def test_emu_extract_four_slots_per_workgroup(emu_lib, port, monkeypatch):
    """Large sparse launches take a tile of windows through four slots per workgroup (tuning extract_slots forces it
    here); a batch whose slot count is not a multiple of four leaves the last workgroup row short."""
    from fastsk_amd import _native
    set_tuning_env(monkeypatch, extract_slots="4")
    for name in ("f5_prot11_exact", "f3_ragged_sigma7_g6m3"):
        d = load_golden(name)
        nc = port.num_combos(d["g"], d["m"])
        combos = np.arange(0, nc, max(1, nc // 13), dtype=np.int32)[:13]
        want, _, _ = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], combos)
        e = _native.Engine(d["g"], d["m"], path=2, lib=emu_lib)
        e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
        e.accumulate(combos[:6])
        e.accumulate(combos[6:])
        e.finalize()
        assert np.array_equal(e.get_counts(), want), name
        e.close()


@pytest.mark.parametrize("global_pairs", ["0", "1"])
def test_emu_sparse_pair_accumulation_variants(emu_lib, port, monkeypatch, global_pairs):
    """Sparse dataflow: owner-slice LDS accumulation (default when a row band of K fits in LDS)
    and direct per-pair global atomics (large N) issue the same updates."""
    from fastsk_amd import _native
    set_tuning_env(monkeypatch, sparse_global=global_pairs)
    d = load_golden("f5_prot11_exact")
    combos = np.arange(0, 210, 30, dtype=np.int32)
    want, _, U = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], combos)
    e = _native.Engine(d["g"], d["m"], path=2, lib=emu_lib)
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    assert e.stats()["cell_updates"] == U


def test_emu_emit_quarter_tile_passes(emu_lib, port):
    """Sparse dataflow, k_sx_emit: 16 long sequences over 4^5 keys give every k-mer run ~16 entries of up to
    16 partners — short entries only, ~17 k update words per 2048-entry tile, more than the 12,288 LDS
    slots: the tile is binned a quarter at a time (every other input bins a tile in one pass)."""
    from fastsk_amd import _native
    rng = np.random.default_rng(44)
    X = rng.integers(1, 5, size=(16, 2500), dtype=np.int32)
    tok, off = _native.flatten(X)
    g, m = 8, 3
    combos = np.array([0, 17, 55], dtype=np.int32)
    want, _, U = port.raw_counts(tok, off, g, m, combos)
    e = _native.Engine(g, m, path=2, lib=emu_lib)
    e.load_sequences(tok, off, 12, 4)
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    assert e.stats()["cell_updates"] == U and U > 3 * 12288 * 8
    e.close()


def test_emu_segment_scan_in_chunks(emu_lib, port, monkeypatch):
    """Sparse dataflow: the tile records of a batch (entries per tile, last run start) are scanned by
    one workgroup, or — batches of many tiles — as chunk totals, a scan over the chunks and the chunks
    with their carries; forced here on a small batch, with and without skip_test_block."""
    from fastsk_amd import _native
    set_tuning_env(monkeypatch, seg_scan_chunked="1")
    d = load_golden("f5_prot11_exact")
    combos = np.arange(0, 210, 40, dtype=np.int32)
    want, _, U = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], combos)
    N, ntr = d["n_train"] + d["n_test"], d["n_train"]
    for skip in (False, True):
        e = _native.Engine(d["g"], d["m"], path=2, skip_test_block=skip, lib=emu_lib)
        e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
        e.accumulate(combos)
        e.finalize()
        ref = want.copy()
        if skip:
            i, j = np.tril_indices(N)
            ref[(j >= ntr) & (i != j)] = 0
        assert np.array_equal(e.get_counts(), ref), skip
        e.close()


@pytest.mark.parametrize("cap", [None, "100"])
def test_emu_sparse_batches_enqueued_ahead_of_their_size(emu_lib, port, monkeypatch, cap):
    """Sparse dataflow: after the first batch of a set of sequences, batches are enqueued without
    waiting for their update-word count; one that does not fit the stream buffer leaves K alone and
    is redone sized exactly (tuning guard_cap makes every such batch overflow)."""
    from fastsk_amd import _native
    if cap:
        set_tuning_env(monkeypatch, guard_cap=cap)
    d = load_golden("f5_prot11_exact")
    combos = np.arange(0, 210, 15, dtype=np.int32)
    want, _, U = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], combos)
    e = _native.Engine(d["g"], d["m"], path=2, lib=emu_lib)
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    for part in np.array_split(combos, 4):
        e.accumulate(part)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    st = e.stats()
    assert st["cell_updates"] == U
    assert st["batches_redone"] == (3 if cap else 0)
    e.close()


@pytest.mark.parametrize("env", [{}, {"sparse_exact_lanes": "2"}, {"sparse_exact_lanes": "1"}, {"guard_cap": "100"}])
def test_emu_sparse_exact_accumulate_in_two_lanes(emu_lib, port, monkeypatch, env):
    """Sparse dataflow: the batches of ONE exact accumulate alternate between two lanes (scratch + stream), their consume
    passes ordered by events; the same counts and U as on one stream, also when every guarded batch is redone."""
    from fastsk_amd import _native
    set_tuning_env(monkeypatch, **env)
    d = load_golden("f5_prot11_exact")
    nfeat = int(sum(max(0, int(b) - int(a) - d["g"] + 1) for a, b in zip(d["offsets"][:-1], d["offsets"][1:])))
    set_tuning_env(monkeypatch, sparse_batch_records=str(3 * nfeat))  # (three combos a batch)
    combos = np.arange(0, 210, 9, dtype=np.int32)
    want, _, U = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], combos)
    e = _native.Engine(d["g"], d["m"], path=2, lib=emu_lib)
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    first = e.stats()["launches"]
    e.accumulate(combos[:4])   # (sizes the streams: two batches)
    e.accumulate(combos[4:])   # (seven batches, none of which waits for its size)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    st = e.stats()
    assert st["cell_updates"] == U and st["combos_done"] == combos.size
    assert st["launches"] - first > 9 * 15
    if "guard_cap" in env:
        assert st["batches_redone"] >= 7
    e.close()


@pytest.mark.parametrize("hint", [None, "0"])
def test_emu_sparse_words_per_record_hint_across_loads(emu_lib, port, monkeypatch, hint):
    """Sparse dataflow: a second set of sequences of the same shape (sequences, windows, alphabet, longest) starts from the
    words per record of the first as a hint — its first batch is enqueued under a guard instead of waited for —, and a hint
    that is far too low (low-complexity sequences after random ones) only costs a redone batch."""
    from fastsk_amd import _native
    if hint is not None:
        set_tuning_env(monkeypatch, sparse_hint=hint)
    rng = np.random.default_rng(5)
    N, L, g, m = 90, 50, 7, 3
    A = rng.integers(1, 5, size=(N, L), dtype=np.int32)
    B = rng.integers(1, 3, size=(N, L), dtype=np.int32)   # two symbols: 16 k-mers, runs through nearly every sequence —
    B[0, :4] = [1, 2, 3, 4]                                 # twice the update words per record (the alphabet stays the same)
    combos = np.arange(0, port.num_combos(g, m), 3, dtype=np.int32)
    e = _native.Engine(g, m, path=2, lib=emu_lib)
    redone = []
    for X in (A, B, B):
        tok, off = _native.flatten(X)
        want, _, U = port.raw_counts(tok, off, g, m, combos)
        before = e.stats()["cell_updates"]
        e.load_sequences(tok, off, 60, 30)
        e.accumulate(combos)
        e.finalize()
        assert np.array_equal(e.get_counts(), want)
        redone.append(e.stats()["batches_redone"])
    # (A sized exactly; B's first batch under A's hint overflows and is redone unless hints are off; B again fits its own hint)
    assert redone == ([0, 1, 1] if hint is None else [0, 0, 0])
    e.close()


@pytest.mark.parametrize("max_words", ["3000", "40000"])
def test_emu_sparse_batches_sized_by_words_per_record(emu_lib, port, monkeypatch, max_words):
    """Sparse dataflow: once a batch of a set of sequences has been counted, later batches take as many
    combos as keep their update words (judged by the most words per record seen) inside one stream, so
    that they do not fall back to atomics on K; a stream limit far below a call's words makes every call
    after the first split into several batches."""
    from fastsk_amd import _native
    d = load_golden("f5_prot11_exact")
    combos = np.arange(0, 210, 10, dtype=np.int32)
    want, _, U = port.raw_counts(d["tokens"], d["offsets"], d["g"], d["m"], combos)
    launches = {}
    for limit in (None, max_words):
        if limit:
            set_tuning_env(monkeypatch, list_max_words=limit)
        e = _native.Engine(d["g"], d["m"], path=2, lib=emu_lib)
        e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
        for part in np.array_split(combos, 3):
            e.accumulate(part)
        e.finalize()
        assert np.array_equal(e.get_counts(), want), limit
        st = e.stats()
        assert st["cell_updates"] == U and st["combos_done"] == combos.size
        launches[limit] = st["launches"]
        e.close()
    assert launches[max_words] > launches[None]


@pytest.mark.parametrize("env", [{"sparse_global": "1"}, {"list_max_words": "2000"}, {}, {"sparse_sync": "1"},
                                 {"guard_cap": "100"}])
def test_emu_variance_mode_sparse_fallbacks(emu_lib, monkeypatch, env):
    """Variance mode on the sparse dataflow: the iterations of a batch share one sparse pass (a u32
    triangle per slot), unless no update streams exist (atomics) or a batch has too many update words
    for one stream (then: one iteration at a time) — all forms give the reference's stdevs; so do
    batches sized exactly (tuning sparse_sync) and batches that overflow their guard and are redone."""
    from fastsk_amd import _native
    set_tuning_env(monkeypatch, **env)
    d = load_golden("f5_prot11_variance_T1_it9")
    e = _native.Engine(d["g"], d["m"], t=d["t"], approx=True, delta=d["delta"], max_iters=d["max_iters"], path=2, lib=emu_lib)
    e.set_combo_order(d["order"])
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_stdevs(), d["stdevs"])
    assert np.array_equal(e.get_triangle(), d["tri"])
    if "guard_cap" in env:
        assert e.stats()["batches_redone"] > 0
    e.close()


@pytest.mark.parametrize("g,m,force,rare", [(8, 4, None, None), (9, 4, None, "1"), (8, 4, "1", "1"), (7, 5, "1", "1"), (8, 4, "1", "0"),
                                            (7, 5, "1", "0")])
def test_emu_key_compaction_rare_symbol(emu_lib, port, monkeypatch, g, m, force, rare):
    """DNA with a few 'n': the 5^k key space is mostly empty; the dense dataflow counts only the
    keys that occur (per-combo rank table) and must still match the oracle bit for bit — with the keys found by
    a marking pass over every window (tuning compact_rare=0) and from the places of the rare symbol alone (=1)."""
    from fastsk_amd import _native
    if force is not None:
        set_tuning_env(monkeypatch, compact=force)
    if rare is not None:
        set_tuning_env(monkeypatch, compact_rare=rare)
    rng = np.random.default_rng(g * 7 + m)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(g, 80, size=150)]
    for i in (3, 70, 149):
        X[i][len(X[i]) // 2] = 5        # a single 'n'
    X[10][:6] = 5                        # a short run of 'n'
    X[20][:] = 1                         # low complexity: exercises the u8 fallback with compaction
    tok, off = _native.flatten(X)
    nc = port.num_combos(g, m)
    combos = np.unique(np.linspace(0, nc - 1, 7).astype(np.int32))
    want, _, _ = port.raw_counts(tok, off, g, m, combos, threads=4)
    e = _native.Engine(g, m, path=1, lib=emu_lib, profile=True)
    e.load_sequences(tok, off, 100, 50)
    e.accumulate(combos[:3])
    e.accumulate(combos[3:])
    e.finalize()
    assert np.array_equal(e.get_counts(), want)
    st = e.stats()
    assert st["key_space"] == 5 ** (g - m) and st["compact_keys_avg"] > 0 and st["compact_keys_avg"] < st["key_space"]


def test_emu_dense_reloads_keep_or_rebuild_the_tile_table(emu_lib, port):
    """Dense dataflow, one engine, several sets of sequences: a set with the same number of sequences and the same train / test
    split keeps the tile table on the device (and reads the kept positions of consecutive combos from the resident table of
    all combos); another size, another split or a combo list with gaps rebuilds / uploads what it needs."""
    from fastsk_amd import _native
    rng = np.random.default_rng(9)
    g, m = 7, 3
    e = _native.Engine(g, m, path=1, lib=emu_lib)
    for (N, ntr, combos) in [(200, 150, np.arange(35, dtype=np.int32)), (200, 150, np.arange(5, 30, dtype=np.int32)),
                             (200, 100, np.array([0, 3, 4, 9, 30], dtype=np.int32)), (140, 140, np.arange(35, dtype=np.int32)),
                             (330, 200, np.arange(10, 20, dtype=np.int32))]:
        X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(g, 70, size=N)]
        tok, off = _native.flatten(X)
        want, _, _ = port.raw_counts(tok, off, g, m, combos, threads=4)
        e.load_sequences(tok, off, ntr, N - ntr)
        e.accumulate(combos)
        e.finalize()
        assert np.array_equal(e.get_counts(), want), (N, ntr)
    e.close()


def test_emu_many_flagged_rows_in_one_stage(emu_lib, port):
    """More k-mers with counts above 15 in one 32-row stage than hi-plane rows ride along with
    the prefetch (4): the tile kernel fetches the rest in extra rounds."""
    from fastsk_amd import _native
    rng = np.random.default_rng(3)
    L = 90
    X = [rng.integers(1, 5, size=L).astype(np.int32) for _ in range(140)]
    pats = [[1], [2], [3], [4], [1, 2], [3, 4], [1, 3], [2, 4], [1, 2, 3], [4, 3, 2, 1]]
    for i, pat in enumerate(pats):
        X[3 + 13 * i] = np.array((pat * L)[:L], dtype=np.int32)   # spread over three panels
    X[70] = np.array([1] * 45 + [2] * 45, dtype=np.int32)
    tok, off = _native.flatten(X)
    combos = np.array([0, 1, 34, 69], dtype=np.int32)
    want, _, _ = port.raw_counts(tok, off, 8, 4, combos, threads=4)
    e = _native.Engine(8, 4, path=1, lib=emu_lib)
    e.load_sequences(tok, off, 100, 40)
    e.accumulate(combos)
    e.finalize()
    assert np.array_equal(e.get_counts(), want)


# ---- the unpacked entry format of the sparse dataflow (N >= 65,535 sequences or a sequence of >= 65,536 windows): the kernel
# instantiations k_sx_seg_write<RecT, false> and k_sx_emit<DIRECT | streams, SKIP, false> (countAndUpdateTri, shared.cpp:268-333)
@pytest.mark.parametrize("global_pairs", ["0", "1"])
@pytest.mark.parametrize("name", ["f6_prot219_skipvar16", "f3_ragged_sigma7_g6m3", "f3_lowcomplexity_g5m2", "f5_prot11_variance_T1_it9"])
def test_emu_sparse_unpacked_entries_forced(emu_lib, monkeypatch, name, global_pairs):
    """tuning sparse_unpacked=1: the general entry format on the golden vectors, through the update streams and as atomics,
    exact and variance mode."""
    set_tuning_env(monkeypatch, sparse_unpacked="1", sparse_global=global_pairs)
    d = load_golden(name)
    e = run_case(emu_lib, d, 2)
    assert e.get_tuning("sparse_unpacked") == 1 and e.stats()["path_used"] == 2
    assert np.array_equal(e.get_triangle(), d["tri"])
    if "counts" in d:
        assert np.array_equal(e.get_counts(), d["counts"])
    if d["approx"] and not d["skip_variance"]:
        assert np.array_equal(e.get_stdevs(), d["stdevs"])
    e.close()


@pytest.mark.parametrize("global_pairs", ["0", "1"])
def test_emu_sparse_unpacked_entries_skip_test_block(emu_lib, port, monkeypatch, global_pairs):
    from fastsk_amd import _native
    set_tuning_env(monkeypatch, sparse_unpacked="1", sparse_global=global_pairs)
    rng = np.random.default_rng(4)
    N, ntr = 300, 100
    X = rng.integers(1, 5, size=(N, 24), dtype=np.int32)
    tokens, offsets = X.reshape(-1), np.arange(N + 1, dtype=np.int64) * 24
    combos = np.arange(0, 35, 2, dtype=np.int32)
    raw, _, U = port.raw_counts(tokens, offsets, 7, 4, combos, threads=4)
    a, b = np.tril_indices(N)
    keep = (b < ntr) | (a == b)
    e = _native.Engine(7, 4, path=2, lib=emu_lib, skip_test_block=True)
    e.load_sequences(tokens, offsets, ntr, N - ntr)
    e.accumulate(combos)
    e.finalize()
    got = e.get_counts()
    assert np.array_equal(got[keep], raw[keep]) and not got[~keep].any() and raw[~keep].any()
    assert e.stats()["cell_updates"] < U
    e.close()


@pytest.mark.parametrize("global_pairs,skip", [("0", False), ("1", False), ("0", True)])
def test_emu_sparse_one_sequence_of_65536_windows(emu_lib, port, monkeypatch, global_pairs, skip):
    """A sequence of more than 65,535 windows among ordinary ones: multiplicities and ranks no longer fit 16 bits, so the
    engine itself takes the unpacked entries — with the update streams (a few multi-word products included) and as atomics."""
    from fastsk_amd import _native
    set_tuning_env(monkeypatch, sparse_global=global_pairs)
    rng = np.random.default_rng(66)
    g, m = 8, 4
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(30, 80, size=40)]
    X.insert(17, rng.integers(1, 5, size=65600 + g - 1).astype(np.int32))
    N, ntr = len(X), 25
    tokens, offsets = _native.flatten(X)
    combos = np.array([11, 52], dtype=np.int32)
    raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=4)
    e = _native.Engine(g, m, path=2, lib=emu_lib, skip_test_block=skip)
    e.load_sequences(tokens, offsets, ntr if skip else N, N - ntr if skip else 0)
    e.accumulate(combos)
    e.finalize()
    st = e.stats()
    assert st["path_used"] == 2 and st["max_windows"] == 65600 and e.get_tuning("sparse_unpacked") == 0
    got = e.get_counts()
    if skip:
        a, b = np.tril_indices(N)
        keep = (b < ntr) | (a == b)
        assert np.array_equal(got[keep], raw[keep]) and not got[~keep].any() and raw[~keep].any()
    else:
        assert np.array_equal(got, raw) and st["cell_updates"] == U
    e.close()


@pytest.mark.parametrize("name", ["f8_sigma300_g5m2", "f8_prot11_g20m4_skipvar12"])
@pytest.mark.parametrize("tune", [{}, {"sparse_global": "1"}, {"sparse_unpacked": "1"}])
def test_emu_wide_alphabets_and_wide_keys(emu_lib, monkeypatch, name, tune):
    """Inputs upstream computes that the packed fast paths do not cover (cntsrtna takes any dictionary size and any k,
    shared.cpp:156-191): 300 distinct tokens (16-bit symbols in HBM) and protein at g=20 m=4 (24^16 > 2^62: the k-mer key
    is the symbols' bit fields side by side, in 128-bit sort records) — the reference's own outputs."""
    set_tuning_env(monkeypatch, **tune)
    d = load_golden(name)
    e = run_case(emu_lib, d, 0)
    st = e.stats()
    assert st["path_used"] == 2 and st["alphabet"] == (300 if "sigma300" in name else st["alphabet"])
    assert st["bits_per_symbol"] == (16 if "sigma300" in name else 8)
    assert np.array_equal(e.get_counts(), d["counts"]) and np.array_equal(e.get_triangle(), d["tri"])
    assert np.array_equal(e.get_train(), d["train"]) and np.array_equal(e.get_test(), d["test"])
    e.close()


@pytest.mark.parametrize("skip", [False, True])
def test_emu_sparse_paired_unit_words(emu_lib, port, monkeypatch, skip):
    """Update streams with PAIRS (bands of fewer than 32767 cells): the words of UNIT entries — multiplicity 1 and every partner
    of multiplicity 1 — travel as bare cells, two to a 32-bit container, padded per owner band and per pass of k_sx_emit. Runs of
    ~20 entries over 400 keys: ten words an entry, so a tile's short entries take several passes, about half of them unit;
    with the pairs off (tuning sparse_pairs=0), in one call and in three, whole and in row bands, with skip_test_block."""
    from fastsk_amd import _native
    rng = np.random.default_rng(12)
    N, ntr, g, m = 300, 180, 4, 2
    X = [rng.integers(1, 21, size=int(L)).astype(np.int32) for L in rng.integers(24, 36, size=N)]
    X[7][:] = 3   # one low-complexity sequence: multiplicities above 1 inside otherwise clean runs
    tokens, offsets = _native.flatten(X)
    combos = np.arange(6, dtype=np.int32)
    raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=4)
    a, b = np.tril_indices(N)
    keep = (b < ntr) | (a == b) if skip else np.ones(len(a), dtype=bool)
    for pairs in ("1", "0"):
        set_tuning_env(monkeypatch, sparse_pairs=pairs)
        for how in ("whole", "three calls", "row bands"):
            e = _native.Engine(g, m, path=2, lib=emu_lib, skip_test_block=skip)
            e.load_sequences(tokens, offsets, ntr if skip else N, N - ntr if skip else 0)
            if how == "whole":
                e.accumulate(combos)
            elif how == "three calls":
                for part in np.array_split(combos, 3):
                    e.accumulate(part)
            else:
                for lo, hi in ((0, 128), (128, 256), (256, N)):
                    e.accumulate_rows(combos, lo, hi)
            e.finalize()
            got = e.get_counts()
            assert np.array_equal(got[keep], raw[keep]), (pairs, how)
            assert skip or e.stats()["cell_updates"] == U
            e.close()


@pytest.mark.parametrize("skip", [False, True])
def test_emu_sparse_long_runs_grouped_entries(emu_lib, port, skip):
    """Runs of hundreds of entries (400 sequences over 16 keys): entries of more than 48 partners are taken in groups of 16 list
    neighbours that share the reads of their partners (k_sx_emit, class 3) — runs that begin before the tile, own cells of
    multiplicities above 1, groups that span two runs; with skip_test_block they go a wave each (class 2)."""
    from fastsk_amd import _native
    rng = np.random.default_rng(77)
    N, ntr, g, m = 400, 250, 5, 3
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(14, 30, size=N)]
    tokens, offsets = _native.flatten(X)
    combos = np.array([0, 4, 9], dtype=np.int32)
    raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=4)
    e = _native.Engine(g, m, path=2, lib=emu_lib, skip_test_block=skip)
    e.load_sequences(tokens, offsets, ntr if skip else N, N - ntr if skip else 0)
    e.accumulate(combos)
    e.finalize()
    got = e.get_counts()
    if skip:
        a, b = np.tril_indices(N)
        keep = (b < ntr) | (a == b)
        assert np.array_equal(got[keep], raw[keep]) and not got[~keep].any()
    else:
        assert np.array_equal(got, raw) and e.stats()["cell_updates"] == U
    e.close()


@pytest.mark.parametrize("case", ["protein_k4", "sigma300_u64", "bitfield_u128", "golden_k3"])
def test_emu_sparse_shared_leading_positions(emu_lib, port, monkeypatch, case):
    """Shared prefixes (k_sx_group_tables ... k_sx_extract_shared): the windows are sorted by the leading kept positions once per
    group of consecutive slots that share them, every slot then sorts the rest of its key alone. Forced at every prefix length
    (tuning sparse_share; the product decides by cost, from 2^24 records a batch on): 32-, 64- and 128-bit sort records, 4- and
    8-byte presort records, a combo list with repeated and non-consecutive ids (groups of one), split calls, skip_test_block."""
    from fastsk_amd import _native
    rng = np.random.default_rng(5)
    skip = False
    if case == "golden_k3":
        d = load_golden("f3_ragged_sigma7_g6m3")
        tokens, offsets, g, m, N, ntr = d["tokens"], d["offsets"], d["g"], d["m"], d["n_train"] + d["n_test"], d["n_train"]
        combos = np.asarray(d["combos"], dtype=np.int32)
        shares = (1, 2)
    else:
        if case == "protein_k4":
            sigma, g, m, N, shares, skip = 20, 7, 3, 48, (1, 2, 3), True
        elif case == "sigma300_u64":
            sigma, g, m, N, shares = 300, 7, 3, 40, (1, 3)         # 300^4 > 2^32: 64-bit records
        else:
            sigma, g, m, N, shares = 200, 16, 4, 16, (5, 8)  # 200^12 > 2^62: bit-field keys in 128-bit records; 8-byte presort records
                                                                  # (5), a leading part no presort record holds (8: sorted plainly)
        X = [rng.integers(1, sigma + 1, size=int(L)).astype(np.int32) for L in rng.integers(g + 4, g + 22, size=N)]
        X[3][:] = 2
        tokens, offsets = _native.flatten(X)
        ntr = N * 2 // 3
        nc = port.num_combos(g, m)
        first = 0 if nc < 40 else int(rng.integers(0, nc - 40))
        combos = np.arange(first, min(nc, first + 35), dtype=np.int32)
        if case == "protein_k4":  # (any list is a valid call: repeats, a jump back)
            combos = np.concatenate([combos, combos[5:8], np.asarray([1, 30, 2], dtype=np.int32)])
    raw, _, _ = port.raw_counts(tokens, offsets, g, m, combos, threads=4)
    a, b = np.tril_indices(N)
    keep = (b < ntr) | (a == b) if skip else np.ones(len(a), dtype=bool)
    for share in shares:
        for split in ((False, True) if case == "golden_k3" or (case == "protein_k4" and share == 2) else (False,)):
            set_tuning_env(monkeypatch, sparse_share=str(share))
            e = _native.Engine(g, m, path=2, lib=emu_lib, skip_test_block=skip)
            e.load_sequences(tokens, offsets, ntr, N - ntr)
            if split:
                e.accumulate(combos[:19])
                e.accumulate(combos[19:])
            else:
                e.accumulate(combos)
            e.finalize()
            assert np.array_equal(e.get_counts()[keep], raw[keep]), (case, share, split)
            e.close()


@pytest.mark.parametrize("skip,desc", [(False, 0), (True, 0), (False, 1), (True, 1)])
def test_emu_sparse_two_level_blocks(emu_lib, port, skip, desc):
    """The two-level form of the update stage (fsk_sparse_blocks.inc: bands binned by k_sx_emit, every band's stream split by
    sub-band, one workgroup a sub-band) — what N beyond the owner bands takes instead of one atomic per += — forced on a small
    input with small blocks: several passes over row ranges (few bands a pass, a word budget that halves ranges), several bands a
    pass, several sub-bands a band; packed and general entries; protein-like runs with a low-complexity sequence and long DNA
    runs; whole, in two calls and in row bands; skip_test_block. desc: every entry above three partners as descriptor records, one
    per sub-band its partners fall into (k_sx_emit's bisection, k_sxb_dcount / k_sxb_dscatter, sx_expand_descriptors in
    k_sxb_consume), its own cell as a word. Against the oracle's counts and its exact U."""
    from fastsk_amd import _native
    rng = np.random.default_rng(21)
    N, ntr = 140, 90
    X1 = [rng.integers(1, 21, size=int(L)).astype(np.int32) for L in rng.integers(20, 30, size=N)]
    X1[5][:] = 7   # a low-complexity sequence: multiplicities above 1, own cells
    X2 = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(24, 40, size=N)]  # 64 keys: runs of a hundred entries
    a, b = np.tril_indices(N)
    keep = (b < ntr) | (a == b) if skip else np.ones(len(a), dtype=bool)
    plans = [(X1, 4, 2, np.arange(4, dtype=np.int32), {"sparse_form": 2}, "whole"),
             (X1, 4, 2, np.arange(4, dtype=np.int32), {"sparse_form": 2, "blocks_sub_shift": 6, "blocks_max_bands": 5, "blocks_band_shift_max": 9}, "row bands"),
             (X2, 5, 2, np.array([0, 9], dtype=np.int32), {"sparse_form": 2, "blocks_sub_shift": 5, "blocks_max_bands": 12, "blocks_band_shift_max": 8,
                                                           "blocks_pass_words": 30000, "sparse_unpacked": 1}, "two calls"),
             # (two bands of 2^5 cells a pass: the rows beyond 64 sequences are passes of ONE row with more bands than blocks_max_bands)
             (X1[:100], 4, 2, np.arange(1, dtype=np.int32), {"sparse_form": 2, "blocks_sub_shift": 4, "blocks_max_bands": 2, "blocks_band_shift_max": 5}, "whole")]
    if skip:
        plans.pop()
    for X, g, m, combos, tun, how in plans:
        tokens, offsets = _native.flatten(X)
        raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=4)
        if len(X) != N:  # (the plan on the first hundred sequences)
            a, b = np.tril_indices(len(X))
            keep = np.ones(len(a), dtype=bool)
        tun = dict(tun, sparse_desc=1, sparse_desc_min=3) if desc else dict(tun, sparse_desc=-1)
        e = _native.Engine(g, m, path=2, lib=emu_lib, skip_test_block=skip, tuning=tun)
        e.load_sequences(tokens, offsets, ntr if skip else len(X), N - ntr if skip else 0)
        if how == "whole":
            e.accumulate(combos)
        elif how == "two calls":
            for part in np.array_split(combos, 2):
                e.accumulate(part)
        else:
            for lo, hi in ((0, 128), (128, N)):  # (row bands: multiples of 128)
                e.accumulate_rows(combos, lo, hi)
        e.finalize()
        st = e.stats()
        assert st["sparse_form"] == 2 and st["sparse_desc"] == desc
        assert "blocks_sub_shift" not in tun or st["sparse_passes"] > 2
        assert np.array_equal(e.get_counts()[keep], raw[keep]), (g, m, tun, how)
        assert skip or st["cell_updates"] == U
        e.close()


@pytest.mark.parametrize("desc_min,skip", [(48, False), (6, True)])
def test_emu_sparse_descriptors(emu_lib, port, monkeypatch, skip, desc_min):
    """Descriptors (tuning sparse_desc=1): an entry of more partners than k_sx_emit bins in LDS leaves as ONE descriptor
    {first partner's entry, partners, row, multiplicity} in its band's descriptor stream and k_sx_consume walks the partners
    itself. Runs of a hundred entries (160 sequences over 16 keys), own cells of multiplicities above 1, runs that begin
    before the tile; every entry above six partners as a descriptor (sparse_desc_min); with the paired unit words on and
    off; in one call, in three, in row bands; with skip_test_block (partners = the run's train entries)."""
    from fastsk_amd import _native
    rng = np.random.default_rng(78)
    N, ntr, g, m = 160, 100, 5, 3
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(14, 22, size=N)]
    X[11][:] = 2  # a low-complexity sequence: multiplicities above 1
    tokens, offsets = _native.flatten(X)
    combos = np.array([0, 4, 9], dtype=np.int32)
    raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=4)
    a, b = np.tril_indices(N)
    keep = (b < ntr) | (a == b) if skip else np.ones(len(a), dtype=bool)
    # (pairs off: also with the bands' streams cut into several parts each — every part takes every nparts-th descriptor of its band
    # and adds into K with atomics — the stream of a band in pieces of whole 16 bytes)
    for pairs, parts, hows in (("1", None, ("row bands",)), ("0", 256, ("three calls",))):
        set_tuning_env(monkeypatch, sparse_desc=1, sparse_desc_min=desc_min, sparse_pairs=pairs, sparse_form=1, sparse_parts_target=parts,
                       sparse_desc_parts=48 if parts else None)
        for how in hows:
            e = _native.Engine(g, m, path=2, lib=emu_lib, skip_test_block=skip)
            e.load_sequences(tokens, offsets, ntr if skip else N, N - ntr if skip else 0)
            if how == "whole":
                e.accumulate(combos)
            elif how == "three calls":
                for part in np.array_split(combos, 3):
                    e.accumulate(part)
            else:
                for lo, hi in ((0, 128), (128, N)):
                    e.accumulate_rows(combos, lo, hi)
            e.finalize()
            got = e.get_counts()
            st = e.stats()
            assert st["sparse_desc"] == 1 and st["sparse_form"] == 0
            assert np.array_equal(got[keep], raw[keep]), (pairs, how)
            assert skip or st["cell_updates"] == U
            e.close()


@pytest.mark.parametrize("name,unpacked,cols", [("f6_prot219_skipvar16", "1", "1"), ("f3_lowcomplexity_g5m2", "0", "3"),
                                                ("f3_lowcomplexity_g5m2", "1", "0"), ("f3_lowcomplexity_g5m2", "0", "2"), ("f5_prot11_variance_T1_it9", "0", "1"),
                                                ("f5_prot11_variance_T1_it9", "1", "1"), ("f4_ep300_variance_T1", "0", "3"), ("f4_ep300_variance_T1", "0", "0")])
def test_emu_sparse_descriptors_forced_on_the_goldens(emu_lib, monkeypatch, name, unpacked, cols):
    """tuning sparse_desc=1 with every entry above two partners as a descriptor, on the golden vectors: exact, skip-variance and
    variance mode (the by-slot form of k_sx_consume: a slot's descriptors are a contiguous piece of every band's descriptor
    stream), both entry formats; the partners read from the 2-byte and the 4-byte column array (an entry of multiplicity above 1:
    from the entries themselves) and from the entries alone (tuning sparse_desc_cols)."""
    set_tuning_env(monkeypatch, sparse_desc="1", sparse_desc_min="2", sparse_unpacked=unpacked, sparse_desc_cols=cols)
    d = load_golden(name)
    e = run_case(emu_lib, d, 2)
    st = e.stats()
    assert st["path_used"] == 2 and st["sparse_desc"] == 1
    assert np.array_equal(e.get_triangle(), d["tri"])
    if "counts" in d:
        assert np.array_equal(e.get_counts(), d["counts"])
    if d["approx"] and not d["skip_variance"]:
        assert np.array_equal(e.get_stdevs(), d["stdevs"])
    e.close()


def test_emu_sparse_descriptors_switch_on_by_the_data(emu_lib, port):
    """tuning sparse_desc=0 (the default): the first batch of a set of sequences goes out as update words; once it has shown
    sparse_desc_from pairs a sort record (here: runs of ~60 entries over 16 keys) the batches that follow send descriptors — and
    forget the words per record seen so far, so the next one is sized exactly, not under the old guard; short runs never switch.
    A reload of sequences of the same shape keeps the decision (sparse_hint), another shape starts over."""
    from fastsk_amd import _native
    rng = np.random.default_rng(9)
    N, g, m = 150, 5, 3
    long_runs = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(12, 20, size=N)]
    short_runs = [rng.integers(1, 21, size=int(L)).astype(np.int32) for L in rng.integers(12, 20, size=N)]
    combos = np.arange(6, dtype=np.int32)
    e = _native.Engine(g, m, path=2, lib=emu_lib)
    for X, want_desc in ((long_runs, [0, 1, 1]), (short_runs, [0, 0, 0])):
        tokens, offsets = _native.flatten(X)
        raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=4)
        e.load_sequences(tokens, offsets, N, 0)
        seen = []
        for part in np.array_split(combos, 3):
            e.accumulate(part)
            e.synchronize()
            seen.append(int(e.stats()["sparse_desc"]))
        e.finalize()
        assert seen == want_desc, seen
        assert np.array_equal(e.get_counts(), raw)
    tokens, offsets = _native.flatten(long_runs)
    e.load_sequences(tokens, offsets, N, 0)   # (another shape than the last one loaded: the decision starts over)
    e.accumulate(combos[:2])
    assert int(e.stats()["sparse_desc"]) == 0
    e.load_sequences(tokens, offsets, N, 0)   # (the same shape again: kept)
    e.accumulate(combos[:2])
    assert int(e.stats()["sparse_desc"]) == 1
    e.close()


def test_emu_sparse_descriptors_two_lds_rounds(emu_lib, port):
    """Owner bands of TWO LDS rounds (N = 4200: 270 bands of 2^15 cells + a row, 20480 cells a round) with descriptors: every round
    of k_sx_consume walks the band's descriptors again and keeps the cells of its own range (the round's first cell in
    sx_expand_descriptors' base); runs of ~100 entries over 400 keys, a low-complexity sequence."""
    from fastsk_amd import _native
    rng = np.random.default_rng(3)
    N, g, m = 4200, 4, 2
    X = [rng.integers(1, 21, size=int(L)).astype(np.int32) for L in rng.integers(8, 12, size=N)]
    X[9][:] = 4
    tokens, offsets = _native.flatten(X)
    combos = np.array([0, 3], dtype=np.int32)
    raw, _, U = port.raw_counts(tokens, offsets, g, m, combos, threads=4)
    e = _native.Engine(g, m, path=2, lib=emu_lib, tuning={"sparse_desc": 1, "sparse_desc_min": 6, "sparse_form": 1})
    e.load_sequences(tokens, offsets, N, 0)
    e.accumulate(combos)
    e.finalize()
    st = e.stats()
    assert st["sparse_desc"] == 1 and st["sparse_form"] == 0 and st["cell_updates"] == U
    assert np.array_equal(e.get_counts(), raw)
    e.close()

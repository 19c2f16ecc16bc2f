"""fsk_create_multi — one engine over several devices of the process — on the CPU: the group's host code
(one worker thread per device, combos dealt round-robin, row bands, the event-ordered exchange, int32
narrowing, variance chains dealt over the engines) runs against the emulated HIP runtime (eight pretend
devices in one address space) with the engine's own peer-to-peer all-reduce kernels as the collective.
The GPU box runs the same code with RCCL over a world of one and with P2P over a device listed twice
(tests/test_gpu_parity.py); real multi-GPU runs are the driver's."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden, synthetic_dna, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


@pytest.fixture(scope="module")
def emu_lib():
    import build_emu
    from fastsk_amd import _native
    return _native.Library(build_emu.build())


def engine_for(emu_lib, d, devices, path=0, **kw):
    from fastsk_amd import _native
    e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"],
                       skip_variance=bool(d["skip_variance"]), path=path, lib=emu_lib, devices=devices, **kw)
    if d["approx"]:
        e.set_combo_order(d["order"])
    return e


@pytest.mark.parametrize("name,devices", [("f4_ep300_exact", [0]), ("f4_ep300_exact", [0, 1]), ("f4_ep300_exact", [3, 1, 0, 2]),
                                          ("f4_ep300_exact", [0, 0, 0]), ("f3_ragged_sigma7_g6m3", [0]),
                                          ("f3_ragged_sigma7_g6m3", [0, 1]), ("f3_ragged_sigma7_g6m3", [5, 5, 2, 7, 1]),
                                          ("f3_lowcomplexity_g5m2:sparse", [0, 1, 2])])
def test_group_exact_equals_golden(emu_lib, name, devices):
    name, _, flow = name.partition(":")   # (":sparse": the group's engines on the sparse dataflow)
    d = load_golden(name)
    e = engine_for(emu_lib, d, devices, **({"path": 2} if flow else {}))
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_counts(), d["counts"])
    assert np.array_equal(e.get_triangle(), d["tri"])
    assert np.array_equal(e.get_train(), d["train"])
    info = e.multi_info()
    assert info["ndev"] == len(devices) and info["devices"] == devices and info["collective"] == "p2p"
    assert info["comm_ranks"] == len(devices) and info["narrow"]
    assert sum(info["combos_per_engine"]) == len(d["combos"]) == e.stats()["combos_done"]
    assert max(info["combos_per_engine"]) - min(info["combos_per_engine"]) <= 1   # fastsk_kernel.cpp:148,275
    e.close()


@pytest.mark.parametrize("name", ["f4_ep300_skipvar_T1", "f4_ep300_skipvar_T3", "f6_prot219_skipvar16"])
def test_group_skip_variance(emu_lib, name):
    d = load_golden(name)
    e = engine_for(emu_lib, d, [0, 1, 2])
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_triangle(), d["tri"])
    assert np.array_equal(e.get_test(), d["test"])
    if "counts" in d:
        assert np.array_equal(e.get_counts(), d["counts"])
    e.close()


@pytest.mark.parametrize("devices,T", [([0, 1], 1), ([0, 1], 2), ([0, 1, 2, 3], 2), ([0, 1, 2], 5)])
def test_group_variance_chains(emu_lib, port, devices, T):
    """The T Welford chains dealt over the engines (chain t on engine t mod R), one fp64 all-reduce of their
    K_hat. stdevs are chain 0's: bit-identical. The kernel is a sum of T fp64 terms: bit-identical for
    T <= 2 (one addition), equal to rounding beyond (the reference adds them in thread-arrival order)."""
    from fastsk_amd import _native
    rng = np.random.default_rng(5)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(12, 40, size=30)]
    tok, off = _native.flatten(X)
    g, m = 7, 3
    order = rng.permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tok, off, 22, 8, g, m, t=T, approx=True, delta=0.5, max_iters=6, order=order)
    e = _native.Engine(g, m, t=T, approx=True, delta=0.5, max_iters=6, lib=emu_lib, devices=devices)
    e.set_combo_order(order)
    e.compute(tok, off, 22, 8)
    assert np.array_equal(e.get_stdevs(), sd)
    got = e.get_triangle()
    if T <= 2:
        assert np.array_equal(got, want)
    else:
        assert np.allclose(got, want, rtol=1e-14, atol=0)
    e.close()


def test_group_staged_calls_bands_widths_and_resets(emu_lib, port):
    """load -> (reset -> accumulate ... -> finalize) passes through the staged entry points: accumulate is
    additive on the group as on one engine (the other engines hold partial sums only and start every
    accumulate from zero), a reset starts a new pass, an engine without combos contributes zeros (not the
    previous pass), both payload widths and several row bands."""
    from fastsk_amd import _native
    rng = np.random.default_rng(9)
    N = 300
    X = rng.integers(1, 5, size=(N, 40), dtype=np.int32)
    X[::13, 3:36] = 3   # counts above 15: the hi plane
    tok, off = _native.flatten(X)
    g, m = 8, 4
    ca, cb, cc = np.arange(0, 70, 5, dtype=np.int32), np.arange(1, 70, 9, dtype=np.int32), np.array([7, 9], dtype=np.int32)
    wa, _, _ = port.raw_counts(tok, off, g, m, ca, threads=4)
    wb, _, _ = port.raw_counts(tok, off, g, m, cb, threads=4)
    wc, _, _ = port.raw_counts(tok, off, g, m, cc, threads=4)
    single = _native.Engine(g, m, lib=emu_lib, path=1)
    single.load_sequences(tok, off, N, 0)
    single.accumulate(ca)
    single.finalize()
    want_digest = single.counts_digest()
    assert np.array_equal(single.get_counts(), wa)
    for devices, bands, path in (([0, 1, 2, 3], 3, 1), ([0, 1, 2], 2, 2)):
        e = _native.Engine(g, m, lib=emu_lib, devices=devices, bands=bands, path=path)
        e.load_sequences(tok, off, N, 0)
        e.accumulate(ca)
        e.finalize()
        assert np.array_equal(e.get_counts(), wa)
        assert e.counts_digest() == want_digest
        lo = e.counts_digest(0, 128), e.counts_digest(128, N)   # digests of row ranges combine
        assert ((lo[0][0] + lo[1][0]) % 2 ** 64, lo[0][1] ^ lo[1][1]) == want_digest
        assert e.multi_info()["bands"] == min(bands, 3)   # (N = 300: three tile rows)
        e.accumulate(cb)                       # additive: K = a + b
        e.finalize()
        assert np.array_equal(e.get_counts(), wa + wb)
        e.reset_counts()
        e.accumulate(cc)                       # two combos over up to four engines: some engines have none
        e.finalize()
        assert np.array_equal(e.get_counts(), wc)
        e.reset_counts()
        assert not e.get_counts().any()
        e.accumulate(cb); e.accumulate(cc); e.accumulate(ca)
        e.synchronize()
        e.finalize()
        assert np.array_equal(e.get_counts(), wa + wb + wc)
        assert np.array_equal(e.get_triangle(), port.normalise((wa + wb + wc).astype(np.float64), N))
        with pytest.raises(_native.FskError):
            e.accumulate_rows(ca, 0, 128)      # a single-engine call
        e.close()


def test_group_wide_cells_travel_as_uint64(emu_lib, port):
    """C(g,m) * max_windows^2 >= 2^31 (one very long sequence): no narrowing, the band is all-reduced in
    place as uint64."""
    from fastsk_amd import _native
    rng = np.random.default_rng(21)
    X = [rng.integers(1, 5, size=n).astype(np.int32) for n in (20000, 50, 64, 41, 77, 58)]
    tok, off = _native.flatten(X)
    g, m = 6, 2   # 15 combos x 19995^2 > 2^31
    want, _, _ = port.compute(tok, off, 4, 2, g, m, t=1)
    for devices, path in (([0, 1], 0), ([0, 1, 2], 2)):
        e = _native.Engine(g, m, lib=emu_lib, devices=devices, path=path)
        e.compute(tok, off, 4, 2)
        info = e.multi_info()
        assert not info["narrow"] and info["reduce_bytes"] == 8 * 21
        assert np.array_equal(e.get_triangle(), want)
        e.close()


def test_group_errors(emu_lib):
    from fastsk_amd import _native
    with pytest.raises(_native.FskError, match="device ordinal"):
        _native.Engine(5, 2, lib=emu_lib, devices=[0, 99])
    with pytest.raises(_native.FskError, match="distinct devices"):
        _native.Engine(5, 2, lib=emu_lib, devices=[0, 0], collective=_native.COLL_RCCL)
    e = _native.Engine(6, 2, lib=emu_lib, devices=[0, 1])
    with pytest.raises(_native.FskError) as err:   # the reference printf+exit(1)s here (fastsk.cpp:53-58)
        e.compute(np.array([1, 2, 3, 1, 2, 3, 1, 2], dtype=np.int32), np.array([0, 3, 8]), 1, 1)
    assert err.value.code == -2 and "shortest" in str(err.value)
    tok = np.tile(np.array([1, 2, 3, 4], dtype=np.int32), 20)
    e.compute(tok, np.array([0, 40, 80]), 1, 1)     # the group is usable after a failed call
    assert e.get_train()[0, 0] == 1.0
    e.close()


# ---- fail fast: a stuck engine must not hang the group (fsk_config.deadline_ms) ---------------------------------
def test_group_deadline_names_the_band_and_poisons_the_group(emu_lib, monkeypatch):
    """Engine 1's worker is late by 2.5 s before the collective of band 0 (the fault_* tuning keys of test builds): with a 300 ms
    deadline the other engines give up at that band's exchange, the call returns FSK_EDEVICE naming the band well
    before a run without a deadline would have returned, every later call repeats the first failure (the group is
    dead, never half alive), and destroying the handle does not hang."""
    import time
    from fastsk_amd import _native
    d = load_golden("f4_ep300_exact")
    e = _native.Engine(d["g"], d["m"], lib=emu_lib, devices=[0, 1, 2], bands=3, deadline_ms=300,
                       tuning={"fault_kind": 1, "fault_rank": 1, "fault_band": 0, "fault_ms": 2500})
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    t0 = time.perf_counter()
    with pytest.raises(_native.FskError) as ei:
        e.accumulate(np.arange(0, 30, dtype=np.int32))
        e.finalize()
    dt = time.perf_counter() - t0
    assert ei.value.code == -4
    msg = str(ei.value)
    assert "300 ms" in msg and "band 0" in msg, msg
    assert dt < 6.0   # (the late worker itself still has to come back: 2.5 s; nothing waits for ever)
    with pytest.raises(_native.FskError) as ei2:
        e.accumulate(np.arange(0, 30, dtype=np.int32))
    assert "dead after an earlier failure" in str(ei2.value) and "band 0" in str(ei2.value)
    with pytest.raises(_native.FskError):
        e.finalize()
    t1 = time.perf_counter()
    e.close()
    assert time.perf_counter() - t1 < 5.0


def test_group_without_fault_is_untouched_by_the_deadline(emu_lib, monkeypatch):
    """The same job with the deadline armed and no fault: identical result (the deadline only bounds waits)."""
    from fastsk_amd import _native
    d = load_golden("f4_ep300_exact")
    e = engine_for(emu_lib, d, [0, 1, 2], bands=3, deadline_ms=20000)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_counts(), d["counts"])
    e.close()
    # a late engine INSIDE the deadline only delays the result
    e = engine_for(emu_lib, d, [0, 1, 2], bands=2, deadline_ms=20000, tuning={"fault_kind": 1, "fault_rank": 2, "fault_band": 0, "fault_ms": 300})
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_counts(), d["counts"])
    e.close()


def test_group_bound_counts_are_not_narrowed(emu_lib):
    """fsk_bind_counts on a group handle: the caller's cells may hold anything, so the exchange stays 64 bits wide
    until a whole reset zeroes them (an int32 exchange would truncate a cell >= 2^31)."""
    from fastsk_amd import _native
    d = load_golden("f4_ep300_exact")
    N = d["n_train"] + d["n_test"]
    pairs = N * (N + 1) // 2
    big = np.full(pairs, (1 << 33) + 5, dtype=np.uint64)      # emulated "device memory" is host memory
    e = _native.Engine(d["g"], d["m"], lib=emu_lib, devices=[0, 1], bands=2)
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    e.bind_counts(big.ctypes.data, pairs, keepalive=big)
    combos = np.asarray(d["combos"], dtype=np.int32)
    e.accumulate(combos)
    e.finalize()
    assert not e.multi_info()["narrow"]
    assert np.array_equal(big, d["counts"] + np.uint64((1 << 33) + 5))
    e.reset_counts()                                           # zeros: the bound is known again
    e.accumulate(combos)
    e.finalize()
    assert e.multi_info()["narrow"]
    assert np.array_equal(big, d["counts"])
    e.close()


# ---- the RCCL collective with R >= 2 (fsk_multi.hip: RcclCollective) against the test build's stand-in for librccl ------
# (tests/emu/rccl_stub.cpp: R worker threads meet in a real rendezvous, sums in rank order, counts / types / devices checked,
# injectable failures). What it replaces is the reduce over the reference's worker threads, fastsk_kernel.cpp:286-315.
STAT = {"init_calls": 0, "comms": 1, "destroyed": 2, "aborted": 3, "allreduce_calls": 4, "int32": 5, "uint64": 6, "float64": 7,
        "bytes": 8, "wrong_device": 9, "mismatched": 10, "completed": 11, "ranks": 12}


def rccl_stats(emu_lib):
    import ctypes as C
    out = (C.c_int64 * 16)()
    emu_lib.L.emu_rccl_stats(out)
    return {k: int(out[i]) for k, i in STAT.items()}


@pytest.fixture
def rccl(emu_lib):
    emu_lib.L.emu_rccl_reset()
    yield emu_lib
    emu_lib.L.emu_rccl_reset()


@pytest.mark.parametrize("name,devices", [("f4_ep300_exact", [0, 1]), ("f4_ep300_exact", [3, 1, 0, 2]), ("f4_ep300_exact", list(range(8))),
                                          ("f3_ragged_sigma7_g6m3", [5, 2, 7, 1]), ("f6_prot219_skipvar16", [0, 1, 2, 3, 4, 5, 6, 7]),
                                          ("f1_small_g3m1", [0, 1, 2, 3, 4, 5, 6, 7])])
def test_rccl_group_exact_equals_golden(rccl, name, devices):
    """fsk_create_multi(collective = RCCL) with 2, 4 and 8 ranks: ncclCommInitAll over the listed devices, one int32
    ncclAllReduce per band and rank issued from the rank's own worker thread on its own device, the golden counts."""
    from fastsk_amd import _native
    d = load_golden(name)
    R = len(devices)
    e = engine_for(rccl, d, devices, collective=_native.COLL_RCCL)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_counts(), d["counts"])
    assert np.array_equal(e.get_triangle(), d["tri"])
    info = e.multi_info()
    assert info["collective"] == "rccl" and info["comm_ranks"] == R and info["ndev"] == R and info["narrow"]
    assert sum(info["combos_per_engine"]) == len(d["combos"])
    st = rccl_stats(rccl)
    assert st["init_calls"] == 1 and st["comms"] == R and st["ranks"] == R
    assert st["allreduce_calls"] == st["int32"] == R * info["bands"] and st["completed"] == info["bands"]
    assert st["bytes"] == R * info["reduce_bytes"]
    assert st["wrong_device"] == 0 and st["mismatched"] == 0 and st["aborted"] == 0
    e.close()
    assert rccl_stats(rccl)["destroyed"] == R


@pytest.mark.parametrize("R", [2, 4, 8])
def test_rccl_banded_exchange_both_widths_by_digest(rccl, port, R):
    """Several row bands, int32 and uint64 payloads, additive accumulates, resets, engines without combos — by the
    order-free digest against one engine and cell by cell against the oracle."""
    from fastsk_amd import _native
    rng = np.random.default_rng(31)
    N = 700
    X = rng.integers(1, 5, size=(N, 26), dtype=np.int32)
    X[::17, 2:24] = 2   # counts above 15: the hi plane
    tok, off = _native.flatten(X)
    g, m = 6, 3
    ca, cb = np.arange(0, 20, 2, dtype=np.int32), np.array([1, 7, 19], dtype=np.int32)
    wa, _, _ = port.raw_counts(tok, off, g, m, ca, threads=4)
    wb, _, _ = port.raw_counts(tok, off, g, m, cb, threads=4)
    single = _native.Engine(g, m, lib=rccl, path=1)
    single.load_sequences(tok, off, N, 0)
    single.accumulate(ca)
    single.finalize()
    want = single.counts_digest()
    single.close()
    e = _native.Engine(g, m, lib=rccl, devices=list(range(R)), collective=_native.COLL_RCCL, bands=5, path=1)
    e.load_sequences(tok, off, N, 0)
    e.accumulate(ca)
    e.finalize()
    info = e.multi_info()
    assert info["bands"] == 5 and info["narrow"] and info["collective"] == "rccl"
    assert e.counts_digest() == want and np.array_equal(e.get_counts(), wa)
    st = rccl_stats(rccl)
    assert st["int32"] == 5 * R and st["uint64"] == 0 and st["mismatched"] == 0 and st["wrong_device"] == 0
    e.accumulate(cb)                       # additive; three combos over R engines: some have none
    e.finalize()
    assert np.array_equal(e.get_counts(), wa + wb)
    # caller memory of unknown contents: the exchange stays 64 bits wide until a whole reset
    pairs = N * (N + 1) // 2
    big = np.full(pairs, (1 << 33) + 5, dtype=np.uint64)
    e.bind_counts(big.ctypes.data, pairs, keepalive=big)
    e.accumulate(ca)
    e.finalize()
    assert not e.multi_info()["narrow"] and rccl_stats(rccl)["uint64"] == 5 * R
    assert np.array_equal(big, wa + np.uint64((1 << 33) + 5))
    e.reset_counts()
    e.accumulate(cb)
    e.finalize()
    assert e.multi_info()["narrow"] and np.array_equal(big, wb)
    assert rccl_stats(rccl)["mismatched"] == 0
    e.close()


@pytest.mark.parametrize("devices,path", [([0, 1], 0), ([4, 5, 6], 2)])
def test_rccl_wide_cells_and_sparse_dataflow(rccl, port, devices, path):
    """C(g,m) * max_windows^2 >= 2^31: the band travels as ncclUint64, in place."""
    from fastsk_amd import _native
    rng = np.random.default_rng(21)
    X = [rng.integers(1, 5, size=n).astype(np.int32) for n in (20000, 50, 64, 41, 77, 58)]
    tok, off = _native.flatten(X)
    g, m = 6, 2
    want, _, _ = port.compute(tok, off, 4, 2, g, m, t=1)
    e = _native.Engine(g, m, lib=rccl, devices=devices, path=path, collective=_native.COLL_RCCL)
    e.compute(tok, off, 4, 2)
    assert not e.multi_info()["narrow"] and np.array_equal(e.get_triangle(), want)
    st = rccl_stats(rccl)
    assert st["uint64"] == len(devices) and st["int32"] == 0 and st["bytes"] == len(devices) * 8 * 21
    e.close()


@pytest.mark.parametrize("name", ["f4_ep300_skipvar_T3", "f6_prot219_skipvar16"])
def test_rccl_skip_variance(rccl, name):
    from fastsk_amd import _native
    d = load_golden(name)
    e = engine_for(rccl, d, [0, 1, 2, 3], collective=_native.COLL_RCCL)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_triangle(), d["tri"]) and np.array_equal(e.get_test(), d["test"])
    e.close()


@pytest.mark.parametrize("R,T", [(2, 1), (2, 2), (4, 2), (8, 5)])
def test_rccl_variance_chains(rccl, port, R, T):
    """Variance mode: chain t on engine t mod R, ONE ncclFloat64 all-reduce of the chains' sums; stdevs are chain 0's."""
    from fastsk_amd import _native
    rng = np.random.default_rng(5)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(12, 40, size=30)]
    tok, off = _native.flatten(X)
    g, m = 7, 3
    order = rng.permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tok, off, 22, 8, g, m, t=T, approx=True, delta=0.5, max_iters=6, order=order)
    e = _native.Engine(g, m, t=T, approx=True, delta=0.5, max_iters=6, lib=rccl, devices=list(range(R)), collective=_native.COLL_RCCL)
    e.set_combo_order(order)
    e.compute(tok, off, 22, 8)
    assert np.array_equal(e.get_stdevs(), sd)
    got = e.get_triangle()
    assert np.array_equal(got, want) if T <= 2 else np.allclose(got, want, rtol=1e-14, atol=0)
    st = rccl_stats(rccl)
    assert st["float64"] == R and st["int32"] == st["uint64"] == 0 and st["completed"] == 1 and st["wrong_device"] == 0
    e.close()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("deadline_ms", [400, -1])
def test_rccl_failed_collective_poisons_the_group(rccl, deadline_ms):
    """Rank 2's ncclAllReduce of band 1 fails (the others are already inside theirs): the call returns FSK_EDEVICE with RCCL's
    message and the band, the communicator is aborted (which releases the ranks that wait for rank 2 — with a deadline or
    without one: nobody sits it out), every later call repeats the first failure, destroying the handle does not hang."""
    import time
    from fastsk_amd import _native
    tok, off = synthetic_dna(400, 26, seed=8)   # (400 sequences: three tile rows, so three bands)
    combos = np.arange(0, 20, dtype=np.int32)
    e = _native.Engine(6, 3, lib=rccl, devices=[0, 1, 2, 3], bands=3, deadline_ms=deadline_ms, collective=_native.COLL_RCCL)
    e.load_sequences(tok, off, 400, 0)
    rccl.L.emu_rccl_set_fault(2, 2, 1, 0)    # rank 2, its second all-reduce (band 1)
    t0 = time.perf_counter()
    with pytest.raises(_native.FskError) as ei:
        e.accumulate(combos)
        e.finalize()
    assert time.perf_counter() - t0 < 30
    assert ei.value.code == -4 and "ncclAllReduce failed" in str(ei.value) and "band 1" in str(ei.value), str(ei.value)
    st = rccl_stats(rccl)
    assert st["aborted"] == 4 and st["completed"] == 1   # (band 0 went through)
    with pytest.raises(_native.FskError) as ei2:
        e.accumulate(combos)
    assert "dead after an earlier failure" in str(ei2.value) and "ncclAllReduce failed" in str(ei2.value)
    with pytest.raises(_native.FskError):
        e.synchronize()
    e.close()
    assert rccl_stats(rccl)["destroyed"] == 0     # (aborted communicators are not destroyed a second time)
    # and a fresh group on the same devices works
    d = load_golden("f4_ep300_exact")
    e = engine_for(rccl, d, [0, 1, 2, 3], collective=_native.COLL_RCCL)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_counts(), d["counts"])
    e.close()


@pytest.mark.timeout(120)
def test_rccl_peer_that_never_answers(rccl):
    """Rank 1 never takes part in band 0's collective: the others' collectives give up (the library's own watchdog, 500 ms
    here), the group is poisoned and aborted — which is also what lets go of rank 1 —, FSK_EDEVICE, no hang."""
    from fastsk_amd import _native
    d = load_golden("f4_ep300_exact")
    e = _native.Engine(d["g"], d["m"], lib=rccl, devices=[0, 1, 2], bands=2, deadline_ms=5000, collective=_native.COLL_RCCL)
    e.load_sequences(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    rccl.L.emu_rccl_set_fault(3, 1, 0, 500)
    with pytest.raises(_native.FskError) as ei:
        e.accumulate(np.arange(0, 30, dtype=np.int32))
        e.finalize()
    assert ei.value.code == -4 and "band 0" in str(ei.value), str(ei.value)
    assert rccl_stats(rccl)["aborted"] == 3 and rccl_stats(rccl)["completed"] == 0
    e.close()


@pytest.mark.timeout(120)
def test_rccl_init_failure_and_init_deadline(rccl):
    """ncclCommInitAll fails -> fsk_create_multi fails with its message (nothing leaks); ncclCommInitAll that takes longer
    than the deadline -> FSK_EDEVICE naming the deadline, and the abandoned helper thread gives back the communicators it
    obtains afterwards."""
    import time
    from fastsk_amd import _native
    rccl.L.emu_rccl_set_fault(1, 0, 0, 0)
    with pytest.raises(_native.FskError, match="ncclCommInitAll failed"):
        _native.Engine(6, 2, lib=rccl, devices=[0, 1], collective=_native.COLL_RCCL)
    rccl.L.emu_rccl_set_fault(4, 0, 1200, 0)   # the init takes 1.2 s
    t0 = time.perf_counter()
    with pytest.raises(_native.FskError, match="did not return within 300 ms"):
        _native.Engine(6, 2, lib=rccl, devices=[0, 1, 2], collective=_native.COLL_RCCL, deadline_ms=300)
    assert time.perf_counter() - t0 < 1.0
    time.sleep(1.6)
    st = rccl_stats(rccl)
    assert st["comms"] == 3 and st["aborted"] == 3 and st["destroyed"] == 0, st
    e = _native.Engine(6, 2, lib=rccl, devices=[0, 1, 2], collective=_native.COLL_RCCL, deadline_ms=300)   # and the next one is fine
    e.close()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("collective", ["rccl", "p2p"])
def test_group_variance_chains_may_outlast_the_deadline(rccl, port, collective):
    """Variance mode: the chains of one engine legitimately take much longer than another's (here engine 1 is 1.5 s late,
    the deadline is 300 ms): the barrier in front of the fp64 all-reduce has NO deadline — the result is the usual one,
    not FSK_EDEVICE 'another engine of the group failed' on a healthy machine."""
    from fastsk_amd import _native
    rng = np.random.default_rng(5)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(12, 40, size=30)]
    tok, off = _native.flatten(X)
    g, m, T = 7, 3, 2
    order = rng.permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tok, off, 22, 8, g, m, t=T, approx=True, delta=0.5, max_iters=6, order=order)
    coll = _native.COLL_RCCL if collective == "rccl" else _native.COLL_P2P
    e = _native.Engine(g, m, t=T, approx=True, delta=0.5, max_iters=6, lib=rccl, devices=[0, 1], collective=coll, deadline_ms=300,
                       tuning={"fault_kind": 3, "fault_rank": 1, "fault_ms": 1500})
    e.set_combo_order(order)
    e.compute(tok, off, 22, 8)
    assert np.array_equal(e.get_stdevs(), sd) and np.array_equal(e.get_triangle(), want)
    e.close()


def test_tuning_through_the_abi_and_the_environment(emu_lib, monkeypatch):
    """fsk_set_tuning / fsk_get_tuning / FSK_TUNING: unknown keys and values out of range fail loudly, a group hands a key to
    all its engines, the test-only keys exist in this (test) build."""
    from fastsk_amd import _native
    keys = emu_lib.tuning_keys()
    assert "fault_kind" in keys and keys["fault_kind"][:3] == (0, 0, 3)
    e = _native.Engine(6, 2, lib=emu_lib, devices=[0, 1], tuning={"sparse_global": 1})
    assert e.get_tuning("sparse_global") == 1 and e.get_tuning("guard_cap") == 0
    with pytest.raises(_native.FskError, match="unknown tuning key"):
        e.set_tuning("no_such_key", 1)
    with pytest.raises(_native.FskError, match="takes 0 .. 1"):
        e.set_tuning("sparse_global", 7)
    e.close()
    monkeypatch.setenv("FSK_TUNING", "guard_cap=64, sparse_sync=1")
    e = _native.Engine(6, 2, lib=emu_lib)
    assert e.get_tuning("guard_cap") == 64 and e.get_tuning("sparse_sync") == 1
    e.close()
    for bad in ("guard_cap", "nonsense=1", "guard_cap=abc", "sparse_sync=9"):
        monkeypatch.setenv("FSK_TUNING", bad)
        with pytest.raises(_native.FskError, match="FSK_TUNING"):
            _native.Engine(6, 2, lib=emu_lib)

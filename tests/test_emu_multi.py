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

from conftest import load_golden, ROOT

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
                                          ("f5_prot11_exact", [0, 1, 2])])
def test_group_exact_equals_golden(emu_lib, name, devices):
    d = load_golden(name)
    e = engine_for(emu_lib, d, devices)
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
    with pytest.raises(_native.FskError, match="RCCL"):
        _native.Engine(5, 2, lib=emu_lib, devices=[0, 1], collective=_native.COLL_RCCL)
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
    """Engine 1's worker is late by 2.5 s before the collective of band 0 (FSK_FAULT, test-only): with a 300 ms
    deadline the other engines give up at that band's exchange, the call returns FSK_EDEVICE naming the band well
    before a run without a deadline would have returned, every later call repeats the first failure (the group is
    dead, never half alive), and destroying the handle does not hang."""
    import time
    from fastsk_amd import _native
    d = load_golden("f4_ep300_exact")
    monkeypatch.setenv("FSK_FAULT", "host:1:0:2500")
    e = _native.Engine(d["g"], d["m"], lib=emu_lib, devices=[0, 1, 2], bands=3, deadline_ms=300)
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
    monkeypatch.delenv("FSK_FAULT", raising=False)
    e = engine_for(emu_lib, d, [0, 1, 2], bands=3, deadline_ms=20000)
    e.compute(d["tokens"], d["offsets"], d["n_train"], d["n_test"])
    assert np.array_equal(e.get_counts(), d["counts"])
    e.close()
    # a late engine INSIDE the deadline only delays the result
    monkeypatch.setenv("FSK_FAULT", "host:2:0:300")
    e = engine_for(emu_lib, d, [0, 1, 2], bands=2, deadline_ms=20000)
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

"""The N>1 host path on the CPU: world_size 2 over gloo, emulated engine, against the oracle.
Checks the combo partition, the banded all-reduce (uint64 and narrowed int32 payloads) and that
every rank normalises the same reduced triangle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT, load_golden


def test_shard_partition():
    from fastsk_amd.distributed import shard, band_edges, cell
    combos = np.arange(495)
    parts = [shard(combos, r, 8) for r in range(8)]
    assert sorted(np.concatenate(parts).tolist()) == combos.tolist()
    assert [len(p) for p in parts] == [62, 62, 62, 62, 62, 62, 62, 61]
    assert shard(combos, 3, 8)[:3].tolist() == [3, 11, 19]
    e = band_edges(100000, 8)
    assert e[0] == 0 and e[-1] == 100000 and all(x % 128 == 0 for x in e[:-1]) and len(e) == 9
    areas = np.diff([cell(x) for x in e])
    assert areas.max() / areas.min() < 1.05  # equal-area bands (edges rounded to tile rows)
    assert band_edges(300, 8) == [0, 128, 256, 300] or band_edges(300, 8)[-1] == 300


def run_world(tmp_path, fixture, n_bands, narrow, port_no, shard_by="combos", replicate=True, world=2):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no), WORLD_SIZE=str(world))
    procs = []
    for rank in range(world):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), fixture,
                                       str(tmp_path), str(n_bands), str(int(narrow)), "cpu", shard_by,
                                       str(int(replicate))], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(outs)
    return [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]


def test_two_rank_gloo_all_reduce(tmp_path):
    name = "f3_ragged_sigma7_g6m3"
    d = load_golden(name)
    done = 0
    for z in run_world(tmp_path, os.path.join(GOLD, name + ".npz"), 1, False, 29611):
        assert int(z["world"]) == 2
        assert np.array_equal(z["counts"], d["counts"])  # identical on every rank after the all-reduce
        assert np.array_equal(z["tri"], d["tri"])
        done += int(z["done"])
    assert done == len(d["combos"])  # every combo processed exactly once across the ranks


@pytest.mark.parametrize("narrow", [False, True])
def test_two_rank_banded_overlapped_reduce(tmp_path, port, narrow):
    """Several row bands (N > 128) with the int32-narrowed and the uint64 payload."""
    rng = np.random.default_rng(8)
    N = 300
    X = rng.integers(1, 5, size=(N, 40), dtype=np.int32)
    tokens, offsets = X.reshape(-1), np.arange(N + 1, dtype=np.int64) * 40
    combos = np.arange(0, 70, 5, dtype=np.int32)
    fx = tmp_path / "in.npz"
    np.savez(fx, tokens=tokens, offsets=offsets, n_train=200, n_test=100, g=8, m=4, combos=combos)
    want, _, _ = port.raw_counts(tokens, offsets, 8, 4, combos, threads=4)
    for z in run_world(tmp_path, str(fx), 3, narrow, 29613 + int(narrow)):
        assert np.array_equal(z["counts"], want)
        assert np.array_equal(z["tri"], port.normalise(want.astype(np.float64), N))


def test_owner_edges():
    from fastsk_amd.distributed import owner_edges, sub_edges, cell
    e = owner_edges(100000, 8)
    assert len(e) == 9 and e[0] == 0 and e[-1] == 100000
    assert owner_edges(300, 8) is None and owner_edges(300, 2) == [0, 256, 300]
    s = sub_edges(e[3], e[4], 4)
    assert s[0] == e[3] and s[-1] == e[4] and all(x % 128 == 0 for x in s[:-1])
    areas = np.diff([cell(x) for x in s])
    assert areas.max() / areas.min() < 1.2


@pytest.mark.parametrize("replicate,narrow", [(False, False), (True, False), (True, True)])
def test_two_rank_row_sharded(tmp_path, port, replicate, narrow):
    """shard_by="rows": each rank runs all combos over its own rows; only the diagonal is exchanged
    (or the finished bands are broadcast when replicate=True)."""
    from fastsk_amd.distributed import owner_edges, cell
    rng = np.random.default_rng(9)
    N = 420
    X = rng.integers(1, 5, size=(N, 36), dtype=np.int32)
    tokens, offsets = X.reshape(-1), np.arange(N + 1, dtype=np.int64) * 36
    combos = np.arange(0, 70, 6, dtype=np.int32)
    fx = tmp_path / "in.npz"
    np.savez(fx, tokens=tokens, offsets=offsets, n_train=300, n_test=120, g=8, m=4, combos=combos)
    want, _, _ = port.raw_counts(tokens, offsets, 8, 4, combos, threads=4)
    tri = port.normalise(want.astype(np.float64), N)
    full = np.zeros((N, N))
    il = np.tril_indices(N)
    full[il] = tri
    full = full + full.T - np.diag(np.diag(full))
    zs = run_world(tmp_path, str(fx), 2, narrow, 29631 + 2 * int(replicate) + int(narrow), "rows", replicate)
    edges = owner_edges(N, 2)
    diag = np.array([cell(i) + i for i in range(N)])
    for r, z in enumerate(zs):
        assert np.array_equal(z["full"], full)  # assembled block: bit-identical normalised kernel on every rank
        if replicate:
            assert np.array_equal(z["counts"], want)
        else:
            lo, hi = cell(edges[r]), cell(edges[r + 1])
            assert np.array_equal(z["counts"][lo:hi], want[lo:hi])  # own rows complete
            assert np.array_equal(z["counts"][diag], want[diag])    # diagonal exchanged
            other = np.ones(len(want), bool)
            other[lo:hi] = False
            other[diag] = False
            assert not z["counts"][other].any()                      # nothing else touched


@pytest.mark.parametrize("shard_by", ["rows", "combos"])
def test_four_ranks(tmp_path, port, shard_by):
    """World size 4 (nothing in the host path may be specific to two ranks): row ownership without
    replication, and the banded all-reduce."""
    from fastsk_amd.distributed import owner_edges
    rng = np.random.default_rng(10)
    N = 600
    assert owner_edges(N, 4) == [0, 256, 384, 512, 600]
    X = rng.integers(1, 5, size=(N, 30), dtype=np.int32)
    tokens, offsets = X.reshape(-1), np.arange(N + 1, dtype=np.int64) * 30
    combos = np.array([0, 17, 33, 50, 69], dtype=np.int32)
    fx = tmp_path / "in.npz"
    np.savez(fx, tokens=tokens, offsets=offsets, n_train=N, n_test=0, g=8, m=4, combos=combos)
    want, _, _ = port.raw_counts(tokens, offsets, 8, 4, combos, threads=4)
    tri = port.normalise(want.astype(np.float64), N)
    il = np.tril_indices(N)
    zs = run_world(tmp_path, str(fx), 2, True, 29641 + (shard_by == "rows"), shard_by, False, world=4)
    for z in zs:
        assert int(z["world"]) == 4
        assert np.array_equal(z["full"][il], tri)
        if shard_by == "combos":
            assert np.array_equal(z["counts"], want)


def run_variance_world(tmp_path, fixture, port_no, world, device="cpu"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no), WORLD_SIZE=str(world))
    procs = []
    for rank in range(world):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_variance_worker.py"), fixture,
                                       str(tmp_path), device], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]


def variance_case(tmp_path, port, T):
    """Small ragged DNA set, variance mode with T chains, and the oracle's result (chains summed in order)."""
    from fastsk_amd import _native
    rng = np.random.default_rng(17)
    X = [rng.integers(1, 5, size=int(L)).astype(np.int32) for L in rng.integers(12, 40, size=30)]
    tok, off = _native.flatten(X)
    g, m, delta, max_iters = 7, 3, 0.3, 9
    order = rng.permutation(port.num_combos(g, m)).astype(np.int32)
    want, sd, _ = port.compute(tok, off, 22, 8, g, m, t=T, approx=True, delta=delta, max_iters=max_iters, order=order)
    fixture = str(tmp_path / "variance_case.npz")
    np.savez(fixture, tokens=tok, offsets=off, n_train=22, n_test=8, g=g, m=m, t=T, delta=delta, max_iters=max_iters, order=order)
    return fixture, want, sd


@pytest.mark.parametrize("T,world", [(2, 2), (5, 2), (3, 4)])
def test_variance_mode_chains_over_ranks(tmp_path, port, T, world):
    """SURVEY 8e, approx variance mode: chain c on rank c mod R, one fp64 all-reduce of the K_hat sums.
    stdevs are chain 0's, bit for bit; the kernel is the oracle's to the bit when the sum has two terms
    (fp64 addition commutes) and to rounding otherwise (the reference's own threads add in arrival order)."""
    fixture, want, sd = variance_case(tmp_path, port, T)
    for z in run_variance_world(tmp_path, fixture, 29650 + T, world):
        assert np.array_equal(z["stdevs"], sd)
        if T == 2:
            assert np.array_equal(z["tri"], want)
        else:
            assert np.allclose(z["tri"], want, rtol=1e-14, atol=0)


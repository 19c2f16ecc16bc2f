"""Multi-GPU host path: one process per GPU, combos sharded, ONE all-reduce of the integer triangle.

The C(g,m) mismatch combinations are independent units (the reference shards them round-robin over
host threads, ``fastsk_kernel.cpp:148,275``, and sum-reduces once, ``:286-315``). Here rank r of R
takes ``combos[r::R]``, accumulates a private uint64 triangle on its GPU through the C ABI, and
``torch.distributed.all_reduce`` (backend "nccl" = RCCL over xGMI) sums the partial triangles.
Integer sums are order independent, so the result is bit-identical for every R. Every rank ends up
with the full triangle and can serve normalised blocks.

The one logical all-reduce is issued in row bands: xGMI is point to point, so a ring all-reduce of
the 40 GB triangle of the 100k-sequence workload is bound by a single link and costs about as much
as a GPU's share of the counting. ``fsk_accumulate_rows`` finishes the triangle band by band (equal
tile counts per band), and the all-reduce of band b runs on RCCL's stream while the engine's
stream accumulates band b+1. When every reduced cell provably fits 31 bits
(``C(g,m) * max_windows^2 < 2^31``) the band travels as int32, halving the bytes on the links.
"""
import math

import numpy as np

from . import _native

TILE = 128


def shard(combos, rank, world):
    """Round-robin partition: rank r takes combos[r], combos[r+R], ... (sizes differ by <= 1)."""
    return np.ascontiguousarray(np.asarray(combos, dtype=np.int32)[rank::world])


def band_edges(N, n_bands):
    """Row boundaries (multiples of 128, last = N) that split the lower triangle into bands of
    about equal area: r_b = N * sqrt(b / n_bands)."""
    edges = {0, N}
    for b in range(1, n_bands):
        r = int(round(N * math.sqrt(b / n_bands) / TILE)) * TILE
        if 0 < r < N:
            edges.add(r)
    return sorted(edges)


def cell(i):
    return i * (i + 1) // 2


def accumulate_and_reduce(eng, K, combos, group=None, n_bands=None, narrow=None, n_combos_total=None, force=False):
    """One pass: accumulate this rank's ``combos`` into the bound triangle ``K`` and sum it over
    the ranks of ``group``, band by band, overlapping RCCL with the next band's kernels."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    N = eng.N
    if world == 1 and not force:  # force: run the banded path on one rank (smoke test of this code)
        eng.accumulate(combos)
        eng.synchronize()
        return
    dense = eng.stats()["path_used"] == _native.PATH_DENSE
    if n_bands is None:
        n_bands = 8 if (dense and N >= 8192) else 1
    if narrow is None:
        total = n_combos_total if n_combos_total is not None else eng.lib.num_combos(eng.g, eng.m)
        narrow = total * eng.stats()["max_windows"] ** 2 < 2 ** 31
    edges = band_edges(N, n_bands)
    pending = []

    def drain(keep):
        while len(pending) > keep:
            work, seg, seg32 = pending.pop(0)
            work.wait()
            if seg32 is not None:
                seg.copy_(seg32)  # widen back into the uint64 triangle

    for lo, hi in zip(edges[:-1], edges[1:]):
        eng.accumulate_rows(combos, lo, hi)
        eng.synchronize()  # the engine has its own HIP stream; RCCL is ordered after the band is final
        seg = K[cell(lo):cell(hi)]
        seg32 = seg.to(torch.int32) if narrow else None
        work = dist.all_reduce(seg32 if narrow else seg, op=dist.ReduceOp.SUM, group=group, async_op=True)
        pending.append((work, seg, seg32))
        drain(2)
    drain(0)
    if K.is_cuda:
        torch.cuda.synchronize(K.device)


def compute_sharded(tokens, offsets, n_train, n_test, g, m, combos=None, group=None, device=None, lib=None,
                    path=_native.PATH_AUTO, profile=False, n_bands=None, narrow=None):
    """Exact (or explicit-combo-list) kernel over the ranks of ``group``.

    Returns ``(engine, K)``: the finalized engine of this rank (use ``get_block`` / ``get_train`` /
    ``get_test``) and the torch tensor that holds the reduced integer triangle (int64 view of the
    uint64 counts). ``device`` is a torch device (defaults to the current CUDA device); ``lib`` is
    only overridden by the CPU test-suite, which runs this same code over gloo against the
    emulated library.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    eng = _native.Engine(g, m, device=(device.index or 0) if device.type == "cuda" else 0, path=path,
                         profile=profile, lib=lib)
    ncomb = eng.lib.num_combos(g, m)
    if combos is None:
        combos = np.arange(ncomb, dtype=np.int32)
    N = n_train + n_test
    pairs = N * (N + 1) // 2
    K = torch.zeros(pairs, dtype=torch.int64, device=device)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    eng.bind_counts(K.data_ptr(), pairs, keepalive=K)
    eng.load_sequences(tokens, offsets, n_train, n_test)
    accumulate_and_reduce(eng, K, shard(combos, rank, world), group=group, n_bands=n_bands, narrow=narrow,
                          n_combos_total=len(combos))
    eng.finalize()
    return eng, K

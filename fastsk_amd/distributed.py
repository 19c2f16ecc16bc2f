"""Multi-GPU host path: one process per GPU over torch.distributed (RCCL), two ways to shard.

``shard="rows"`` (large N): rank r OWNS an equal-area band of rows of the triangle and runs ALL
combos over it (``fsk_accumulate_rows``). No cell is shared, so the only exchange is the N-entry
diagonal the cosine normalisation needs (0.8 MB at N = 100k); the kernel matrix stays distributed
(each rank serves its rows; ``get_block_distributed`` assembles any rectangle) unless
``replicate=True`` broadcasts the finished bands under the next band's kernels.

``shard="combos"`` (small N, and the reference's own decomposition):

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


def owner_edges(N, world):
    """Row ownership for shard="rows": world+1 tile-aligned edges of equal-area bands, or None when
    N is too small to give every rank a band."""
    e = band_edges(N, world)
    return e if len(e) == world + 1 else None


def sub_edges(lo, hi, n):
    """Equal-area split of rows [lo, hi) of the lower triangle into <= n tile-aligned sub-bands."""
    e = {lo, hi}
    for k in range(1, n):
        r = int(round(math.sqrt(lo * lo + (hi * hi - lo * lo) * k / n) / TILE)) * TILE
        if lo < r < hi:
            e.add(r)
    return sorted(e)


def diag_index(N, device):
    import torch
    i = torch.arange(N, dtype=torch.int64, device=device)
    return i * (i + 3) // 2  # tri_index(i, i)


def accumulate_owned_rows(eng, K, combos, group=None, replicate=False, n_sub=None, narrow=None, edges=None):
    """One pass of shard="rows": all ``combos`` over this rank's band of rows, then the diagonal
    exchange. With ``replicate`` every finished sub-band is broadcast from its owner (int32 when it
    provably fits) while the next sub-band is computed, so every rank ends with the whole triangle.
    Returns this rank's (row_begin, row_end)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    N = eng.N
    if edges is None:
        edges = owner_edges(N, world)
    if edges is None:
        raise ValueError("shard='rows' needs at least one 128-row tile band per rank")
    lo, hi = edges[rank], edges[rank + 1]
    if world == 1:
        eng.accumulate_rows(combos, lo, hi)
        eng.synchronize()
        return lo, hi
    if not replicate:
        eng.accumulate_rows(combos, lo, hi)
        eng.synchronize()
        idx = diag_index(N, K.device)
        d = K[idx]
        d[:lo] = 0  # entries of other ranks' rows may still hold the previous pass's exchange
        d[hi:] = 0
        dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
        K[idx] = d
        if K.is_cuda:
            torch.cuda.synchronize(K.device)
        return lo, hi
    if n_sub is None:
        n_sub = 4
    if narrow is None:
        narrow = len(combos) * eng.stats()["max_windows"] ** 2 < 2 ** 31
    subs = [sub_edges(edges[r], edges[r + 1], n_sub) for r in range(world)]
    side = _side_stream(K)
    pending = []

    def drain(keep):
        while len(pending) > keep:
            work, seg, seg32, mine = pending.pop(0)
            work.wait()
            if seg32 is not None and not mine:
                seg.copy_(seg32)

    for k in range(max(len(s) for s in subs) - 1):
        if k < len(subs[rank]) - 1:
            eng.accumulate_rows(combos, subs[rank][k], subs[rank][k + 1])
            if side is None:
                eng.synchronize()
            else:
                eng.stream_wait_engine(side.cuda_stream)  # the sub-band is final before its broadcast reads it
        with _on(side):
            for r in range(world):  # same order on every rank
                if k >= len(subs[r]) - 1:
                    continue
                seg = K[cell(subs[r][k]):cell(subs[r][k + 1])]
                seg32 = None
                if narrow:
                    seg32 = seg.to(torch.int32) if r == rank else torch.empty(seg.shape, dtype=torch.int32, device=K.device)
                src = dist.get_global_rank(group, r) if group is not None else r
                work = dist.broadcast(seg32 if narrow else seg, src=src, group=group, async_op=True)
                pending.append((work, seg, seg32, r == rank))
            drain(2 * world)
    with _on(side):
        drain(0)
    if side is not None:
        eng.engine_wait_stream(side.cuda_stream)
        side.synchronize()
    eng.synchronize()
    return lo, hi


def get_block_distributed(eng, i0, i1, j0, j1, group=None, device=None):
    """Normalised block [i0,i1) x [j0,j1) of a row-sharded kernel, on every rank: an off-diagonal
    cell is non-zero only on the owner of row max(i,j), the (exchanged) diagonal is the same
    everywhere and nothing is negative, so an element-wise MAX assembles the block exactly."""
    import torch
    import torch.distributed as dist

    if device is not None and torch.device(device).type == "cuda":
        blk = eng.get_block_torch(i0, i1, j0, j1)
    else:
        blk = torch.from_numpy(eng.get_block(i0, i1, j0, j1))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(blk, op=dist.ReduceOp.MAX, group=group)
    return blk


_side_streams = {}


def _side_stream(K):
    """A torch stream of its own for the exchange (narrowing copies, RCCL collectives — ordered behind
    the stream they are issued on — and widening copies): torch's default stream is HIP's legacy
    stream, which serialises against every other blocking stream, the engine's included; on a side
    stream the copies run under the engine's tile kernels (VALU-bound: the HBM bandwidth is free).
    None for a host tensor."""
    import torch
    if not K.is_cuda:
        return None
    key = (K.device.type, K.device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=K.device)
    return _side_streams[key]


class _on:
    """`with _on(stream):` = torch.cuda.stream(stream), or nothing for a host tensor."""

    def __init__(self, stream):
        self.stream, self.ctx = stream, None

    def __enter__(self):
        if self.stream is not None:
            import torch
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def accumulate_and_reduce(eng, K, combos, group=None, n_bands=None, narrow=None, n_combos_total=None, force=False):
    """One pass: accumulate this rank's ``combos`` into the bound triangle ``K`` and sum it over
    the ranks of ``group``, band by band, overlapping RCCL with the next band's kernels. The host
    never waits for a band: torch's stream (and RCCL behind it) waits for an event the engine
    records after the band's last kernel. Returns when ``K`` holds the reduced triangle."""
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
        # more, smaller bands leave less of the last band's exchange exposed — as long as a band's tile
        # launch keeps one workgroup per tile (>= 16384 tiles), the form that stores instead of adding
        tiles = ((N + TILE - 1) // TILE) * ((N + TILE - 1) // TILE + 1) // 2
        n_bands = (16 if tiles >= 16 * 16384 else 8) if (dense and N >= 8192) else 1
    if narrow is None:
        total = n_combos_total if n_combos_total is not None else eng.lib.num_combos(eng.g, eng.m)
        narrow = total * eng.stats()["max_windows"] ** 2 < 2 ** 31
    edges = band_edges(N, n_bands)
    side = _side_stream(K)
    pending = []

    def drain(keep):
        while len(pending) > keep:
            work, seg, seg32 = pending.pop(0)
            work.wait()  # (the current stream waits, not the host)
            if seg32 is not None:
                seg.copy_(seg32)  # widen back into the uint64 triangle

    for lo, hi in zip(edges[:-1], edges[1:]):
        eng.accumulate_rows(combos, lo, hi)
        if side is None:
            eng.synchronize()
        else:
            eng.stream_wait_engine(side.cuda_stream)  # the band is final before anything torch / RCCL does to it
        with _on(side):
            seg = K[cell(lo):cell(hi)]
            seg32 = seg.to(torch.int32) if narrow else None
            work = dist.all_reduce(seg32 if narrow else seg, op=dist.ReduceOp.SUM, group=group, async_op=True)
            pending.append((work, seg, seg32))
            drain(2)
    with _on(side):
        drain(0)
    if side is not None:
        eng.engine_wait_stream(side.cuda_stream)  # the engine's next pass starts after the reduced cells are in place
        side.synchronize()
    eng.synchronize()  # (fills rows a reset left for a storing launch that never came: none on this path)


def compute_sharded(tokens, offsets, n_train, n_test, g, m, combos=None, group=None, device=None, lib=None,
                    path=_native.PATH_AUTO, profile=False, n_bands=None, narrow=None, shard_by="combos",
                    replicate=True):
    """Exact (or explicit-combo-list) kernel over the ranks of ``group``.

    Returns ``(engine, K)``: the finalized engine of this rank (use ``get_block`` / ``get_train`` /
    ``get_test``) and the torch tensor that holds the integer triangle (int64 view of the uint64
    counts). ``shard_by="combos"``: combos dealt round-robin, triangle all-reduced, every rank holds
    all of it. ``shard_by="rows"``: every rank runs all combos over its own band of rows
    (``owner_edges``); with ``replicate=False`` the triangle stays distributed (cells outside the
    rank's rows are zero; assemble blocks with ``get_block_distributed``). "rows" falls back to
    "combos" when N is too small for one tile band per rank or the sparse dataflow is in use. ``device`` is a torch device (defaults to the current CUDA device); ``lib`` is
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
    if shard_by not in ("combos", "rows"):
        raise ValueError("shard_by must be 'combos' or 'rows'")
    dense = eng.stats()["path_used"] == _native.PATH_DENSE  # (the sparse dataflow would repeat its sort on every rank)
    if shard_by == "rows" and dense and owner_edges(N, world) is not None and world > 1:
        accumulate_owned_rows(eng, K, np.ascontiguousarray(combos, dtype=np.int32), group=group, replicate=replicate,
                              n_sub=n_bands, narrow=narrow)
    else:
        accumulate_and_reduce(eng, K, shard(combos, rank, world), group=group, n_bands=n_bands, narrow=narrow,
                              n_combos_total=len(combos))
    eng.finalize()
    return eng, K


def compute_variance_sharded(tokens, offsets, n_train, n_test, g, m, t, delta=0.025, max_iters=-1, order=None, seed=None,
                             group=None, device=None, lib=None, path=_native.PATH_AUTO):
    """Variance (convergence) mode over the ranks of ``group``: the ``t`` Welford chains of the
    reference's worker threads (``fastsk_kernel.cpp:188-281``) are the units — chain c runs on rank
    ``c mod R`` — and their K_hat are summed in fp64 with ONE all-reduce (``fastsk_kernel.cpp:286-315``
    adds them under locks, in whatever order the threads arrive; here the order is the collective's).
    Every rank needs the same combo order: pass ``order`` or ``seed``.

    Returns ``(engine, stdevs)``: the finalized engine of this rank (every rank holds the whole
    kernel) and chain 0's convergence trace, broadcast from the rank that ran it. With ``t <= R`` some
    ranks have no chain and only take part in the sum."""
    import torch
    import torch.distributed as dist

    distributed_run = dist.is_initialized()
    world = dist.get_world_size(group) if distributed_run else 1
    rank = dist.get_rank(group) if distributed_run else 0
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    eng = _native.Engine(g, m, t=t, approx=True, delta=delta, max_iters=max_iters,
                         device=(device.index or 0) if device.type == "cuda" else 0, path=path, lib=lib)
    if order is not None:
        eng.set_combo_order(order)
    elif seed is not None:
        eng.set_seed(seed)
    elif world > 1:
        raise ValueError("every rank needs the same combo order: pass order= or seed=")
    eng.load_sequences(tokens, offsets, n_train, n_test)
    eng.run_chains(rank, world)
    if world > 1:
        N = n_train + n_test
        total = torch.empty(N * (N + 1) // 2, dtype=torch.float64, device=device)
        eng.get_kernel_sum_device(total.data_ptr())
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        eng.set_kernel_sum_device(total.data_ptr())
    eng.finalize()
    sd = [eng.get_stdevs() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(sd, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return eng, np.asarray(sd[0], dtype=np.float64)


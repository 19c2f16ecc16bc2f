"""Multi-GPU host path: one process per GPU, combos sharded, ONE all-reduce of the integer triangle.

The C(g,m) mismatch combinations are independent units (the reference shards them round-robin over
host threads, ``fastsk_kernel.cpp:148,275``, and sum-reduces once, ``:286-315``). Here rank r of R
takes ``combos[r::R]``, accumulates a private uint64 triangle on its GPU through the C ABI, and a
single ``torch.distributed.all_reduce`` (backend "nccl" = RCCL over xGMI) sums the partial
triangles. Integer sums are order independent, so the result is bit-identical for every R.
Every rank ends up with the full triangle and can serve normalised blocks.
"""
import numpy as np

from . import _native


def shard(combos, rank, world):
    """Round-robin partition: rank r takes combos[r], combos[r+R], ... (sizes differ by <= 1)."""
    return np.ascontiguousarray(np.asarray(combos, dtype=np.int32)[rank::world])


def compute_sharded(tokens, offsets, n_train, n_test, g, m, combos=None, group=None, device=None, lib=None,
                    path=_native.PATH_AUTO, profile=False):
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
    eng = _native.Engine(g, m, device=device.index or 0 if device.type == "cuda" else 0, path=path, profile=profile,
                         lib=lib)
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
    eng.accumulate(shard(combos, rank, world))
    eng.synchronize()  # the engine runs on its own HIP stream; RCCL runs on torch's
    if world > 1:
        dist.all_reduce(K, op=dist.ReduceOp.SUM, group=group)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
    eng.finalize()
    return eng, K

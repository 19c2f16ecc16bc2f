"""Worker for tests/test_distributed.py: one rank of a world_size-2 gloo job on the CPU, running
fastsk_amd.distributed.compute_sharded against the emulated engine library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def main():
    import torch
    import torch.distributed as dist
    from fastsk_amd import _native, distributed
    import build_emu

    fixture, outdir = sys.argv[1], sys.argv[2]
    device = torch.device(sys.argv[5] if len(sys.argv) > 5 else "cpu")
    dist.init_process_group("gloo")  # gloo also reduces CUDA tensors: two ranks may share one GPU
    rank, world = dist.get_rank(), dist.get_world_size()
    # CPU: the emulated engine (test-only). CUDA: the product library, real HIP kernels.
    lib = _native.Library(build_emu.build()) if device.type == "cpu" else _native.library()
    d = np.load(fixture)
    eng, K = distributed.compute_sharded(d["tokens"], d["offsets"], int(d["n_train"]), int(d["n_test"]), int(d["g"]),
                                         int(d["m"]), combos=d["combos"], device=device, lib=lib,
                                         n_bands=int(sys.argv[3]), narrow=bool(int(sys.argv[4])),
                                         shard_by=sys.argv[6] if len(sys.argv) > 6 else "combos",
                                         replicate=bool(int(sys.argv[7])) if len(sys.argv) > 7 else True)
    N = int(d["n_train"]) + int(d["n_test"])
    if len(sys.argv) > 7 and sys.argv[6] == "rows" and not int(sys.argv[7]):
        # a second pass on the same engine, zeroing only the owned rows: the diagonal entries of
        # the other ranks' rows still hold the first pass's exchange and must not be summed again
        lo, hi = distributed.owner_edges(N, world)[rank:rank + 2]
        eng.reset_counts_rows(lo, hi)
        distributed.accumulate_owned_rows(eng, K, d["combos"].astype(np.int32))
        eng.finalize()
    full = distributed.get_block_distributed(eng, 0, N, 0, N, device=device).cpu().numpy()  # == local block when replicated
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), counts=K.cpu().numpy().view(np.uint64), tri=eng.get_triangle(),
             done=eng.stats()["combos_done"], world=world, full=full)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Worker for tests/test_distributed.py: one rank of a gloo job running
fastsk_amd.distributed.compute_variance_sharded (the Welford chains dealt over the ranks) — on the CPU
against the emulated engine library, or on a GPU (ranks may share one) against the product library."""
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

    fixture, outdir, device = sys.argv[1], sys.argv[2], torch.device(sys.argv[3])
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    if device.type == "cpu":
        import build_emu
        lib = _native.Library(build_emu.build())
    else:
        lib = _native.library()
    d = np.load(fixture)
    eng, sd = distributed.compute_variance_sharded(d["tokens"], d["offsets"], int(d["n_train"]), int(d["n_test"]), int(d["g"]),
                                                   int(d["m"]), int(d["t"]), delta=float(d["delta"]), max_iters=int(d["max_iters"]),
                                                   order=d["order"], device=device, lib=lib)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), tri=eng.get_triangle(), stdevs=sd, done=eng.stats()["combos_done"])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

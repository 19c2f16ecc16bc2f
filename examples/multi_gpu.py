#!/usr/bin/env python3
"""Multi-GPU kernel: one process per GPU over RCCL — exact mode sharded by row bands (or combos), or
the approximate (variance / convergence) mode with its Welford chains dealt over the GPUs.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/multi_gpu.py --n-seq 100000 --seq-len 300 -g 12 -m 8

--shard rows (default): every rank runs all combos over its own band of rows, the kernel stays
distributed and blocks are assembled on demand; --shard combos: combos dealt round-robin and the
triangle all-reduced. --approx T: variance mode with T chains (chain c on GPU c mod R, one fp64
all-reduce of their sums, stdevs from chain 0). Rank 0 prints a corner of the normalised kernel.
See fastsk_amd/distributed.py.
"""
import argparse
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between processes
import torch  # noqa: E402
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastsk_amd import distributed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-seq", type=int, default=20000)
    ap.add_argument("--seq-len", type=int, default=300)
    ap.add_argument("-g", type=int, default=12)
    ap.add_argument("-m", type=int, default=8)
    ap.add_argument("--shard", choices=["rows", "combos"], default="rows")
    ap.add_argument("--approx", type=int, default=0, metavar="T", help="variance mode with T chains instead of the exact kernel")
    ap.add_argument("--delta", type=float, default=0.025)
    args = ap.parse_args()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rng = np.random.Generator(np.random.PCG64(20201214))
    X = rng.integers(1, 5, size=(args.n_seq, args.seq_len), dtype=np.int32)
    tokens, offsets = X.reshape(-1), np.arange(args.n_seq + 1, dtype=np.int64) * args.seq_len
    if args.approx:
        t0 = time.time()
        eng, sd = distributed.compute_variance_sharded(tokens, offsets, args.n_seq, 0, args.g, args.m, args.approx, delta=args.delta,
                                                       seed=20201214)  # (every rank must draw the same combo order)
        dt = time.time() - t0
        if not dist.is_initialized() or dist.get_rank() == 0:
            print("approximate gkm kernel, %d sequences, %d chains, %d GPUs: %.2f s, chain 0 ran %d iterations (last stdev %.4g)"
                  % (args.n_seq, args.approx, dist.get_world_size() if dist.is_initialized() else 1, dt, len(sd), sd[-1]))
            print(eng.get_block(0, 4, 0, 4))
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    t0 = time.time()
    eng, K = distributed.compute_sharded(tokens, offsets, args.n_seq, 0, args.g, args.m, shard_by=args.shard,
                                         replicate=False)
    dt = time.time() - t0
    corner = distributed.get_block_distributed(eng, 0, 4, 0, 4, device="cuda")  # collective: every rank calls it
    if not dist.is_initialized() or dist.get_rank() == 0:
        print("exact gkm kernel, %d sequences, %d GPUs: %.2f s" % (args.n_seq, dist.get_world_size() if dist.is_initialized() else 1, dt))
        print(corner.cpu().numpy())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

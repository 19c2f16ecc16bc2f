#!/usr/bin/env python3
"""All GPUs of the node from ONE process, through the drop-in class: the reference parallelises one
compute_kernel call over t host threads (fastsk_kernel.cpp:54-93); FastSK(devices=[...]) parallelises it over the
listed GPUs — combos dealt round-robin, one banded RCCL all-reduce of the partial triangles, called from the
engine's host C++ (no torch, no launcher).

    python examples/multi_gpu_inproc.py --gpus 8 --n-seq 100000 --seq-len 300 -g 12 -m 8
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastsk import FastSK  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--n-seq", type=int, default=20000)
    ap.add_argument("--seq-len", type=int, default=300)
    ap.add_argument("-g", type=int, default=12)
    ap.add_argument("-m", type=int, default=8)
    ap.add_argument("--collective", choices=["auto", "rccl", "p2p"], default="auto")
    args = ap.parse_args()
    rng = np.random.Generator(np.random.PCG64(20201214))
    X = rng.integers(1, 5, size=(args.n_seq, args.seq_len), dtype=np.int32)
    n_train = args.n_seq * 9 // 10
    f = FastSK(g=args.g, m=args.m, devices=list(range(args.gpus)), collective=args.collective)
    t0 = time.time()
    f.compute_kernel(X[:n_train], X[n_train:])   # 2-D int32 arrays: no per-element boxing
    dt = time.time() - t0
    st = f.stats()
    print("exact gkm kernel, %d + %d sequences, %d combos on GPUs %s (%s, %d ranks, %d bands, int32 exchange: %s): %.2f s"
          % (n_train, args.n_seq - n_train, st["combos_done"], st["devices"], st["collective"], st["comm_ranks"],
             st["exchange_bands"], st["exchange_int32"], dt))
    print(f.get_block(0, 4, 0, 4))
    # the SVM stage can take the blocks without a host bounce:
    #   import torch; Ktr = torch.from_dlpack(f.get_train_kernel_dlpack())


if __name__ == "__main__":
    main()

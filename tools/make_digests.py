#!/usr/bin/env python3
"""Single-GPU digests of the integer triangle (fsk_counts_digest) for the BASELINE workloads: config 5 at
full size and at the sizes of the N series, config 4. Written to profiles/k_digests.json; bench.py compares
every multi-GPU result with them (`bit_identical_to_1gpu`). Run on one GPU: python tools/make_digests.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fastsk_amd import _native  # noqa: E402


def main():
    out = {}
    if os.path.exists(bench.DIGESTS):
        out = json.load(open(bench.DIGESTS))
    sizes = [int(x) for x in sys.argv[1:]] or [4000, 8000, 16000, 32000, 64000, 100000]
    for N in sizes:
        tokens, offsets, _ = bench.synthetic(N, 300)
        e = _native.Engine(12, 8)
        e.load_sequences(tokens, offsets, N, 0)
        e.accumulate(np.arange(495, dtype=np.int32))
        e.finalize()
        key = bench.digest_key(5, N, 300, 12, 8, 495)
        out[key] = dict(bench.digest_hex(e.counts_digest()), n_gpus=1)
        print(key, out[key], flush=True)
        e.close()
    z = np.load(os.path.join(ROOT, "tests", "golden", "tokens_2.19.npz"))
    tokens, offsets = z["tokens"].astype(np.int32), z["offsets"].astype(np.int64)
    N = len(offsets) - 1
    e = _native.Engine(14, 10)
    e.load_sequences(tokens, offsets, N, 0)
    e.accumulate(np.arange(1001, dtype=np.int32))
    e.finalize()
    key = bench.digest_key(4, N, None, 14, 10, 1001)
    out[key] = dict(bench.digest_hex(e.counts_digest()), n_gpus=1)
    print(key, out[key], flush=True)
    e.close()
    json.dump(out, open(bench.DIGESTS, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

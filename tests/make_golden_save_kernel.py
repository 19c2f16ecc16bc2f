#!/usr/bin/env python3
"""Generate tests/golden/save_kernel.npz (build container only): the bytes FastSK::save_kernel
(fastsk.cpp:223-237) writes — the reference's own code, run through oracle/_ref (ref_save_kernel:
compute_kernel / compute_train, then save_kernel) — for inputs that are already committed fixtures.
Stored per case: the file's bytes (uint8) and their sha256. Only data travels."""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import loader  # noqa: E402
from conftest import load_golden, GOLD  # noqa: E402

# exact, ragged, train-only, and a variance-mode run (fp64 K_hat, seeded order) of committed fixtures
CASES = ["f3_ragged_sigma7_g6m3", "f4_ep300_exact", "f3_train_only", "f4_ep300_variance_T1", "f6_prot219_exact"]


def main():
    ref = loader.ref()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in CASES:
            d = load_golden(name)
            path = os.path.join(tmp, name + ".txt")
            ref.save_kernel(path, d["tokens"], d["offsets"], d["n_train"], d["n_test"], d["g"], d["m"], t=d["t"],
                            approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"],
                            skip_variance=bool(d["skip_variance"]), seed=d.get("seed", 0))
            data = open(path, "rb").read()
            N = d["n_train"] + d["n_test"]
            assert data.count(b"\n") == N
            out[name] = np.frombuffer(data, dtype=np.uint8)
            out[name + "__sha256"] = np.array(hashlib.sha256(data).hexdigest())
            print("%-28s N=%4d  %8d bytes  %s" % (name, N, len(data), hashlib.sha256(data).hexdigest()[:16]))
    np.savez_compressed(os.path.join(GOLD, "save_kernel.npz"), **out)


if __name__ == "__main__":
    main()

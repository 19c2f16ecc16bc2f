import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def set_tuning_env(monkeypatch, **keys):
    """FSK_TUNING for the engines created from here on (fsk_create parses it once; the pybind11 class and the ctypes view
    both go through it): merges `keys` into what is already set; a value of None removes a key."""
    cur = dict(kv.split("=") for kv in os.environ.get("FSK_TUNING", "").split(",") if kv)
    for k, v in keys.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv("FSK_TUNING", ",".join("%s=%s" % kv for kv in cur.items()))
    else:
        monkeypatch.delenv("FSK_TUNING", raising=False)


def golden_names(prefixes=("f1", "f2", "f3", "f4", "f5", "f6", "f8")):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, "*.npz"))):
        n = os.path.basename(p)[:-4]
        if n.split("_")[0] in prefixes:
            out.append(n)
    return out


def load_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    d = {k: z[k] for k in z.files}
    for k in ("n_train", "n_test", "g", "m", "t", "approx", "max_iters", "skip_variance", "seed"):
        if k in d:
            d[k] = int(d[k])
    if "delta" in d:
        d["delta"] = float(d["delta"])
    for k in ("tri_sha256", "counts_sha256", "data"):
        if k in d:
            d[k] = str(d[k])
    return d


def load_tokens(data):
    z = np.load(os.path.join(GOLD, "tokens_%s.npz" % data))
    return (z["tokens"].astype(np.int32), z["offsets"].astype(np.int64), int(z["n_train"]),
            int(z["n_test"]), z["y_train"], z["y_test"])


def tri_to_square(tri, N):
    full = np.zeros((N, N), dtype=tri.dtype)
    il = np.tril_indices(N)
    full[il] = tri
    full.T[il] = tri
    return full


def synthetic_dna(N, L, seed=20201214):
    """BASELINE config 5 generator: tokens 1..4, i.i.d. uniform."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.integers(1, 5, size=(N, L), dtype=np.int32)
    offsets = np.arange(N + 1, dtype=np.int64) * L
    return X.reshape(-1).copy(), offsets


@pytest.fixture(scope="session")
def port():
    from oracle import loader
    return loader.port()


@pytest.fixture(scope="session")
def ref():
    from oracle import loader
    if not loader.have_ref():
        pytest.skip("compiled reference (oracle/_ref) not present")
    return loader.ref()


# degenerate shapes the reference accepts: (sequences, n_train, n_test, g, m)
EDGE_CASES = [
    ([[1, 2, 3, 4, 1, 2]], 1, 0, 6, 2),                                  # N = 1, one window
    ([[1, 2, 3, 4, 1, 2], [2, 2, 3, 4, 1, 2], [1, 2, 3, 4, 1, 1]], 2, 1, 6, 2),   # every length == g
    ([[3, 1, 4, 1, 5, 2, 2, 6]] * 5, 3, 2, 5, 2),                        # identical sequences -> all ones
    ([[1, 2, 1, 2, 1, 2, 1], [2, 1, 2, 1, 2, 1, 2, 1, 2]], 1, 1, 4, 3),  # k = 1
    ([[1, 1, 1, 1, 1], [2, 2, 2, 2, 2, 2]], 1, 1, 3, 1),                 # nothing shared: zero off-diagonal
    ([list(range(1, 30)), list(range(5, 40)), list(range(1, 12))], 2, 1, 11, 0),  # m = 0: one combo, plain 11-mers
]

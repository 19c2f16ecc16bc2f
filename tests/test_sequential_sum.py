"""fsk_sequential_sum — the device-side replacement of the sequential fp64 sum inside the reference's
get_variance (fastsk_kernel.cpp:116-131) — must equal the plain left-to-right sum TO THE LAST BIT on
any input: it is computed from per-block integer totals where that is provably the same, and by
falling back to plain additions where it is not (binade crossings, ties, negative values).
CPU: the kernels run under the emulator; the same cases run on the GPU in tests/test_gpu_parity.py."""
import numpy as np
import pytest


def sequential(values):
    s = np.float64(0.0)
    for v in np.asarray(values, dtype=np.float64):
        s = s + v
    return float(s)


def cases():
    rng = np.random.default_rng(12)
    out = {}
    out["empty"] = np.zeros(0)
    out["one"] = np.array([3.25])
    out["zeros_then_values"] = np.concatenate([np.zeros(100), rng.random(300)])
    out["uniform_20k"] = rng.random(20000) * 7.0                       # several blocks, crossings early on
    out["welford_like"] = (rng.integers(0, 40, 30000) ** 2) * (1.0 - 1.0 / 7.0)   # delta^2 (1 - 1/iter): what variance mode sums
    out["halves"] = rng.integers(0, 9, 25000) * 0.5                    # exact multiples: no rounding anywhere
    out["geometric"] = 1.5 ** np.arange(0, 900, dtype=np.float64)      # a binade crossing at almost every step
    out["tiny_after_big"] = np.concatenate([[2.0 ** 60], rng.random(9000)])        # everything absorbed or half-absorbed
    out["ties_to_even"] = np.concatenate([[2.0 ** 53], np.ones(5000)])             # every addition is a tie
    out["ties_mixed"] = np.concatenate([[2.0 ** 52], rng.integers(0, 4, 9000) * 0.5])
    out["block_edges"] = rng.random(8192 * 2 + 1) * 1e-3
    out["with_negatives"] = rng.standard_normal(17000) * 100.0         # not monotone: everything via the fallback
    out["subnormals"] = np.concatenate([np.full(200, 5e-324), rng.random(100) * 1e-310, rng.random(50)])
    out["huge_values"] = np.concatenate([rng.random(100), [1e300, 1e300], rng.random(100)])
    out["one_big_in_block"] = np.concatenate([rng.random(8000) + 1e6, [2.0 ** 40], rng.random(9000) + 1e6])
    return out


@pytest.fixture(scope="module")
def emu_engine():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import build_emu
    from fastsk_amd import _native
    lib = _native.Library(path=build_emu.build())
    e = _native.Engine(4, 2, lib=lib)
    yield e
    e.close()


@pytest.mark.parametrize("name", sorted(cases()))
def test_sequential_sum_is_the_left_to_right_sum(emu_engine, name):
    x = cases()[name]
    got = emu_engine.sequential_sum(x)
    want = sequential(x)
    assert np.float64(got).tobytes() == np.float64(want).tobytes(), (name, got, want)


def test_sequential_sum_random_vectors(emu_engine):
    """Random vectors spanning several blocks (group records, tails, ties in dyadic data, mostly-zero
    data) — the same five kinds tests/test_gpu_parity.py runs at larger sizes on the GPU."""
    rng = np.random.default_rng(19)
    for t in range(25):
        n = int(rng.integers(1, 30000))
        kind = t % 5
        if kind == 0:
            x = rng.random(n) * 10.0 ** rng.integers(-8, 8)
        elif kind == 1:
            x = (rng.integers(0, 40, n) ** 2) * (1.0 - 1.0 / rng.integers(2, 60))
        elif kind == 2:
            x = np.ldexp(rng.integers(1, 1 << 20, n).astype(np.float64), rng.integers(-30, 30, n))
        elif kind == 3:
            x = np.where(rng.random(n) < 0.7, 0.0, rng.random(n))
        else:
            x = np.full(n, 0.5 * 2.0 ** rng.integers(-5, 5))
            x[rng.integers(0, n)] = 2.0 ** 40
        assert np.float64(emu_engine.sequential_sum(x)).tobytes() == np.float64(sequential(x)).tobytes(), (t, kind, n)

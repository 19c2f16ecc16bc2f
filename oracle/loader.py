"""ctypes loaders for the two CPU checkers (test infrastructure only)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libfastsk_ref.so")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference exists, _ref/libfastsk_ref.so."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def have_ref():
    return os.path.exists(_REF)


def flatten(X):
    """list of int sequences -> (tokens int32[sum len], offsets int64[n+1])."""
    lens = np.fromiter((len(x) for x in X), dtype=np.int64, count=len(X))
    offsets = np.zeros(len(X) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    tokens = np.empty(int(offsets[-1]), dtype=np.int32)
    for i, x in enumerate(X):
        tokens[offsets[i]:offsets[i + 1]] = x
    return tokens, offsets


class _Ref:
    """The compiled reference (QData/FastSK) behind oracle/ref_harness.cpp."""

    def __init__(self):
        if not have_ref():
            raise RuntimeError("oracle/_ref/libfastsk_ref.so missing (run `make -C oracle`)")
        L = C.CDLL(_REF)
        L.ref_shuffle_order.argtypes = [C.c_long, C.c_int, _i32p]
        L.ref_shuffle_order.restype = None
        common = [_i32p, _i64p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                  C.c_int, C.c_int, C.c_long]
        L.ref_compute.argtypes = common + [_f64p, _f64p, _f64p, C.c_int, C.c_int]
        L.ref_compute.restype = C.c_int
        L.ref_full_triangle.argtypes = common + [_f64p, _f64p, C.c_int, C.c_int]
        L.ref_full_triangle.restype = C.c_int
        if hasattr(L, "ref_save_kernel"):  # (a prebuilt library of an earlier round lacks it)
            L.ref_save_kernel.argtypes = common + [C.c_char_p, C.c_int]
            L.ref_save_kernel.restype = C.c_int
        L.ref_raw_counts.argtypes = [_i32p, _i64p, C.c_int64, C.c_int, C.c_int, _i32p, C.c_int,
                                     C.c_int, C.c_void_p]
        L.ref_raw_counts.restype = C.c_double
        self.L = L

    def shuffle_order(self, seed, n):
        out = np.empty(n, dtype=np.int32)
        self.L.ref_shuffle_order(seed, n, out)
        return out

    def compute(self, tokens, offsets, n_train, n_test, g, m, t=-1, approx=False, delta=0.025,
                max_iters=-1, skip_variance=False, seed=0, quiet=True):
        train = np.zeros((n_train, n_train), dtype=np.float64)
        test = np.zeros((max(n_test, 1), n_train), dtype=np.float64)
        sd = np.zeros(4096, dtype=np.float64)
        n = self.L.ref_compute(tokens, offsets, n_train, n_test, g, m, t, int(approx), delta,
                               max_iters, int(skip_variance), seed, train, test, sd, sd.size,
                               int(quiet))
        return train, test[:n_test], sd[:n].copy()

    def full_triangle(self, tokens, offsets, n_train, n_test, g, m, t=-1, approx=False,
                      delta=0.025, max_iters=-1, skip_variance=False, seed=0, quiet=True):
        N = n_train + n_test
        tri = np.zeros(N * (N + 1) // 2, dtype=np.float64)
        sd = np.zeros(4096, dtype=np.float64)
        n = self.L.ref_full_triangle(tokens, offsets, n_train, n_test, g, m, t, int(approx), delta,
                                     max_iters, int(skip_variance), seed, tri, sd, sd.size,
                                     int(quiet))
        return tri, sd[:n].copy()

    def save_kernel(self, path, tokens, offsets, n_train, n_test, g, m, t=-1, approx=False, delta=0.025,
                    max_iters=-1, skip_variance=False, seed=0, quiet=True):
        """compute_kernel / compute_train + FastSK::save_kernel(path) (fastsk.cpp:223-237)."""
        self.L.ref_save_kernel(tokens, offsets, n_train, n_test, g, m, t, int(approx), delta, max_iters,
                               int(skip_variance), seed, str(path).encode(), int(quiet))

    def raw_counts(self, tokens, offsets, g, m, combos, threads=1, want_counts=True):
        N = len(offsets) - 1
        combos = np.ascontiguousarray(combos, dtype=np.int32)
        out = np.zeros(N * (N + 1) // 2, dtype=np.uint64) if want_counts else None
        secs = self.L.ref_raw_counts(tokens, offsets, N, g, m, combos, len(combos), threads,
                                     out.ctypes.data if want_counts else None)
        return out, secs


_ref_singleton = None


def ref():
    global _ref_singleton
    if _ref_singleton is None:
        _ref_singleton = _Ref()
    return _ref_singleton


class _Port:
    """liboracle.so — our plain-C restatement (fastsk_oracle.c)."""

    def __init__(self):
        if not os.path.exists(_PORT):
            build()
        L = C.CDLL(_PORT)
        L.orc_num_combos.argtypes = [C.c_int, C.c_int]
        L.orc_num_combos.restype = C.c_int64
        L.orc_combo_positions.argtypes = [C.c_int, C.c_int, C.c_int64, _i32p]
        L.orc_combo_positions.restype = C.c_int
        L.orc_raw_counts.argtypes = [_i32p, _i64p, C.c_int64, C.c_int, C.c_int, _i32p, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p]
        L.orc_raw_counts.restype = C.c_double
        L.orc_normalise.argtypes = [_f64p, C.c_int64]
        L.orc_normalise.restype = None
        L.orc_compute.argtypes = [_i32p, _i64p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_double, C.c_int, C.c_int, _i32p, C.c_int,
                                  _f64p, _f64p, C.c_int, C.c_void_p]
        L.orc_compute.restype = C.c_int
        self.L = L

    def num_combos(self, g, m):
        return int(self.L.orc_num_combos(g, m))

    def combo_positions(self, g, k, combo):
        out = np.zeros(k, dtype=np.int32)
        rc = self.L.orc_combo_positions(g, k, combo, out)
        if rc != 0:
            raise ValueError("bad combo id")
        return out

    def raw_counts(self, tokens, offsets, g, m, combos, threads=1, want_counts=True):
        """Sum of per-combo partial kernels as uint64 triangle; returns (counts, seconds, U)."""
        N = len(offsets) - 1
        combos = np.ascontiguousarray(combos, dtype=np.int32)
        out = np.zeros(N * (N + 1) // 2, dtype=np.uint64) if want_counts else None
        U = C.c_uint64(0)
        secs = self.L.orc_raw_counts(tokens, offsets, N, g, m, combos, len(combos), threads,
                                     out.ctypes.data if want_counts else None, C.byref(U))
        return out, secs, int(U.value)

    def normalise(self, tri, N):
        tri = np.ascontiguousarray(tri, dtype=np.float64).copy()
        self.L.orc_normalise(tri, N)
        return tri

    def compute(self, tokens, offsets, n_train, n_test, g, m, t=-1, approx=False, delta=0.025,
                max_iters=-1, skip_variance=False, order=None):
        """Full path incl. approx modes, for an explicit combo order. Returns (tri, stdevs, iters)."""
        N = n_train + n_test
        nc = self.num_combos(g, m)
        if order is None:
            order = np.arange(nc, dtype=np.int32)
        order = np.ascontiguousarray(order, dtype=np.int32)
        tri = np.zeros(N * (N + 1) // 2, dtype=np.float64)
        sd = np.zeros(max(nc, 1), dtype=np.float64)
        iters = np.zeros(max(1, min(nc, 20 if t == -1 else max(t, 1))), dtype=np.int32)
        n = self.L.orc_compute(tokens, offsets, n_train, n_test, g, m, t, int(approx), delta,
                               max_iters, int(skip_variance), order, len(order), tri, sd, sd.size,
                               iters.ctypes.data)
        if n < 0:
            raise ValueError("oracle rejected the arguments (rc=%d)" % n)
        return tri, sd[:n].copy(), iters


_port_singleton = None


def port():
    global _port_singleton
    if _port_singleton is None:
        _port_singleton = _Port()
    return _port_singleton

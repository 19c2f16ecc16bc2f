"""ctypes view of the C ABI in ``include/fastsk_amd.h`` (``libfastsk_amd.so``).

Used by the staged / multi-GPU host code (``fastsk_amd.distributed``), ``bench.py`` and the parity
tests; the drop-in ``FastSK`` class itself is the pybind11 module ``fastsk_amd._fastsk`` built on
the same library. There is no CPU fallback: if the HIP library is missing this module raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfastsk_amd.so")

PATH_AUTO, PATH_DENSE, PATH_SPARSE = 0, 1, 2
COLL_AUTO, COLL_RCCL, COLL_P2P = 0, 1, 2
ABI_VERSION = 5

ERRORS = {-1: "FSK_EINVAL", -2: "FSK_ESHORT", -3: "FSK_ESTATE", -4: "FSK_EDEVICE", -5: "FSK_ENOMEM",
          -6: "FSK_EUNSUPPORTED"}


class FskError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "FSK_E?"), code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("g", C.c_int32), ("m", C.c_int32), ("t", C.c_int32), ("approx", C.c_int32),
                ("delta", C.c_double), ("max_iters", C.c_int32), ("skip_variance", C.c_int32),
                ("device", C.c_int32), ("path", C.c_int32), ("profile", C.c_int32),
                ("skip_test_block", C.c_int32), ("collective", C.c_int32), ("bands", C.c_int32),
                ("deadline_ms", C.c_int32), ("reserved", C.c_int32 * 1)]


class Stats(C.Structure):
    _fields_ = [("n_seq", C.c_int64), ("n_train", C.c_int64), ("n_test", C.c_int64), ("n_feat", C.c_int64),
                ("n_pairs", C.c_int64), ("alphabet", C.c_int32), ("bits_per_symbol", C.c_int32),
                ("key_space", C.c_int64), ("path_used", C.c_int32), ("n_combos_total", C.c_int32),
                ("combos_done", C.c_int64), ("cell_updates", C.c_uint64), ("sort_records", C.c_uint64),
                ("sort_passes", C.c_int32), ("launches", C.c_int32), ("ms_count", C.c_double),
                ("ms_tile", C.c_double), ("ms_extract", C.c_double), ("ms_sort", C.c_double),
                ("ms_segment", C.c_double), ("ms_pairs", C.c_double), ("ms_total", C.c_double),
                ("n_tile_launches", C.c_int64), ("dense_macs", C.c_uint64), ("panel_bytes", C.c_uint64),
                ("u4_tile_launches", C.c_double), ("max_windows", C.c_double), ("count_launches", C.c_double),
                ("compact_keys_avg", C.c_double), ("batches_redone", C.c_double), ("combos_issued", C.c_double),
                ("sparse_form", C.c_double), ("sparse_passes", C.c_double), ("share_positions", C.c_double), ("share_groups", C.c_double),
                ("sparse_desc", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class MultiInfo(C.Structure):
    _fields_ = [("ndev", C.c_int32), ("devices", C.c_int32 * 16), ("collective", C.c_int32), ("comm_ranks", C.c_int32),
                ("bands", C.c_int32), ("narrow", C.c_int32), ("reduce_bytes", C.c_int64),
                ("combos_per_engine", C.c_int64 * 16), ("reserved", C.c_double * 4)]

    def as_dict(self):
        n = self.ndev
        return {"ndev": n, "devices": list(self.devices[:n]),
                "collective": {COLL_RCCL: "rccl", COLL_P2P: "p2p"}.get(self.collective, "none"),
                "comm_ranks": self.comm_ranks, "bands": self.bands, "narrow": bool(self.narrow),
                "reduce_bytes": self.reduce_bytes, "combos_per_engine": list(self.combos_per_engine[:n])}


# every symbol include/fastsk_amd.h declares (checked by tests/test_abi.py)
SYMBOLS = ["fsk_create", "fsk_destroy", "fsk_last_error", "fsk_abi_version", "fsk_device_count", "fsk_compute",
           "fsk_set_combo_order", "fsk_set_seed", "fsk_load_sequences", "fsk_bind_counts",
           "fsk_counts_device_ptr", "fsk_reset_counts", "fsk_reset_counts_rows", "fsk_accumulate", "fsk_accumulate_rows", "fsk_synchronize", "fsk_finalize",
           "fsk_get_block", "fsk_get_block_device", "fsk_get_train", "fsk_get_test", "fsk_get_triangle", "fsk_get_counts",
           "fsk_get_counts_block", "fsk_get_counts_cells", "fsk_get_stdevs", "fsk_save_kernel", "fsk_get_stats", "fsk_num_combos",
           "fsk_combo_positions", "fsk_stream_wait_engine", "fsk_engine_wait_stream", "fsk_read_fasta", "fsk_sequential_sum",
           "fsk_run_chains", "fsk_get_kernel_sum_device", "fsk_set_kernel_sum_device", "fsk_create_multi", "fsk_get_multi_info",
           "fsk_counts_digest", "fsk_alloc_block_device", "fsk_free_device", "fsk_set_skip_test_block",
           "fsk_get_triangle_device", "fsk_alloc_triangle_device", "fsk_set_tuning", "fsk_get_tuning", "fsk_tuning_keys", "fsk_seed_order"]


_hip_shared = False


def _mapped_runtimes():
    """Paths of the libamdhip64 / libhsa-runtime64 images mapped into this process."""
    found = {}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                base = os.path.basename(path)
                for stem in ("libamdhip64.so", "libhsa-runtime64.so"):
                    if base.startswith(stem):
                        found.setdefault(stem, set()).add(path)
    except OSError:
        pass
    return found


def share_hip_runtime_with_torch():
    """One HIP runtime per process. A PyTorch-ROCm wheel carries its own libamdhip64 /
    libhsa-runtime64; if this engine binds /opt/rocm's copy first and torch is imported later, the
    process ends up with two HSA runtimes and the second one to initialise finds no GPU. So when
    torch is installed but not imported yet, and no HIP runtime is mapped yet (a profiler's preload, an
    earlier import), its copies (same SONAMEs) are loaded first and the engine binds to them —
    exactly what happens anyway when torch is imported before this package.
    No torch installed: nothing to do, /opt/rocm's runtime is used."""
    global _hip_shared
    if _hip_shared or "torch" in sys.modules:
        _hip_shared = True
        return
    _hip_shared = True
    if _mapped_runtimes():
        return  # a runtime is already in the process: loading torch's copy by path would add a second image
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def check_single_hip_runtime():
    """Warn when two different libamdhip64 images ended up in the process (e.g. a torch wheel built
    against another ROCm major: different SONAME, so the preload above cannot unify them)."""
    images = _mapped_runtimes().get("libamdhip64.so", set())
    real = {os.path.realpath(p) for p in images}
    if len(real) > 1:
        import warnings
        warnings.warn("two HIP runtimes are mapped into this process (%s): the engine and torch will not see the same "
                      "GPU state; import torch before fastsk_amd, or use a torch wheel built for this ROCm" % ", ".join(sorted(real)))


class Library:
    """The loaded shared library with typed entry points."""

    def __init__(self, path=None):
        if path is None:
            share_hip_runtime_with_torch()  # the product library; an explicit path is the CPU emulation of the tests
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                "%s not found: build the HIP engine first (python -c 'import __graft_entry__ as g; g.build()'). "
                "fastsk_amd has no CPU fallback." % path)
        L = C.CDLL(path)
        if path == LIB_PATH:
            check_single_hip_runtime()
        self.path = path
        L.fsk_abi_version.restype = C.c_int
        if L.fsk_abi_version() != ABI_VERSION:  # a stale build: its structs and symbols are not the ones bound below
            raise ImportError("%s speaks ABI version %d, this package needs %d: rebuild it (python -c 'import __graft_entry__ "
                              "as g; g.build()')" % (path, L.fsk_abi_version(), ABI_VERSION))
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        sig = {
            "fsk_create": ([C.POINTER(Config), C.POINTER(vp)], C.c_int),
            "fsk_destroy": ([vp], None),
            "fsk_last_error": ([vp], C.c_char_p),
            "fsk_abi_version": ([], C.c_int),
            "fsk_device_count": ([], C.c_int),
            "fsk_compute": ([vp, vp, vp, i64, i64], C.c_int),
            "fsk_set_combo_order": ([vp, vp, i32], C.c_int),
            "fsk_set_seed": ([vp, C.c_uint64], C.c_int),
            "fsk_load_sequences": ([vp, vp, vp, i64, i64], C.c_int),
            "fsk_bind_counts": ([vp, vp, i64], C.c_int),
            "fsk_counts_device_ptr": ([vp, C.POINTER(vp)], C.c_int),
            "fsk_reset_counts": ([vp], C.c_int),
            "fsk_reset_counts_rows": ([vp, i64, i64], C.c_int),
            "fsk_accumulate": ([vp, vp, i32], C.c_int),
            "fsk_accumulate_rows": ([vp, vp, i32, i64, i64], C.c_int),
            "fsk_synchronize": ([vp], C.c_int),
            "fsk_finalize": ([vp], C.c_int),
            "fsk_get_block": ([vp, i64, i64, i64, i64, vp], C.c_int),
            "fsk_get_block_device": ([vp, i64, i64, i64, i64, vp], C.c_int),
            "fsk_get_train": ([vp, vp], C.c_int),
            "fsk_get_test": ([vp, vp], C.c_int),
            "fsk_get_triangle": ([vp, vp], C.c_int),
            "fsk_get_counts": ([vp, vp], C.c_int),
            "fsk_get_counts_block": ([vp, i64, i64, i64, i64, vp], C.c_int),
            "fsk_get_counts_cells": ([vp, vp, vp, i64, vp], C.c_int),
            "fsk_stream_wait_engine": ([vp, vp], C.c_int),
            "fsk_engine_wait_stream": ([vp, vp], C.c_int),
            "fsk_read_fasta": ([C.c_char_p, vp, C.POINTER(i32), vp, i64, vp, vp, i64, C.POINTER(i64), C.POINTER(i64), C.c_char_p, i32], C.c_int),
            "fsk_sequential_sum": ([vp, vp, i64, C.POINTER(C.c_double)], C.c_int),
            "fsk_run_chains": ([vp, i32, i32], C.c_int),
            "fsk_get_kernel_sum_device": ([vp, vp], C.c_int),
            "fsk_set_kernel_sum_device": ([vp, vp], C.c_int),
            "fsk_get_stdevs": ([vp, vp, i32, C.POINTER(i32)], C.c_int),
            "fsk_save_kernel": ([vp, C.c_char_p], C.c_int),
            "fsk_get_stats": ([vp, C.POINTER(Stats)], C.c_int),
            "fsk_num_combos": ([i32, i32], i64),
            "fsk_combo_positions": ([i32, i32, i64, vp], C.c_int),
            "fsk_create_multi": ([C.POINTER(Config), vp, i32, C.POINTER(vp)], C.c_int),
            "fsk_get_multi_info": ([vp, C.POINTER(MultiInfo)], C.c_int),
            "fsk_counts_digest": ([vp, i64, i64, vp], C.c_int),
            "fsk_alloc_block_device": ([vp, i64, i64, i64, i64, C.POINTER(vp)], C.c_int),
            "fsk_free_device": ([vp, vp], C.c_int),
            "fsk_set_skip_test_block": ([vp, i32], C.c_int),
            "fsk_get_triangle_device": ([vp, vp], C.c_int),
            "fsk_alloc_triangle_device": ([vp, C.POINTER(vp)], C.c_int),
            "fsk_set_tuning": ([vp, C.c_char_p, i64], C.c_int),
            "fsk_get_tuning": ([vp, C.c_char_p, C.POINTER(i64)], C.c_int),
            "fsk_tuning_keys": ([], C.c_char_p),
            "fsk_seed_order": ([C.c_uint64, i64, vp], C.c_int),
        }
        for name, (argtypes, restype) in sig.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = restype
        self.L = L

    def num_combos(self, g, m):
        return int(self.L.fsk_num_combos(g, m))

    def combo_positions(self, g, k, combo):
        out = np.zeros(k, dtype=np.int32)
        rc = self.L.fsk_combo_positions(g, k, combo, out.ctypes.data)
        if rc:
            raise FskError(rc, "bad combination id")
        return out

    def seed_order(self, seed, n):
        """The combo order ``fsk_set_seed(seed)`` stands for: the reference's std::shuffle with time(0) == seed."""
        out = np.zeros(max(n, 1), dtype=np.int32)
        rc = self.L.fsk_seed_order(seed, n, out.ctypes.data)
        if rc:
            raise FskError(rc, "bad length")
        return out[:n]

    def device_count(self):
        return int(self.L.fsk_device_count())

    def tuning_keys(self):
        """{key: (default, lowest, highest, what it does)} — every knob fsk_set_tuning / FSK_TUNING takes."""
        out = {}
        for line in self.L.fsk_tuning_keys().decode().splitlines():
            head, rng, doc = line.split(" ", 2)
            key, default = head.split("=")
            lo, hi = rng.strip("[]").split("..")
            out[key] = (int(default), int(lo), int(hi), doc)
        return out


_default = None


def library():
    """The product library (hipcc build). Raises ImportError when it has not been built."""
    global _default
    if _default is None:
        _default = Library()
    return _default


def flatten(X):
    """Nested int sequences (or a 2-D array) -> (tokens int32, offsets int64)."""
    if isinstance(X, np.ndarray) and X.ndim == 2:
        n, L = X.shape
        return np.ascontiguousarray(X, dtype=np.int32).reshape(-1), np.arange(n + 1, dtype=np.int64) * L
    lens = np.fromiter((len(x) for x in X), dtype=np.int64, count=len(X))
    offsets = np.zeros(len(X) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    tokens = np.empty(int(offsets[-1]), dtype=np.int32)
    for i, x in enumerate(X):
        tokens[offsets[i]:offsets[i + 1]] = x
    return tokens, offsets


class Engine:
    """One engine handle = one device, one HIP stream — or, with ``devices=[...]``, one engine per listed GPU
    of this process behind the same handle (``fsk_create_multi``). Mirrors the C ABI one to one."""

    def __init__(self, g, m, t=-1, approx=False, delta=0.025, max_iters=-1, skip_variance=False, device=0,
                 path=PATH_AUTO, profile=False, lib=None, skip_test_block=False, devices=None, collective=COLL_AUTO,
                 bands=0, deadline_ms=0, tuning=None):
        self.lib = lib or library()
        if devices is not None:
            devices = [int(d) for d in devices]
            if not devices:
                raise ValueError("devices must list at least one GPU")
            device = devices[0]
        cfg = Config(g=g, m=m, t=t, approx=int(bool(approx)), delta=delta, max_iters=max_iters,
                     skip_variance=int(bool(skip_variance)), device=device, path=path, profile=int(profile),
                     skip_test_block=int(bool(skip_test_block)), collective=int(collective), bands=int(bands),
                     deadline_ms=int(deadline_ms))
        h = C.c_void_p()
        if devices is None:
            rc = self.lib.L.fsk_create(C.byref(cfg), C.byref(h))
        else:
            arr = (C.c_int32 * len(devices))(*devices)
            rc = self.lib.L.fsk_create_multi(C.byref(cfg), arr, len(devices), C.byref(h))
        if rc:
            raise FskError(rc, (self.lib.L.fsk_last_error(None) or b"").decode())
        self.devices = devices
        self.h = h
        self.g, self.m = g, m
        self.device = device
        self._keep = None
        for key, value in (tuning or {}).items():
            self.set_tuning(key, value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.L.fsk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise FskError(rc, (self.lib.L.fsk_last_error(self.h) or b"").decode())

    # ---- inputs
    @staticmethod
    def _prep(tokens, offsets):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        return tokens, offsets

    def compute(self, tokens, offsets, n_train, n_test):
        tokens, offsets = self._prep(tokens, offsets)
        self.N, self.n_train, self.n_test = n_train + n_test, n_train, n_test
        self._ck(self.lib.L.fsk_compute(self.h, tokens.ctypes.data, offsets.ctypes.data, n_train, n_test))

    def load_sequences(self, tokens, offsets, n_train, n_test):
        tokens, offsets = self._prep(tokens, offsets)
        self.N, self.n_train, self.n_test = n_train + n_test, n_train, n_test
        self._ck(self.lib.L.fsk_load_sequences(self.h, tokens.ctypes.data, offsets.ctypes.data, n_train, n_test))

    def set_combo_order(self, order):
        order = np.ascontiguousarray(order, dtype=np.int32)
        self._ck(self.lib.L.fsk_set_combo_order(self.h, order.ctypes.data, len(order)))

    def set_seed(self, seed):
        self._ck(self.lib.L.fsk_set_seed(self.h, seed))

    # ---- staged path
    def bind_counts(self, device_ptr, n_cells, keepalive=None):
        self._keep = keepalive
        self._ck(self.lib.L.fsk_bind_counts(self.h, C.c_void_p(device_ptr), n_cells))

    def counts_device_ptr(self):
        p = C.c_void_p()
        self._ck(self.lib.L.fsk_counts_device_ptr(self.h, C.byref(p)))
        return p.value

    def reset_counts(self):
        self._ck(self.lib.L.fsk_reset_counts(self.h))

    def reset_counts_rows(self, row_begin, row_end):
        self._ck(self.lib.L.fsk_reset_counts_rows(self.h, row_begin, row_end))

    def accumulate(self, combos):
        combos = np.ascontiguousarray(combos, dtype=np.int32)
        self._ck(self.lib.L.fsk_accumulate(self.h, combos.ctypes.data, len(combos)))

    def accumulate_rows(self, combos, row_begin, row_end):
        combos = np.ascontiguousarray(combos, dtype=np.int32)
        self._ck(self.lib.L.fsk_accumulate_rows(self.h, combos.ctypes.data, len(combos), row_begin, row_end))

    def synchronize(self):
        self._ck(self.lib.L.fsk_synchronize(self.h))

    def stream_wait_engine(self, hip_stream):
        """Work enqueued on ``hip_stream`` (a raw hipStream_t, e.g. ``torch.cuda.current_stream().cuda_stream``)
        from now on waits for what the engine has enqueued so far; the host does not block."""
        self._ck(self.lib.L.fsk_stream_wait_engine(self.h, C.c_void_p(hip_stream)))

    def engine_wait_stream(self, hip_stream):
        """The engine's later work waits for what ``hip_stream`` holds now."""
        self._ck(self.lib.L.fsk_engine_wait_stream(self.h, C.c_void_p(hip_stream)))

    def finalize(self):
        self._ck(self.lib.L.fsk_finalize(self.h))

    # ---- results
    @property
    def pairs(self):
        return self.N * (self.N + 1) // 2

    def get_block(self, i0, i1, j0, j1):
        out = np.empty((i1 - i0, j1 - j0), dtype=np.float64)
        self._ck(self.lib.L.fsk_get_block(self.h, i0, i1, j0, j1, out.ctypes.data))
        return out

    def get_block_torch(self, i0, i1, j0, j1):
        """The normalised block as a float64 torch tensor ON THE GPU (no host round trip)."""
        import torch
        dev = torch.device("cuda", self.device)  # the engine's device, whatever torch's current one is
        out = torch.empty((i1 - i0, j1 - j0), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        self._ck(self.lib.L.fsk_get_block_device(self.h, i0, i1, j0, j1, C.c_void_p(out.data_ptr())))
        return out

    def get_train(self):
        out = np.empty((self.n_train, self.n_train), dtype=np.float64)
        self._ck(self.lib.L.fsk_get_train(self.h, out.ctypes.data))
        return out

    def get_test(self):
        out = np.empty((self.n_test, self.n_train), dtype=np.float64)
        self._ck(self.lib.L.fsk_get_test(self.h, out.ctypes.data))
        return out

    def get_triangle(self):
        out = np.empty(self.pairs, dtype=np.float64)
        self._ck(self.lib.L.fsk_get_triangle(self.h, out.ctypes.data))
        return out

    def get_triangle_torch(self, out=None):
        """The whole normalised triangle as a float64 torch tensor ON THE GPU (pairs doubles, the reference's K)."""
        import torch
        dev = torch.device("cuda", self.device)
        if out is None:
            out = torch.empty(self.pairs, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        self._ck(self.lib.L.fsk_get_triangle_device(self.h, C.c_void_p(out.data_ptr())))
        return out

    def get_counts(self):
        out = np.empty(self.pairs, dtype=np.uint64)
        self._ck(self.lib.L.fsk_get_counts(self.h, out.ctypes.data))
        return out

    def counts_digest(self, row_begin=0, row_end=None):
        """(sum, xor of cell * (index | 1)) over the integer cells of rows [row_begin, row_end), mod 2^64: an
        order-free digest computed on the device; digests of disjoint row ranges combine by + and ^."""
        out = (C.c_uint64 * 2)()
        self._ck(self.lib.L.fsk_counts_digest(self.h, row_begin, self.N if row_end is None else row_end, out))
        return int(out[0]), int(out[1])

    def multi_info(self):
        info = MultiInfo()
        self._ck(self.lib.L.fsk_get_multi_info(self.h, C.byref(info)))
        return info.as_dict()

    def set_tuning(self, key, value):
        """One knob of the engine (``Library.tuning_keys()`` lists them); never changes a result."""
        self._ck(self.lib.L.fsk_set_tuning(self.h, key.encode(), int(value)))

    def get_tuning(self, key):
        v = C.c_int64(0)
        self._ck(self.lib.L.fsk_get_tuning(self.h, key.encode(), C.byref(v)))
        return int(v.value)

    def set_skip_test_block(self, skip):
        self._ck(self.lib.L.fsk_set_skip_test_block(self.h, int(bool(skip))))

    def get_counts_block(self, i0, i1, j0, j1):
        out = np.empty((i1 - i0, j1 - j0), dtype=np.uint64)
        self._ck(self.lib.L.fsk_get_counts_block(self.h, i0, i1, j0, j1, out.ctypes.data))
        return out

    def get_counts_cells(self, rows, cols):
        """Raw integer cells (rows[q], cols[q]) of the symmetric matrix."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        assert rows.shape == cols.shape and rows.ndim == 1
        out = np.empty(len(rows), dtype=np.uint64)
        self._ck(self.lib.L.fsk_get_counts_cells(self.h, rows.ctypes.data, cols.ctypes.data, len(rows), out.ctypes.data))
        return out

    def sequential_sum(self, values):
        """Sum of ``values`` in index order (the reduction of the reference's get_variance), on the device."""
        values = np.ascontiguousarray(values, dtype=np.float64)
        out = C.c_double(0.0)
        self._ck(self.lib.L.fsk_sequential_sum(self.h, values.ctypes.data, len(values), C.byref(out)))
        return out.value

    # ---- variance mode over several engines: the Welford chains are the units
    def run_chains(self, first, step):
        """Chains ``first, first + step, ...`` of the ``t`` Welford chains (after ``load_sequences``)."""
        self._ck(self.lib.L.fsk_run_chains(self.h, first, step))

    def get_kernel_sum_device(self, device_ptr):
        """The fp64 sum of the chains' K_hat into ``pairs`` doubles at ``device_ptr`` (device memory)."""
        self._ck(self.lib.L.fsk_get_kernel_sum_device(self.h, C.c_void_p(device_ptr)))

    def set_kernel_sum_device(self, device_ptr):
        self._ck(self.lib.L.fsk_set_kernel_sum_device(self.h, C.c_void_p(device_ptr)))

    def get_stdevs(self):
        n = C.c_int32(0)
        self._ck(self.lib.L.fsk_get_stdevs(self.h, None, 0, C.byref(n)))
        out = np.empty(max(n.value, 1), dtype=np.float64)
        self._ck(self.lib.L.fsk_get_stdevs(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def save_kernel(self, path):
        self._ck(self.lib.L.fsk_save_kernel(self.h, path.encode()))

    def stats(self):
        s = Stats()
        self._ck(self.lib.L.fsk_get_stats(self.h, C.byref(s)))
        return s.as_dict()

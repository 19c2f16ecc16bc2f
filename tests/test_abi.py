"""The C-ABI library loads and exports every symbol include/fastsk_amd.h declares (no compute
calls: there is no GPU here), and the product refuses to run without a device."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "fastsk_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fsk_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def product_lib():
    import __graft_entry__ as ge
    ge.build_engine()
    from fastsk_amd import _native
    return _native.Library()


def test_every_declared_symbol_is_exported(product_lib):
    from fastsk_amd import _native
    names = header_functions()
    assert len(names) >= 25
    assert sorted(_native.SYMBOLS) == names, "ctypes view and header disagree"
    for n in names:
        assert hasattr(product_lib.L, n), n
    assert product_lib.L.fsk_abi_version() == _native.ABI_VERSION == 5


def test_tuning_keys_and_no_ambient_switches(product_lib):
    """One tuning table (fsk_tuning_keys), ONE environment variable (FSK_TUNING, parsed by fsk_create): the engine's
    sources read nothing else from the environment; the test-only keys do not exist in the product library."""
    keys = product_lib.tuning_keys()
    assert {"trace", "compact", "sparse_global", "sparse_unpacked", "guard_cap", "deadline_ms", "collective"} <= set(keys)
    assert keys["deadline_ms"][0] == 120000 and keys["sparse_global"][:3] == (0, 0, 1)
    assert not any(k.startswith("fault_") for k in keys), "test hooks in the product library"
    csrc = os.path.join(ROOT, "fastsk_amd", "csrc")
    reads = []
    for f in sorted(os.listdir(csrc)):
        for n, line in enumerate(open(os.path.join(csrc, f), errors="replace"), 1):
            if re.search(r"\bgetenv\s*\(", line):
                reads.append((f, n, line.strip()))
    assert len(reads) == 1 and "FSK_TUNING" in reads[0][2], reads


def test_product_library_holds_no_test_infrastructure(product_lib):
    """The stand-in for librccl of the CPU test build (tests/emu/rccl_stub.cpp) and the fault hooks (-DFSK_TEST_HOOKS) are
    in tests/ only: the product library neither defines nor needs their symbols, and the package never names tests/."""
    import subprocess
    syms = subprocess.run(["nm", "-D", product_lib.path], capture_output=True, text=True, check=True).stdout
    for name in ("emu_rccl_set_fault", "emu_rccl_stats", "ncclAllReduce", "ncclCommInitAll", "ncclCommAbort"):
        assert not re.search(r"\b%s\b" % name, syms), name      # (RCCL itself is bound with dlopen: not even an undefined reference)
    blob = open(product_lib.path, "rb").read()
    assert b"fault_kind" not in blob and b"rccl_stub" not in blob
    pkg = os.path.join(ROOT, "fastsk_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            text = open(os.path.join(pkg, f)).read()
            assert "tests/emu" not in text and "libfastsk_emu" not in text and "rccl_stub" not in text and "hooks" not in text, f


def test_host_helpers_without_gpu(product_lib):
    import itertools
    assert product_lib.num_combos(12, 8) == 495 and product_lib.num_combos(14, 10) == 1001
    want = list(itertools.combinations(range(10), 4))
    for c in (0, 57, 209):
        assert tuple(product_lib.combo_positions(10, 4, c)) == want[c]


def test_no_cpu_fallback(product_lib):
    """Without a device the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from fastsk_amd import _native
    with pytest.raises(_native.FskError) as ei:
        _native.Engine(10, 6, lib=product_lib)
    assert ei.value.code == -4 and "no CPU fallback" in str(ei.value)


def test_pybind_surface_signature():
    """Same class, keywords and defaults as the reference (bindings.cpp:12-44)."""
    import __graft_entry__ as ge
    ge.build_engine()
    ge.build_bindings()
    from fastsk_amd import _fastsk
    doc = _fastsk.FastSK.__init__.__doc__
    order = [doc.index(" %s: " % kw) for kw in ("g", "m", "t", "approx", "delta", "max_iters", "skip_variance")]
    assert order == sorted(order), "positional order differs from the reference"
    for pat in (r"t: [^,]*= -1", r"approx: bool = False", r"delta: [^,]*= 0.025", r"max_iters: [^,]*= -1",
                r"skip_variance: bool = False"):
        assert re.search(pat, doc), pat
    for meth in ("compute_kernel", "compute_train", "get_train_kernel", "get_test_kernel", "get_stdevs",
                 "save_kernel", "fit", "score"):
        assert hasattr(_fastsk.FastSK, meth)
    assert _fastsk.__version__ == "dev"
    import fastsk  # the drop-in package name
    assert fastsk.FastSK is _fastsk.FastSK and fastsk.FastaUtility is not None


def test_headline_tile_kernel_resources():
    """k_dense_tile_dma — 98.6 % of the headline's GPU time — keeps the register allocation it is priced with (126 VGPRs, no
    scratch, 40 KB of LDS: four workgroups a CU) whatever else the library gains: the small-N variants of the same body live
    in a translation unit of their own (fsk_engine_dense_small.hip). From the compiler's own resource report (cross-compiled)."""
    import re
    import subprocess
    import __graft_entry__ as ge
    src = os.path.join(ge.CSRC, "fsk_engine_dense.hip")
    r = subprocess.run([ge.HIPCC] + ge.HIPCC_FLAGS + ["-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)
    found = {}
    for b in blocks[1:]:
        name = b.split()[0]
        get = lambda key: int(re.search(key + r": (\d+)", b).group(1))
        found[name] = (get("VGPRs"), get(r"ScratchSize \[bytes/lane\]"), get(r"LDS Size \[bytes/block\]"))
    head = [v for k, v in found.items() if "k_dense_tile_dma" in k and "compact" not in k]
    assert head and head[0] == (126, 0, 40960), found
    compact = [v for k, v in found.items() if "k_dense_tile_dma_compact" in k]
    assert compact and compact[0][1] == 0 and compact[0][0] <= 128 and compact[0][2] == 40960, found

"""fastsk_amd — MI355X-native gapped-k-mer string-kernel engine (drop-in for QData/FastSK's
``FastSK(g, m, ...).compute_kernel(Xtrain, Xtest)`` path)."""
from .utils import FastaUtility, Vocabulary  # noqa: F401

__all__ = ["FastSK", "FastaUtility", "Vocabulary"]


def __getattr__(name):
    # The pybind11 class needs the HIP library; import it lazily so that the pure-Python helpers
    # (FASTA reader, ctypes view) stay importable on a box where the engine is not built yet.
    if name == "FastSK":
        from ._native import share_hip_runtime_with_torch
        share_hip_runtime_with_torch()  # before the extension pulls in libamdhip64
        from ._fastsk import FastSK
        return FastSK
    raise AttributeError(name)

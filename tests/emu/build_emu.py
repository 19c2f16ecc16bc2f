#!/usr/bin/env python3
"""Build tests/emu/libfastsk_emu.so: the engine's kernel source compiled for the CPU against
hip_emu.h (TEST INFRASTRUCTURE ONLY — see hip_emu.h). Used by `pytest -m "not gpu"`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "fastsk_amd", "csrc", "fsk_engine.hip")
OUT = os.path.join(HERE, "libfastsk_emu.so")


def _current():
    deps = [SRC, os.path.join(HERE, "hip_emu.h"), os.path.join(ROOT, "include", "fastsk_amd.h")]
    deps += [os.path.join(ROOT, "fastsk_amd", "csrc", f)
             for f in ("fsk_kernels.h", "fsk_tile_kernel.inc", "fsk_tile_kernel_dma.inc", "fsk_sparse_kernels.inc", "fsk_platform.h", "fsk_fasta.cpp")]
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)


def build(force=False):
    """Several test processes may ask at once (the gloo workers of tests/test_distributed.py): one
    compiles under a lock into a temporary file and renames it, the others wait and reuse it."""
    import fcntl
    if not force and _current():
        return OUT
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and _current():  # somebody else built it while we waited
            return OUT
        tmp = OUT + ".tmp%d" % os.getpid()
        cmd = ["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-pthread", "-DFSK_EMU", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-I", HERE, "-x", "c++", SRC,
               os.path.join(ROOT, "fastsk_amd", "csrc", "fsk_fasta.cpp"), "-o", tmp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("emu build failed:\n" + r.stderr)
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

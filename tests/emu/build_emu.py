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


def build(force=False):
    deps = [SRC, os.path.join(HERE, "hip_emu.h"), os.path.join(ROOT, "include", "fastsk_amd.h")]
    deps += [os.path.join(ROOT, "fastsk_amd", "csrc", f) for f in ("fsk_kernels.h", "fsk_tile_kernel.inc", "fsk_platform.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-pthread", "-DFSK_EMU", "-ffp-contract=off",
           "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-I", HERE, "-x", "c++", SRC, "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emu build failed:\n" + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

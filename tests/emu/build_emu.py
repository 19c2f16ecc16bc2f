#!/usr/bin/env python3
"""Build tests/emu/libfastsk_emu.so: the engine's kernel source compiled for the CPU against
hip_emu.h (TEST INFRASTRUCTURE ONLY — see hip_emu.h). Used by `pytest -m "not gpu"`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "fastsk_amd", "csrc")
OUT = os.path.join(HERE, "libfastsk_emu.so")


def _current():
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".cpp"))] + [os.path.join(ROOT, "include", "fastsk_amd.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc", ".cpp")) and f != "bindings.cpp"]
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)


def units():
    """The translation units of libfastsk_amd.so (everything but the pybind11 module) — and, in this library ONLY, the stand-in
    for librccl that the emulated build's RCCL collective is bound to (rccl_stub.cpp)."""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")) + [os.path.join(CSRC, "fsk_fasta.cpp"),
                                                                                               os.path.join(HERE, "rccl_stub.cpp")]


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
        from concurrent.futures import ThreadPoolExecutor
        tmp = OUT + ".tmp%d" % os.getpid()
        objs = []

        def compile_unit(src):
            obj = "%s.%s.o" % (tmp, os.path.splitext(os.path.basename(src))[0])
            objs.append(obj)
            cmd = ["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-pthread", "-DFSK_EMU", "-DFSK_TEST_HOOKS", "-ffp-contract=off", "-Wall",
                   "-Wno-unused-function", "-Wno-unknown-pragmas", "-I", HERE, "-x", "c++", "-c", src, "-o", obj]
            return subprocess.run(cmd, capture_output=True, text=True)
        try:
            with ThreadPoolExecutor(max_workers=4) as pool:
                results = list(pool.map(compile_unit, units()))
            bad = [r for r in results if r.returncode != 0]
            if not bad:
                r = subprocess.run(["g++", "-shared", "-pthread"] + objs + ["-ldl", "-o", tmp], capture_output=True, text=True)
                bad = [r] if r.returncode != 0 else []
            if bad:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("emu build failed:\n" + "\n".join(r.stderr for r in bad))
            os.replace(tmp, OUT)
        finally:
            for o in objs:
                if os.path.exists(o):
                    os.remove(o)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

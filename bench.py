#!/usr/bin/env python3
"""bench.py — headline benchmark of the gapped-k-mer kernel build (BASELINE.json metric).

A "step" is one pass of the hot path over the whole workload with the packed sequences already
resident in HBM when the timed region starts:
  --config 5 (default)  BASELINE config 5: synthetic 100,000 x 300 bp DNA, g=12, m=8, exact, all
                        C(12,8)=495 mismatch combinations (dense dataflow);
  --config 4            BASELINE config 4: protein 2.19 (tests/golden/tokens_2.19.npz), g=14, m=10,
                        exact, C(14,10)=1001 combinations (sparse dataflow: sort -> segments -> pairs).
With N GPUs the job is the same (STRONG scaling: total work fixed). `python bench.py --gpus N` starts
its own ranks (torch.distributed.run, one process per GPU, before this process touches the GPU);
under an external torch.distributed.run (WORLD_SIZE set) it is one of the ranks. Two decompositions
(fastsk_amd/distributed.py), both timed with the same --steps/--warmup:
  combos  (`value`)  the reference's own decomposition (fastsk_kernel.cpp:148,275,286-315) and the
          one BASELINE.json names: combos c = rank (mod N), every rank accumulates a private
          triangle, ONE logical RCCL all-reduce sums them over xGMI — issued in row bands on RCCL's
          stream under the next band's kernels;
  rows    (`alt`; config 5 only) every rank owns an equal-area band of rows of the triangle and runs
          all combos over it: no cell is shared, the only exchange is the 0.8 MB diagonal, and the
          kernel matrix stays distributed.
--shard rows swaps the two. value = combos/s of the whole job = C(g,m) * steps / max-over-ranks seconds.
`--gpus N --inproc` runs the same job in ONE process: fastsk.FastSK(devices=[0..N-1])'s engine (fsk_create_multi:
a host thread, a compute stream and an exchange stream per GPU, RCCL called from the host C++); a multi-process
line carries that measurement as `inproc` (rank 0 starts it as a fresh child process once the ranks have
released their GPUs).

Every line carries `k_digest`: an order-free digest of the integer triangle computed on the device(s) after the
timed steps (fsk_counts_digest). profiles/k_digests.json holds the single-GPU digests of the BASELINE workloads;
a multi-GPU line compares its own — on every rank, after the reduce — with them: `bit_identical_to_1gpu`
(BASELINE.md's "result identical at 1/2/4/8 GPUs" gate), and the run exits non-zero when that is false.

One JSON line on rank 0. Extra objects:
  roofline      the dominant kernel. Config 5: k_dense_tile_dma is bound by the integer-VALU issue
                rate (v_dot8_u32_u4), so `frac` = count-MACs/s over that peak; the SURVEY 8(d)
                algorithmic bytes (16*U + sort + input per combo), the measured HBM traffic and the
                useful-update rate are given beside it. Config 4: the sparse pipeline against the
                8 TB/s HBM peak on the same algorithmic bytes.
  cpu_baseline  FastSK's own multithreaded engine (oracle/_ref, kind "reference"; our C restatement
                as "port" when the compiled reference did not travel) on the host cores at
                N = 4000 and 8000 of the same generator, T = physical cores and T = 20; rank 0, N=1 only.
  end_to_end    load (pack + H2D) + one step + the WHOLE normalised triangle on the device: SURVEY 8(d)'s metric boundary.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# the host driver only supports dmabuf IPC: RCCL between processes needs this before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
# v_dot8_u32_u4 / v_dot4_u32_u8 issue at HALF the v_fma_f32 rate on gfx950 (measured:
# profiles/ubench_valu_rates.txt): 64 lanes/clk/CU. Peak = CUs * 64 lanes * 8 MACs * 2.4 GHz.
VALU_DOT8_PEAK_TMACS = 256 * 64 * 8 * 2.4e9 / 1e12  # = 314.6 T MAC/s
KERNEL_FILES = ("fastsk_amd/csrc/fsk_kernels_dense.h", "fastsk_amd/csrc/fsk_tile_kernel_dma.inc")


def synthetic(N, L, seed=20201214):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.integers(1, 5, size=(N, L), dtype=np.int32)
    return X.reshape(-1), np.arange(N + 1, dtype=np.int64) * L, X


def git_blob_hash(path):
    """What `git hash-object` prints (needs no git on the GPU box)."""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def kernel_hashes(files=KERNEL_FILES):
    return {f: git_blob_hash(os.path.join(ROOT, f)) for f in files}


# the sources of the sparse pipeline (kernels + the host code that batches and sizes them)
SPARSE_FILES = ("fastsk_amd/csrc/fsk_sparse_kernels.inc", "fastsk_amd/csrc/fsk_sparse_blocks.inc", "fastsk_amd/csrc/fsk_common.h",
                "fastsk_amd/csrc/fsk_engine_sparse.hip")


def host_cpus():
    """(physical cores, logical CPUs, model name) from /proc/cpuinfo."""
    cores, model, phys, logical = set(), "", None, 0
    try:
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                logical += 1
            elif key == "model name" and not model:
                model = val
            elif key == "physical id":
                phys = val
            elif key == "core id":
                cores.add((phys, val))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return (len(cores) or logical), logical, model


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 1 << 36


def cpu_baseline(g, m, L, n_full, budget_s, sizes=(4000, 8000)):
    """FastSK's own engine on the host cores: the config-5 generator at N = 4000 and 8000,
    T = 20 (the reference's default, fastsk_kernel.cpp:55-60) and T = physical cores. Each row runs
    KernelFunction::compute_kernel itself (thread pool, private uint32 triangles, locked reduce,
    normalisation) on T combos — approx/skip_variance/max_iters=1 makes every thread take exactly one
    combo, i.e. the budget is kept by trimming combos, never N. A row predicted (from the rows before
    it: time ~ N^2 per combo, and no better than proportional to the combos) not to fit the remaining
    budget is skipped and says so."""
    from oracle import loader
    phys, logical, model = host_cpus()
    kind = "reference" if loader.have_ref() else "port"
    ncomb = int(loader.port().num_combos(g, m))
    rows, spent, per_unit = [], 0.0, None  # per_unit: seconds per (N^2 * combo) of the slowest row so far
    # T = 20 at both sizes first, then ALL PHYSICAL CORES AT THE LARGEST N (SURVEY 8d asks for that row in the same run;
    # measured on the 2 x 64-core EPYC 9575F of the GPU box: 128 threads take 7.4x as long as 20 for 6.4x the combos — the
    # reference is memory-bound, more threads buy nothing — so that row needs 150-200 s and the default budget covers it);
    # only when it does not fit the budget is T = physical cores taken at the smaller size instead
    plan = [(n, 20) for n in sizes] + [(n, phys) for n in reversed(sizes)]
    have_phys = False
    for n, T in plan:
        T = max(1, min(T, ncomb))
        if T == max(1, min(phys, ncomb)) and T != 20 and have_phys:
            continue  # (the row at the largest N ran: the smaller size adds nothing)
        tokens, offsets, _ = synthetic(n, L)
        pairs = n * (n + 1) // 2
        need = T * pairs * 4 + pairs * 16 + n * L * 64
        predicted = None if per_unit is None else per_unit * n * n * T
        row = {"n": n, "threads": T, "combos": T}
        if need > 0.6 * mem_available_bytes():
            row["skipped"] = "needs %.0f GB of host memory for %d private triangles" % (need / 1e9, T)
        elif predicted is not None and spent + predicted > budget_s:
            row["skipped"] = "predicted %.0f s, %.0f s of the %.0f s budget left (rerun with --cpu-seconds)" % (
                predicted, max(0.0, budget_s - spent), budget_s)
        else:
            t0 = time.perf_counter()
            if kind == "reference":
                loader.ref().full_triangle(tokens, offsets, n, 0, g, m, t=T, approx=True, skip_variance=True, max_iters=1, seed=1)
            else:
                loader.port().raw_counts(tokens, offsets, g, m, np.arange(T, dtype=np.int32), threads=T, want_counts=False)
            dt = time.perf_counter() - t0
            spent += dt
            row.update(seconds=dt, combos_per_s=T / dt,
                       extrapolated_combos_per_s_at_full_n=T / dt * (n / n_full) ** 2)
            per_unit = max(per_unit or 0.0, dt / (n * n * T))
            have_phys |= T != 20
        rows.append(row)
    done = [r for r in rows if "combos_per_s" in r]
    best = max(done, key=lambda r: (r["n"], r["combos_per_s"])) if done else None
    return {
        "value": best["extrapolated_combos_per_s_at_full_n"] if best else None, "unit": "combos/s",
        "cores": best["threads"] if best else 0, "kind": kind, "cpu_model": model,
        "physical_cores": phys, "logical_cpus": logical, "rows": rows, "seconds_spent": spent,
        "sample": ("config-5 generator at N = %s, %d bp, g=%d m=%d; every row = one call of the %s engine on T combos "
                   "(one per thread); value = the fastest row at the largest N, x (N/%d)^2 (count time scales as N^2; "
                   "the reference itself cannot index N > 46340)" % (
                       "/".join(str(s) for s in sizes), L, g, m,
                       "reference's KernelFunction" if kind == "reference" else "restated", n_full)),
    }


def live_traffic(args, dense):
    """HBM bytes of the dominant kernel (dense: the tile launch; sparse: one pass of the pipeline), measured NOW: two child
    runs of this script (`--one-pass`: one fsk_compute of the workload, nothing else) under `rocprofv3 --pmc` — FETCH_SIZE,
    then WRITE_SIZE, never together; `--kernel-trace` only, the program itself after `--`, cwd /tmp. gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE reads
    half of 16-byte streaming fetches, so it is doubled (an upper bound for narrow reads); both counters are KiB.
    Returns (bytes or None, note)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        return None, "this run is itself under a profiler"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="fsk_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--one-pass", "--config", str(args.config), "--n-seq", str(args.n_seq), "--seq-len", str(args.seq_len)]
        if args.g is not None:
            cmd += ["-g", str(args.g)]
        if args.m is not None:
            cmd += ["-m", str(args.m)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s pass failed (rc %s): %s" % (counter, r.returncode, (r.stderr or "")[-160:].replace("\n", " "))
            per_kernel = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and "fsk::" in row["Kernel_Name"]:
                        name = row["Kernel_Name"].split("(")[0]
                        per_kernel.setdefault(name, []).append(float(row["Counter_Value"]))
            if dense:
                tile = [v for k, vs in per_kernel.items() if "k_dense_tile" in k for v in vs]
                if not tile:
                    return None, "no k_dense_tile dispatch in the --pmc %s pass" % counter
                got[counter] = max(tile)
            else:
                got[counter] = sum(v for k, vs in per_kernel.items() if "k_sx_" in k for v in vs)
        except Exception as exc:
            return None, "rocprofv3 --pmc %s pass: %r" % (counter, exc)
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    return got["FETCH_SIZE"] * 1024.0 * 2.0 + got["WRITE_SIZE"] * 1024.0, (
        "measured in THIS run: two child passes of this command under rocprofv3 --pmc (FETCH_SIZE %.0f KiB x 2 for the gfx950 "
        "16-byte-fetch correction + WRITE_SIZE %.0f KiB), %s" % (got["FETCH_SIZE"], got["WRITE_SIZE"],
        "the 495-combo tile launch" if dense else "summed over the sparse pipeline's kernels of one pass"))


def config_roofline(st, wall_s):
    """SURVEY 8(d) for one whole fsk_compute call: algorithmic bytes = 16*U + per combo (16*P*nfeat + packed
    input), over the wall time of the call, as a fraction of the HBM peak (sparse dataflow); the dense
    dataflow's tile kernel is priced against the v_dot8 issue peak (count-MACs / tile-kernel time)."""
    keybits = max(1, int(np.ceil(np.log2(max(2, st["key_space"])))))
    P = (keybits + 7) // 8
    # (variance mode runs batches of iterations ahead of its stop test and drops what lies beyond the stop: the
    # update count of the combos that make up the result is the issued combos' pro rata)
    useful = st["combos_done"] / st["combos_issued"] if st.get("combos_issued") else 1.0
    alg = 16.0 * st["cell_updates"] * useful + st["combos_done"] * (16.0 * P * st["n_feat"] + st["n_feat"] * st["bits_per_symbol"] / 8.0)
    out = {"path": "dense" if st["path_used"] == 1 else "sparse", "cell_updates_U": int(st["cell_updates"] * useful),
           "combos_issued": int(st.get("combos_issued", st["combos_done"])),
           "algorithmic_GB": alg / 1e9, "algorithmic_GBs": alg / 1e9 / wall_s, "gpu_ms": 1e3 * wall_s,
           "frac_of_hbm_peak": alg / 1e9 / wall_s / HBM_PEAK_GBS,
           "kernel_ms": {k[3:]: round(st[k], 3) for k in ("ms_count", "ms_tile", "ms_extract", "ms_sort", "ms_segment", "ms_pairs") if st[k]}}
    if st["path_used"] == 1 and st["ms_tile"] > 0:
        out["valu_frac"] = st["dense_macs"] / (st["ms_tile"] * 1e-3) / 1e12 / VALU_DOT8_PEAK_TMACS   # over the tile launch (HIP events)
        out["valu_frac_of_call"] = st["dense_macs"] / wall_s / 1e12 / VALU_DOT8_PEAK_TMACS         # over the whole fsk_compute call (load, counting, tile launch, finalize)
        out["bound"] = "valu (v_dot8 issue); frac_of_hbm_peak prices a dataflow this kernel does not run"
    else:
        out["bound"] = "hbm"
    return out


def other_configs(_native):
    """The other BASELINE configs on the same GPU: the whole fsk_compute call, host buffers in, result
    resident on the device (best of 4), and its roofline (one more call with HIP-event timing and the
    exact update count U). Config 2 = BASELINE configs[1] (EP300 DNA, 2000+2000 x 100 bp, g=10 m=6 exact);
    configs 1, 3, 4 from the golden descriptors (same modes and combo orders as the parity tests)."""
    out = {}

    def measure(make, tokens, offsets, ntr, nte):
        e = make(False)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            e.compute(tokens, offsets, ntr, nte)
            best = min(best, time.perf_counter() - t0)
        done = int(e.stats()["combos_done"])
        try:
            digest = digest_hex(e.counts_digest())
        except _native.FskError:
            digest = None  # (variance mode keeps a floating-point mean, not integer counts)
        e.close()
        e = make(True)
        e.compute(tokens, offsets, ntr, nte)  # (allocations)
        e.compute(tokens, offsets, ntr, nte)
        st = e.stats()
        e.close()
        return best, done, config_roofline(st, best), digest

    path = os.path.join(ROOT, "tests", "golden", "tokens_EP300.npz")
    if os.path.exists(path):
        z = np.load(path)
        tokens, offsets = z["tokens"].astype(np.int32), z["offsets"].astype(np.int64)
        ntr, nte = int(z["n_train"]), int(z["n_test"])
        best, done, roof, digest = measure(lambda prof: _native.Engine(10, 6, profile=prof), tokens, offsets, ntr, nte)
        out["config2_ep300_exact"] = {"n_seq": ntr + nte, "seq_len": 100, "g": 10, "m": 6, "combos": 210, "seconds": best,
                                      "combos_per_s": 210 / best, "reference_cpu_seconds_8_threads": 29.9, "roofline": roof,
                                      "k_digest": digest}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import load_golden, load_tokens, GOLD
        for key, name in (("config1_prot11_approx_t1", "f7_cfg1_prot11_approx_t1"), ("config3_ep47848_100combos", "f7_cfg3_ep47848_100combos"),
                          ("config4_prot219_exact", "f7_cfg4_prot219_exact")):
            if not os.path.exists(os.path.join(GOLD, name + ".npz")):
                continue
            d = load_golden(name)
            tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])

            def make(prof, d=d):
                e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"],
                                   skip_variance=bool(d["skip_variance"]), profile=prof)
                if d["approx"]:
                    e.set_combo_order(d["order"])
                return e
            best, done, roof, digest = measure(make, tokens, offsets, ntr, nte)
            out[key] = {"n_seq": ntr + nte, "g": int(d["g"]), "m": int(d["m"]), "combos": done, "seconds": best,
                        "combos_per_s": done / best, "reference_cpu_seconds": float(d["ref_seconds"]), "roofline": roof,
                        "k_digest": digest}
    except Exception as exc:  # the headline line must not depend on the extras
        out["error"] = repr(exc)
    # the sparse dataflow beyond the owner bands (N = 32k / 64k / 100k protein-like, DNA k = 8 at N = 32k): ms a combo, U,
    # algorithmic bytes, fraction of the HBM roofline and of the 64-bit-atomic ceiling, the form the update stage took
    # (tools/bench_sparse_large_n.py; 20 combos each, a few seconds in all)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_sparse_large_n as large_n
        out["sparse_large_n"] = {name: large_n.run(name, 20) for name in large_n.DEFAULT}
    except Exception as exc:
        out["sparse_large_n_error"] = repr(exc)
    # the reference paper's large-g regime (EP300, k = 6, g = 16 and 20: 8,008 and 38,760 combos; tools/bench_large_g.py):
    # seconds a whole kernel, U, algorithmic bytes and the fraction of the HBM roofline (the tool run by itself also checks a 12-combo subset against the oracle)
    try:
        import bench_large_g as large_g
        out["large_g"] = {"g%d_m%d" % gm: large_g.run(*gm, check=False) for gm in ((16, 10), (20, 14))}
    except Exception as exc:
        out["large_g_error"] = repr(exc)
    return out or None


# ---- digests of the integer triangle (fsk_counts_digest): the bit-identity gate of multi-GPU runs
DIGESTS = os.path.join(ROOT, "profiles", "k_digests.json")


def digest_hex(d):
    return {"sum": "%016x" % d[0], "xor": "%016x" % d[1]}


def digest_key(config, N, L, g, m, ncomb):
    return "config%d:n_seq=%d,seq_len=%s,g=%d,m=%d,combos=%d" % (config, N, L, g, m, ncomb)


def committed_digest(key):
    try:
        return json.load(open(DIGESTS)).get(key)
    except (OSError, ValueError):
        return None


def describe(mode, world, replicate):
    if world == 1:
        return "single GPU"
    if mode == "rows":
        return ("row-band sharded x%d: every rank runs all combos over its own equal-area band of rows of the triangle; "
                "exchange = the N-entry diagonal only%s" % (world, "; finished bands broadcast to every rank" if replicate
                                                            else "; the kernel matrix stays distributed"))
    return ("combo-sharded x%d (combos c = rank mod %d, private triangles) + one RCCL all-reduce of the triangle, issued in "
            "row bands under the next band's kernels" % (world, world))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Watchdog:
    """Fail fast, and always leave a line (multi-GPU first contact must not burn the lease silently).

    A daemon thread per rank. `stage(name, bound_s)` names what the main thread is doing and how long it may take;
    when a stage outlives its bound — a collective waiting for a rank that never comes, a communicator that never
    initialises — rank 0 prints ONE JSON line {"error", "stage", "n_gpus", "timings", ...} on stdout and the process
    ends with os._exit(2): no re-exec, no restart of a process that has touched the GPU, no Python-level cleanup that
    could itself block on the device. The other ranks wait `grace_s` longer than rank 0 so that rank 0's line is out
    before the launcher, seeing a rank die, terminates the rest; and when the launcher's SIGTERM arrives first (some
    other rank crashed), the thread — woken through the signal wake-up pipe: the main thread may be stuck inside a
    collective, where no Python handler would run — writes the same line with what it knows and exits 143.
    """

    def __init__(self, rank, world, args, grace_s=20.0):
        import signal
        import threading
        self.rank, self.world, self.args = rank, world, args
        self.grace = 0.0 if rank == 0 else grace_s
        self.lock = threading.Lock()
        self.name, self.t0, self.bound = "start", time.perf_counter(), None
        self.timings = {}          # finished stages: seconds
        self.partial = {}          # whatever the main thread wants reported on failure
        self.done = False
        self._signal = signal
        # SIGTERM: a do-nothing Python handler makes the signal "caught" (so it no longer kills the process), and
        # CPython's C-level handler — which runs at once, on whichever thread the kernel picks, while the main thread
        # may be stuck inside a collective where no Python handler can run — writes the signal number to this pipe;
        # the watchdog thread selects on it.
        self.sig_r = None
        try:
            r, w = os.pipe()
            os.set_blocking(w, False)
            os.set_blocking(r, False)
            signal.set_wakeup_fd(w, warn_on_full_buffer=False)  # (first: if this fails, SIGTERM keeps its default action)
            signal.signal(signal.SIGTERM, lambda *a: None)
            self.sig_r = r
        except (AttributeError, ValueError, OSError):
            self.sig_r = None
            try:
                signal.signal(signal.SIGTERM, signal.SIG_DFL)
            except (ValueError, OSError):
                pass
        self.thread = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
        self.thread.start()

    def stage(self, name, bound_s):
        now = time.perf_counter()
        with self.lock:
            self.timings[self.name] = round(self.timings.get(self.name, 0.0) + now - self.t0, 4)
            self.name, self.t0, self.bound = name, now, bound_s

    def finish(self):
        with self.lock:
            self.done = True

    def _line(self, why, code):
        with self.lock:
            out = {"error": why, "stage": self.name, "stage_seconds": round(time.perf_counter() - self.t0, 3),
                   "stage_bound_seconds": self.bound, "n_gpus": self.world, "rank": self.rank,
                   "metric": "gkm kernel build: mismatch-combos/s", "value": None, "unit": "combos/s",
                   "steps": self.args.steps, "warmup": self.args.warmup, "timings": dict(self.timings), "partial": dict(self.partial)}
        try:
            if self.rank == 0:
                sys.stdout.write(json.dumps(out) + "\n")
                sys.stdout.flush()
            sys.stderr.write("bench.py rank %d: %s (stage %r)\n" % (self.rank, why, out["stage"]))
            sys.stderr.flush()
        finally:
            os._exit(code)

    def _run(self):
        import select
        while True:
            got = False
            if self.sig_r is not None:
                try:
                    ready, _, _ = select.select([self.sig_r], [], [], 0.25)
                    if ready:
                        got = self._signal.SIGTERM in os.read(self.sig_r, 64)
                except (InterruptedError, OSError, ValueError):
                    got = False
            else:
                time.sleep(0.25)
            with self.lock:
                if self.done:
                    if got:
                        os._exit(143)
                    continue
                name, t0, bound = self.name, self.t0, self.bound
            if got:
                self._line("terminated by the launcher (SIGTERM) — another rank failed or the run was cancelled", 143)
            if bound is not None and time.perf_counter() - t0 > bound + self.grace:
                self._line("stage %r exceeded its bound of %.0f s: a rank, a link or a collective is not answering" % (name, bound), 2)


def fatal(wd, rank, world, args, why, stage, extra=None):
    """A check failed (not a hang): rank 0 prints the JSON error line, every rank exits non-zero."""
    out = {"error": why, "stage": stage, "n_gpus": world, "rank": rank, "metric": "gkm kernel build: mismatch-combos/s",
           "value": None, "unit": "combos/s", "steps": args.steps, "warmup": args.warmup,
           "timings": dict(wd.timings) if wd else {}, "partial": dict(extra or {})}
    if rank == 0:
        print(json.dumps(out), flush=True)
    sys.stderr.write("bench.py rank %d: %s\n" % (rank, why))
    sys.stderr.flush()
    os._exit(2)


def expected_step_seconds(args):
    """What one step of the workload takes on ONE MI355X (profiles/: config 5 at 100k x 300 = 2.12 s, ~N^2; config 4
    = 10 ms): the yardstick the watchdog's bounds are derived from until this run has measured its own step."""
    if args.config == 4:
        return 0.05
    return 2.2 * (args.n_seq / 100000.0) ** 2 * (args.seq_len / 300.0) + 0.05


def memory_plan(N, pairs, world, dense, narrow_possible):
    """Bytes one rank needs on its GPU: the integer triangle, the int32 staging of the largest exchanged band, the
    count panels (the engine takes up to 32 GB or 60 % of what is free), headroom."""
    k = pairs * 8
    staging = (pairs * 4 // 8 if world > 1 and narrow_possible else 0)   # <= the largest of >= 8 equal-area bands
    panels = (13 << 30) if (dense and N >= 50000) else (2 << 30)
    return {"triangle_u64": k, "exchange_staging": staging, "count_panels_or_sort_scratch": panels, "headroom": 2 << 30,
            "total": k + staging + panels + (2 << 30)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=[4, 5], default=5, help="BASELINE config: 5 = synthetic DNA (default), 4 = protein 2.19")
    ap.add_argument("--n-seq", type=int, default=100000)
    ap.add_argument("--seq-len", type=int, default=300)
    ap.add_argument("-g", type=int, default=None)
    ap.add_argument("-m", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=260.0, help="budget of the CPU baseline (rows that do not fit are skipped)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra measurements of configs 1-4 (profiling runs)")
    ap.add_argument("--one-pass", action="store_true", help=argparse.SUPPRESS)  # (live_traffic's child: one fsk_compute, nothing else)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from profiles/traffic*.json instead of two rocprofv3 --pmc child passes of this command")
    ap.add_argument("--bands", type=int, default=None, help="row bands of the overlapped all-reduce (default: auto)")
    ap.add_argument("--shard", choices=["combos", "rows"], default="combos",
                    help="multi-GPU decomposition reported as `value` (the other one is `alt`)")
    ap.add_argument("--replicate", action="store_true", help="rows: broadcast finished bands so every rank holds all of K")
    ap.add_argument("--no-alt", action="store_true", help="multi-GPU: do not time the other decomposition")
    ap.add_argument("--inproc", action="store_true",
                    help="run the N GPUs from ONE process: FastSK(devices=[0..N-1])'s engine (fsk_create_multi), RCCL from the host C++")
    ap.add_argument("--collective", choices=["auto", "rccl", "p2p"], default="auto", help="--inproc: the exchange (auto = RCCL)")
    ap.add_argument("--no-inproc-leg", action="store_true", help="multi-process run: do not add the in-process measurement")
    ap.add_argument("--inproc-timeout", type=float, default=420.0, help="seconds the in-process child of a multi-process run may take")
    ap.add_argument("--no-preflight", action="store_true", help="multi-GPU: skip the 1-step N = 16000 job whose digest is committed")
    ap.add_argument("--init-timeout", type=float, default=180.0, help="seconds init_process_group / a collective may take (torch timeout)")
    ap.add_argument("--no-watchdog", action="store_true", help="no stage bounds (debugging under a profiler)")
    ap.add_argument("--engine-lib", default=None, help=argparse.SUPPRESS)  # (tests: a build of the engine with fault hooks, tests/hooks)
    ap.add_argument("--step-bound", type=float, default=0.0, help="seconds one step may take before the watchdog gives up (default: from the 1-GPU step time)")
    return ap.parse_args()


def workload_of(args):
    """(tokens, offsets, N, L, g, m, workload text, data text)"""
    if args.config == 4:
        z = np.load(os.path.join(ROOT, "tests", "golden", "tokens_2.19.npz"))
        tokens, offsets = z["tokens"].astype(np.int32), z["offsets"].astype(np.int64)
        N, L = len(offsets) - 1, None
        g, m = args.g or 14, args.m or 10
        workload = "config4: protein 2.19, %d sequences (mean length %d), g=%d m=%d exact" % (N, int(offsets[-1] // N), g, m)
        data = "real (data/2.19 FASTA tokens, tests/golden/tokens_2.19.npz)"
    else:
        N, L = args.n_seq, args.seq_len
        g, m = args.g or 12, args.m or 8
        tokens, offsets, _ = synthetic(N, L)
        workload = "config5: synthetic DNA %d x %d bp, g=%d m=%d exact" % (N, L, g, m)
        data = "synthetic"
    return tokens, offsets, N, L, g, m, workload, data


def roofline_of(args, s0, s1, dense, world, combos_rank, share_rows, offsets, n_mine, prof=None):
    """`roofline`, `phases_ms_per_step`, `dtype` from the engine's stats before / after the timed steps (HIP events recorded
    around every kernel family of those steps: fsk_config.profile = 2, the product dataflow). `prof` = (before, after, launches
    of one step) of ONE extra step in measurement mode (profile = 1), run after the timed region: the exact update count U and
    the count-MACs with the flagged rows' remainder products, which the product dataflow does not compute."""
    d = lambda k: s1[k] - s0[k]
    per_step = lambda k: ((prof[1][k] - prof[0][k]) if prof else d(k) / max(1, args.steps))
    N = s1["n_seq"]
    nfeat = s1["n_feat"]
    b_in = (int(offsets[-1]) * s1["bits_per_symbol"] + 7) // 8
    keybits = max(1, int(np.ceil(np.log2(max(2, s1["key_space"])))))
    P = (keybits + 7) // 8  # 8-bit LSD passes SURVEY 8(d) prices the sort with
    if dense:
        # ---- roofline of the dominant kernel (tile accumulate), per launch
        launches = max(1, d("n_tile_launches"))
        tile_ms = d("ms_tile") / launches
        # exact, from the count panels (whole triangle); a row-band launch owns its share of the cells
        launches_per_step = max(1.0, launches / max(1, args.steps))
        U = per_step("cell_updates") / launches_per_step * share_rows
        combos_per_launch = combos_rank / launches
        alg_bytes = 16.0 * U + combos_per_launch * (b_in + 16.0 * P * nfeat)
        secs = tile_ms * 1e-3
        alg_gbs = alg_bytes / secs / 1e9 if secs > 0 else 0.0
        macs = per_step("dense_macs") / launches_per_step
        tmacs = macs / secs / 1e12 if secs > 0 else 0.0
        traffic, traffic_note = None, "no profiles/traffic.json for this launch shape"
        live_note = None
        if world == 1 and not args.no_live_traffic and not args.inproc and launches == args.steps:  # (one launch per step: the un-banded form)
            traffic, live_note = live_traffic(args, True)
            traffic_note = live_note
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if traffic is None and world == 1 and os.path.exists(tpath):  # (measured on the single-GPU launch: does not describe a rank's band)
            try:
                tj = json.load(open(tpath))
                if tj.get("n_seq") != N or tj.get("combos_per_launch") != int(combos_per_launch):
                    traffic_note = "profiles/traffic.json describes another launch shape"
                elif tj.get("kernel_files") != kernel_hashes():
                    traffic_note = "profiles/traffic.json was measured on other kernel sources (blob hashes differ): refused"
                else:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_note = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, NOT this run), commit %s" % tj.get("commit", "?")
            except Exception as exc:
                traffic_note = "profiles/traffic.json unreadable: %r" % exc
            if live_note:
                traffic_note += "; live measurement: " + live_note
        roofline = {
            "bound": "valu", "kernel": "k_dense_tile_dma",
            "achieved": tmacs, "peak": VALU_DOT8_PEAK_TMACS, "unit": "T count-MAC/s (v_dot8_u32_u4: 64 lanes/clk/CU x 8 MACs, 256 CUs, 2.4 GHz)",
            "frac": tmacs / VALU_DOT8_PEAK_TMACS, "traffic": traffic, "traffic_source": traffic_note,
            "hbm_measured_frac": (traffic / secs / 1e9 / HBM_PEAK_GBS) if (traffic and secs > 0) else None,
            "hbm_peak_GBs": HBM_PEAK_GBS,
            "useful_update_frac": (U / secs / 1e12) / VALU_DOT8_PEAK_TMACS if secs > 0 else 0.0,
            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_equiv_GBs": alg_gbs,
            "algorithmic_x_hbm_peak": alg_gbs / HBM_PEAK_GBS,
            "launch_ms": tile_ms, "launches": int(launches), "combos_per_launch": combos_per_launch,
            "cell_updates_per_launch": U, "count_macs_per_launch": macs,
            "note": "the tile kernel computes K += sum_v cnt_i(v) cnt_j(v) as on-chip integer dot products, so the binding ceiling is "
                    "the v_dot8 issue rate (frac); useful_update_frac = the reference's `+=` count U per second over the same peak "
                    "(the rest of the MACs multiply by a zero count); algorithmic_* = SURVEY 8(d) bytes of the direct-atomic dataflow "
                    "(16*U + sort + input per combo), which this kernel does NOT move through HBM — not a fraction of anything",
        }
        phases = {"count": d("ms_count") / args.steps, "tile": d("ms_tile") / args.steps, "accumulate_total": d("ms_total") / args.steps}
        dtype = "u4 count planes (v_dot8_u32_u4), u32 register sums, u64 triangle (plain stores on the first pass over reset rows, else atomics)"
    else:
        # ---- sparse pipeline: SURVEY 8(d) algorithmic bytes over the GPU time of the pipeline
        U = d("cell_updates") / max(1, args.steps)   # (the sparse dataflow counts its `+=` in every mode)
        alg_bytes = 16.0 * U + n_mine * (b_in + 16.0 * P * nfeat)
        gpu_ms = (d("ms_extract") + d("ms_sort") + d("ms_segment") + d("ms_pairs")) / args.steps
        alg_gbs = alg_bytes / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else 0.0
        fam = {"extract": d("ms_extract"), "sort": d("ms_sort"), "segment": d("ms_segment"), "pairs": d("ms_pairs")}
        # measured HBM bytes of one config-4 pass: the sum over the pipeline's kernels of the rocprofv3 --pmc passes in
        # profiles/ (FETCH_SIZE x2 + WRITE_SIZE), accepted only for this workload and these sources
        traffic, traffic_note = None, "not collected in this run (profiles/ holds the rocprofv3 --pmc passes of the sparse kernels)"
        live_note = None
        if args.config == 4 and world == 1 and not args.no_live_traffic and not args.inproc:
            traffic, live_note = live_traffic(args, False)
            traffic_note = live_note
        tpath = os.path.join(ROOT, "profiles", "traffic_config4.json")
        if traffic is None and args.config == 4 and world == 1 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("combos") != n_mine:
                    traffic_note = "profiles/traffic_config4.json describes another combo count"
                elif tj.get("kernel_files") != kernel_hashes(SPARSE_FILES):
                    traffic_note = "profiles/traffic_config4.json was measured on other sources (blob hashes differ): refused"
                else:
                    traffic = tj.get("hbm_bytes_per_step")
                    traffic_note = "profiles/traffic_config4.json (sum over the pipeline's kernels of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, NOT this run), commit %s" % tj.get("commit", "?")
            except Exception as exc:
                traffic_note = "profiles/traffic_config4.json unreadable: %r" % exc
            if live_note:
                traffic_note += "; live measurement: " + live_note
        roofline = {
            "bound": "hbm", "kernel": "sparse pipeline (k_sx_extract, k_sx_scan_slot/scatter, k_sx_seg_*, k_sx_emit + k_sx_consume); largest family: %s" % max(fam, key=fam.get),
            "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_note,
            "hbm_measured_frac": (traffic / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and gpu_ms > 0) else None,
            "algorithmic_bytes_per_step": alg_bytes, "cell_updates_per_step": U, "gpu_ms_per_step": gpu_ms,
            "sort_passes_priced": P, "sort_passes_run": s1["sort_passes"],
            "note": "algorithmic bytes = 16*U + 16*P*nfeat + input per combo (SURVEY 8d) over the HIP-event time of the pipeline's kernels",
        }
        phases = {k: v / args.steps for k, v in fam.items()}
        phases["accumulate_total"] = d("ms_total") / args.steps
        dtype = "u32/u64 packed k-mer keys, u32 LDS sums, u64 triangle (read-modify-write by the owner band, atomics beyond)"
    return roofline, phases, dtype


def finish(out, args, N, L, g, m, _native):
    """rank 0 / the single process: extras that do not depend on the job's ranks, then the ONE JSON line."""
    world = out["n_gpus"]
    if world == 1 and not args.no_also:
        out["also"] = other_configs(_native)
    if world == 1 and not args.no_cpu_baseline:
        full_n, full_L = (N, L) if args.config == 5 else (100000, 300)
        bg, bm = (g, m) if args.config == 5 else (12, 8)
        out["cpu_baseline"] = cpu_baseline(bg, bm, full_L, full_n, args.cpu_seconds)
        if args.config != 5:
            out["cpu_baseline"]["sample"] += " [config-5 generator: this run's own workload is --config %d]" % args.config
    # RCCL prints a version banner through C stdio; flush it first so the JSON line is last
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if out.get("bit_identical_to_1gpu") is False:
        sys.exit("k_digest differs from the committed single-GPU digest (or between ranks): the result is NOT the 1-GPU result")


def main_inproc(args):
    """One process, N GPUs: the engine behind FastSK(devices=[0..N-1])."""
    from fastsk_amd import _native
    tokens, offsets, N, L, g, m, workload, data = workload_of(args)
    # (FSK_BENCH_SHARE_GPU=1: a smoke test of the group on a 1-GPU box, every engine on device 0 over the P2P kernels)
    devices = [0] * args.gpus if os.environ.get("FSK_BENCH_SHARE_GPU") == "1" else list(range(args.gpus))
    coll = {"auto": _native.COLL_AUTO, "rccl": _native.COLL_RCCL, "p2p": _native.COLL_P2P}[args.collective]
    exp = expected_step_seconds(args)
    step_bound = args.step_bound if args.step_bound > 0 else max(45.0, 12.0 * exp)
    wd = None if args.no_watchdog else Watchdog(0, args.gpus, args)
    stage = (lambda name, bound: wd.stage(name, bound)) if wd else (lambda name, bound: None)
    stage("fsk_create_multi (ncclCommInitAll)", args.init_timeout + 30.0)
    try:
        # the engine's own deadline (fsk_config.deadline_ms) bounds every host-side wait of the exchange: communicator
        # set-up, the engines' barriers, each band's all-reduce; the watchdog above is the backstop for a thread that
        # never returns from the runtime
        lib = _native.Library(args.engine_lib) if args.engine_lib else None
        eng = _native.Engine(g, m, devices=devices, collective=coll, bands=args.bands or 0, profile=2, lib=lib,
                             deadline_ms=int(1e3 * min(args.init_timeout, step_bound)))
    except _native.FskError as exc:
        fatal(wd, 0, args.gpus, args, "fsk_create_multi failed: %s" % exc, "fsk_create_multi")
    ncomb = eng.lib.num_combos(g, m)
    workload += ", %d combos" % ncomb
    pairs = N * (N + 1) // 2
    plan = memory_plan(N, pairs, args.gpus, dense=args.config == 5, narrow_possible=True)
    try:
        stage("load sequences", 240.0)
        t_load = time.perf_counter()
        eng.load_sequences(tokens, offsets, N, 0)
        eng.synchronize()
        t_load = time.perf_counter() - t_load
        every = np.arange(ncomb, dtype=np.int32)
        dense = eng.stats()["path_used"] == 1

        def step():
            eng.reset_counts()
            eng.accumulate(every)   # combos r, r + R, ... on engine r; the all-reduce band by band behind the kernels
            eng.finalize()

        for i in range(args.warmup):
            stage("warm-up step %d of %d" % (i + 1, args.warmup), step_bound + 120.0)
            step()
        eng.synchronize()
        s0 = eng.stats()
        stage("%d timed steps" % args.steps, args.steps * (step_bound + (0.0 if args.warmup else 120.0)) + 30.0)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        eng.synchronize()
        elapsed = time.perf_counter() - t0
        s1 = eng.stats()
        stage("one step in measurement mode (exact U)", step_bound + 60.0)
        eng.set_tuning("profile", 1)
        p0 = eng.stats()
        step()
        eng.synchronize()
        p1 = eng.stats()
        eng.set_tuning("profile", 2)
        stage("digest + line", step_bound)
    except _native.FskError as exc:
        fatal(wd, 0, args.gpus, args, "the in-process group failed: %s" % exc, wd.name if wd else "step",
              {"memory_plan_GB": {k: round(v / 1e9, 2) for k, v in plan.items()}})
    info = eng.multi_info()
    digest = eng.counts_digest()
    key = digest_key(args.config, N, L, g, m, ncomb)
    want = committed_digest(key)
    # engine 0's share: its stats carry the HIP-event times
    n0 = info["combos_per_engine"][0]
    e0, s0e, q0, q1 = dict(s1), dict(s0), dict(p0), dict(p1)
    # (cell_updates / dense_macs in a group's stats are sums over the engines: bring them back to engine 0's share)
    for k in ("cell_updates", "dense_macs"):
        for dct, src in ((e0, s1), (s0e, s0), (q0, p0), (q1, p1)):
            dct[k] = src[k] * n0 / max(1, ncomb)
    roofline, phases, dtype = roofline_of(args, s0e, e0, dense, args.gpus, n0 * args.steps, 1.0, offsets, n0, prof=(q0, q1))
    out = {
        "metric": "gkm kernel build: mismatch-combos/s", "value": ncomb * args.steps / elapsed, "unit": "combos/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": data,
        "config": {"workload": workload, "n_seq": N, "seq_len": L, "g": g, "m": m, "combos": int(ncomb),
                   "parallelism": "in-process x%d: one engine per GPU behind one handle (fsk_create_multi / FastSK(devices=...)), combos c = "
                                  "engine (mod %d), one logical all-reduce issued in %d row bands on exchange streams under the next band's kernels"
                                  % (args.gpus, args.gpus, info["bands"]),
                   "path": "dense" if dense else "sparse"},
        "roofline": roofline, "load_seconds_untimed": t_load, "phases_ms_per_step": phases,
        "comm": {"backend": "%s called from the host C++ (no torch in this process)" % ("RCCL" if info["collective"] == "rccl" else "engine P2P kernels"),
                 "rccl_ranks": info["comm_ranks"] if info["collective"] == "rccl" else 0, "ranks": info["comm_ranks"],
                 "allreduce_payload_bytes": info["reduce_bytes"], "allreduce_dtype": "int32" if info["narrow"] else "uint64",
                 "bands": info["bands"], "combos_per_engine": info["combos_per_engine"],
                 "latency_bound": not dense},
        "k_digest": dict(digest_hex(digest), key=key, committed=want),
        "bit_identical_to_1gpu": None if want is None else (digest_hex(digest) == {"sum": want["sum"], "xor": want["xor"]}),
    }
    eng.close()
    if wd:
        wd.finish()
    finish(out, args, N, L, g, m, _native)


def main_one_pass(args):
    """What live_traffic()'s children run under rocprofv3: ONE fsk_compute of the workload (load, every combo once,
    finalize) and nothing else, so that the counters of the pass can be summed per kernel name."""
    from fastsk_amd import _native
    tokens, offsets, N, L, g, m, _, _ = workload_of(args)
    e = _native.Engine(g, m)
    e.compute(tokens, offsets, N, 0)
    print(int(e.stats()["combos_done"]))
    e.close()


def main():
    args = parse_args()
    if args.one_pass:
        return main_one_pass(args)
    if args.inproc:
        return main_inproc(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher around us: start one fresh process per GPU and hand back their exit code. Nothing
        # in THIS process has touched the GPU (torch is not even imported yet), and nothing is exec'd.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist
    from fastsk_amd import _native, distributed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # Smoke tests of the multi-rank code on a 1-GPU box, not measurements: FSK_BENCH_FORCE_DIST=1 runs
    # the RCCL leg with world size 1; FSK_BENCH_SHARE_GPU=1 puts every rank on cuda:0 over gloo (RCCL
    # cannot run two ranks on one device).
    share = os.environ.get("FSK_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("FSK_BENCH_FORCE_DIST") == "1"
    # ---- fail fast (first contact with several GPUs): every stage below has a bound; see Watchdog
    wd = None
    if use_dist and not args.no_watchdog:
        wd = Watchdog(rank, world, args)
    stall = os.environ.get("FSK_BENCH_STALL", "").split(":")  # test-only: "rank:stage substring:seconds" — this rank hangs there

    def stage(name, bound):
        if wd:
            wd.stage(name, bound)
        if len(stall) == 3 and stall[0] == str(rank) and stall[1] in name:
            time.sleep(float(stall[2]))

    exp = expected_step_seconds(args)
    # one whole step of one rank, exchange included (1 GPU: `exp` seconds)
    step_bound = args.step_bound if args.step_bound > 0 else max(45.0, 12.0 * exp)
    backend = None
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = "gloo" if share else "nccl"
        stage("init_process_group(%s)" % backend, args.init_timeout + 30.0)
        tmo = datetime.timedelta(seconds=args.init_timeout)
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=tmo)
        stage("first collective (communicator set-up)", args.init_timeout + 30.0)
        probe = torch.ones(1, dtype=torch.int64, device="cpu" if share else "cuda")
        dist.all_reduce(probe)
        if int(probe.item()) != world:
            fatal(wd, rank, world, args, "the first all-reduce over %d ranks returned %d" % (world, int(probe.item())), "first collective")

    stage("workload + memory plan", 180.0)
    tokens, offsets, N, L, g, m, workload, data = workload_of(args)
    pairs = N * (N + 1) // 2
    # ---- per-rank memory plan, checked before anything large is allocated
    free_b, total_b = torch.cuda.mem_get_info()
    plan = memory_plan(N, pairs, world, dense=args.config == 5, narrow_possible=True)
    if wd:
        wd.partial["memory_plan_GB"] = {k: round(v / 1e9, 2) for k, v in plan.items()}
        wd.partial["hbm_free_GB"] = round(free_b / 1e9, 2)
    if free_b < plan["total"] and not share:
        fatal(wd, rank, world, args,
              "rank %d (cuda:%d): %.1f GB of HBM free, the plan needs %.1f GB (%.1f GB integer triangle + %.2f GB exchange staging + "
              "%.1f GB count panels / sort scratch + headroom): free the GPU or lower --n-seq"
              % (rank, local_rank, free_b / 1e9, plan["total"] / 1e9, plan["triangle_u64"] / 1e9, plan["exchange_staging"] / 1e9,
                 plan["count_panels_or_sort_scratch"] / 1e9), "memory plan", {"memory_plan_GB": {k: round(v / 1e9, 2) for k, v in plan.items()}})

    # ---- preflight: the same multi-GPU code on a job that takes a fraction of a second and whose single-GPU digest
    # is committed (config 5 at N = 16000): a broken interconnect or a wrong reduce is reported in seconds
    preflight = None
    if use_dist and world > 1 and args.config == 5 and not args.no_preflight and N >= 16000:
        stage("preflight (N = 16000, one step, digest committed)", max(90.0, step_bound))
        pN = 16000
        ptok, poff, _ = synthetic(pN, args.seq_len)
        pg, pm = args.g or 12, args.m or 8
        peng = _native.Engine(pg, pm, device=local_rank)
        pncomb = peng.lib.num_combos(pg, pm)
        ppairs = pN * (pN + 1) // 2
        pK = torch.zeros(ppairs, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        peng.bind_counts(pK.data_ptr(), ppairs, keepalive=pK)
        peng.load_sequences(ptok, poff, pN, 0)
        t0 = time.perf_counter()
        peng.reset_counts()
        distributed.accumulate_and_reduce(peng, pK, np.arange(rank, pncomb, world, dtype=np.int32), n_combos_total=pncomb, n_bands=args.bands)
        peng.finalize()
        torch.cuda.synchronize()
        pdt = time.perf_counter() - t0
        pd = peng.counts_digest()
        pwant = committed_digest(digest_key(5, pN, args.seq_len, pg, pm, pncomb))
        ok = None if pwant is None else digest_hex(pd) == {"sum": pwant["sum"], "xor": pwant["xor"]}
        flag = torch.tensor([0 if ok is False else 1], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        preflight = {"n_seq": pN, "seconds": pdt, "k_digest": digest_hex(pd), "committed": pwant, "bit_identical_to_1gpu": ok,
                     "every_rank_agrees": bool(int(flag.item()) == 1)}
        if wd:
            wd.partial["preflight"] = preflight
        peng.close()
        del pK
        torch.cuda.empty_cache()
        if int(flag.item()) != 1:
            fatal(wd, rank, world, args, "preflight: the %d-GPU triangle of the N = 16000 job is not the committed single-GPU one "
                  "(rank %d digest %s)" % (world, rank, digest_hex(pd)), "preflight", {"preflight": preflight})

    stage("allocate the triangle + load sequences", 240.0)
    # profile = 2: the product dataflow, with HIP events recorded (never waited for) around every kernel family of the timed steps
    eng = _native.Engine(g, m, device=local_rank, profile=2)
    ncomb = eng.lib.num_combos(g, m)
    workload += ", %d combos" % ncomb
    K = torch.zeros(pairs, dtype=torch.int64, device="cuda")  # the integer triangle RCCL reduces
    torch.cuda.synchronize()
    eng.bind_counts(K.data_ptr(), pairs, keepalive=K)
    t_load = time.perf_counter()
    eng.load_sequences(tokens, offsets, N, 0)  # host packing + H2D: outside the timed region (see end_to_end)
    eng.synchronize()
    t_load = time.perf_counter() - t_load
    # The normalised result of a step (fastsk_kernel.cpp:96-103 over every cell — part of SURVEY 8(d)'s metric): the whole
    # triangle of doubles on the device when it fits beside the integer one, else a 4096 x 4096 block
    nblk = min(4096, N)
    free_now, _ = torch.cuda.mem_get_info()
    whole = free_now > pairs * 8 + (24 << 30 if args.config == 5 and N >= 50000 else 4 << 30)
    tri = torch.empty(pairs, dtype=torch.float64, device="cuda") if whole else None
    every = np.arange(ncomb, dtype=np.int32)
    dense = eng.stats()["path_used"] == 1
    edges = distributed.owner_edges(N, world) if dense else None
    mode = args.shard
    if mode == "rows" and (edges is None or world == 1):
        mode = "combos"  # rows needs the dense dataflow and one 128-row band per rank
    my_rows = (edges[rank], edges[rank + 1]) if mode == "rows" else (0, N)
    mine = every if mode == "rows" else np.arange(rank, ncomb, world, dtype=np.int32)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step(how=mode):
        if use_dist and how == "rows" and not args.replicate:
            eng.reset_counts_rows(*edges[rank:rank + 2])  # a rank only ever writes (and zeroes) the rows it owns
        else:
            eng.reset_counts()
        if use_dist and how == "rows":
            # every combo over this rank's rows; only the diagonal is exchanged
            distributed.accumulate_owned_rows(eng, K, every, replicate=args.replicate, n_sub=args.bands, edges=edges)
        elif use_dist:
            # combos sharded; the all-reduce of the partial triangles runs band by band on RCCL's
            # stream under the next band's kernels
            distributed.accumulate_and_reduce(eng, K, np.arange(rank, ncomb, world, dtype=np.int32), n_combos_total=ncomb,
                                              n_bands=args.bands, force=world == 1)
        else:
            eng.accumulate(mine)
        eng.finalize()
        if how == "rows" and use_dist and not args.replicate:
            return  # (the kernel matrix stays distributed: its blocks are normalised where they are served)
        if whole:
            eng.get_triangle_torch(tri)   # fsk_get_triangle_device: returns when the triangle is written
        else:
            eng.get_block_torch(0, nblk, 0, nblk)

    def gather_digest(how):
        """The digest of the job's triangle after a step of decomposition `how`, and whether every rank agrees:
        combos (and replicated rows): every rank holds all of K and folds all of it; rows: a rank folds the
        rows it owns and the words are combined (sum: +, xor: ^)."""
        owned = use_dist and how == "rows" and not args.replicate
        d = eng.counts_digest(*edges[rank:rank + 2]) if owned else eng.counts_digest()
        if not use_dist:
            return d, True
        words = np.array(d, dtype=np.uint64).view(np.int64)
        t = torch.from_numpy(words.copy()).to("cuda" if backend == "nccl" else "cpu")
        allw = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allw, t)
        allw = [w.cpu().numpy().view(np.uint64) for w in allw]
        if owned:
            total, x = 0, 0
            for w in allw:
                total = (total + int(w[0])) % (1 << 64)
                x ^= int(w[1])
            return (total, x), True
        return (int(allw[0][0]), int(allw[0][1])), all(int(w[0]) == int(allw[0][0]) and int(w[1]) == int(allw[0][1]) for w in allw)

    def timed(how, warmup, steps):
        eng.reset_counts()  # whatever the other decomposition left in K (untimed)
        bound = step_bound
        for i in range(warmup):
            stage("%s: warm-up step %d of %d" % (how, i + 1, warmup), bound + (120.0 if i == 0 else 0.0))  # (first: allocations, RCCL channels)
            tw = time.perf_counter()
            step(how)
            torch.cuda.synchronize()
            bound = max(30.0, 10.0 * (time.perf_counter() - tw)) if i > 0 or warmup == 1 else bound  # this run's own step time
        stage("%s: barrier before the timed steps" % how, step_bound)
        barrier()
        a = eng.stats()
        stage("%s: %d timed steps" % (how, steps), steps * (bound if warmup else step_bound + 120.0) + 30.0)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(how)
        barrier()
        dt = time.perf_counter() - t0
        if wd:
            wd.partial["%s_ms_per_step_this_rank" % how] = round(1e3 * dt / max(1, steps), 3)
        stage("%s: max over ranks + digest" % how, step_bound)
        b = eng.stats()
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, a, b, gather_digest(how)

    key = digest_key(args.config, N, L, g, m, ncomb)
    want = committed_digest(key)

    def verdict(dg, agree):
        """(the `k_digest` object, bit_identical_to_1gpu): False when the ranks disagree or the digest is not the
        committed single-GPU one; None when no single-GPU digest of this workload is committed."""
        obj = dict(digest_hex(dg), key=key, committed=want, ranks_agree=agree)
        if not agree:
            return obj, False
        if want is None:
            return obj, None
        return obj, digest_hex(dg) == {"sum": want["sum"], "xor": want["xor"]}

    elapsed, s0, s1, (dg, agree) = timed(mode, args.warmup, args.steps)
    k_digest, identical = verdict(dg, agree)
    # ---- one more step in measurement mode (profile = 1, outside the timed region): the exact update count U and the
    # count-MACs including the flagged rows' remainder products — what the product dataflow does not compute
    stage("one step in measurement mode (exact U)", step_bound + 60.0)
    eng.set_tuning("profile", 1)
    p0 = eng.stats()
    step(mode)
    torch.cuda.synchronize()
    p1 = eng.stats()
    eng.set_tuning("profile", 2)
    alt = None
    if world > 1 and not args.no_alt and edges is not None:
        other = "combos" if mode == "rows" else "rows"
        dt, _, _, (adg, aagree) = timed(other, args.warmup, args.steps)
        akd, aid = verdict(adg, aagree)
        alt = {"parallelism": describe(other, world, args.replicate), "value": ncomb * args.steps / dt, "unit": "combos/s",
               "ms_per_step": 1e3 * dt / args.steps, "steps": args.steps, "warmup": args.warmup, "k_digest": akd,
               "bit_identical_to_1gpu": aid}
        if aid is False:
            identical = False

    # ---- the exchange by itself: the same band-wise all-reduce, nothing overlapping it (untimed extras;
    # K is garbage afterwards and is reset by whatever runs next)
    comm = None
    stage("the exchange by itself", step_bound)
    if use_dist:
        narrow = ncomb * eng.stats()["max_windows"] ** 2 < 2 ** 31
        tiles = ((N + 127) // 128) * ((N + 127) // 128 + 1) // 2
        nb = args.bands or ((16 if tiles >= 16 * 16384 else 8) if (dense and N >= 8192) else 1)
        be = distributed.band_edges(N, nb)
        segs = [K[distributed.cell(lo):distributed.cell(hi)] for lo, hi in zip(be[:-1], be[1:])]
        barrier()
        t0 = time.perf_counter()
        for seg in segs:
            buf = seg.to(torch.int32) if narrow else seg
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            if narrow:
                seg.copy_(buf)
        barrier()
        t_ar = time.perf_counter() - t0
        payload = pairs * (4 if narrow else 8)
        comm = {"backend": backend + (" (RCCL)" if backend == "nccl" else " (smoke test, not RCCL)"),
                "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0,
                "allreduce_ms_per_step_not_overlapped": 1e3 * t_ar, "allreduce_payload_bytes": payload,
                "allreduce_dtype": "int32" if narrow else "int64", "bands": len(segs),
                "algbw_GBs": payload / t_ar / 1e9,
                "busbw_GBs": payload / t_ar / 1e9 * 2 * (world - 1) / max(1, world),
                # a small triangle (the sparse workloads: 26 MB at config 4) is exchanged in one piece after a few
                # milliseconds of kernels: its latency, not the links' bandwidth, is what a step pays
                "latency_bound": bool(not dense or payload < (64 << 20))}

    # ---- SURVEY 8(d)'s metric boundary, literally: H2D of the packed sequences + every combo + normalise, the
    # whole triangle (fastsk_kernel.cpp:96-103 over all N(N+1)/2 cells, device resident) when a second
    # triangle of doubles fits beside the integer one, else a 4096 x 4096 block (single GPU only)
    end_to_end = None
    stage("end_to_end", 4 * step_bound)
    if world == 1 and not use_dist:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.load_sequences(tokens, offsets, N, 0)
        eng.accumulate(every)
        eng.finalize()
        t1 = time.perf_counter()
        if whole:
            eng.get_triangle_torch(tri)   # fsk_get_triangle_device: returns when the triangle is written
        else:
            blk = eng.get_block_torch(0, nblk, 0, nblk)
        torch.cuda.synchronize()
        t_e2e = time.perf_counter() - t0
        t_norm = time.perf_counter() - t1
        if whole:
            rr = torch.arange(N, device="cuda", dtype=torch.int64)
            assert bool((tri[rr * (rr + 1) // 2 + rr] == 1.0).all()), "a normalised diagonal is not 1.0"
            blk = eng.get_block_torch(0, nblk, 0, nblk)   # and a block of it against the block getter
            ii = torch.arange(nblk, device="cuda", dtype=torch.int64)
            low = torch.tril(torch.ones(nblk, nblk, dtype=torch.bool, device="cuda"))
            cells = (ii[:, None] * (ii[:, None] + 1) // 2 + ii[None, :])[low]
            assert bool(torch.equal(tri[cells], blk[low])), "normalised triangle and block getter disagree"
        else:
            assert bool((blk.diagonal() == 1.0).all())
        end_to_end = {"seconds": t_e2e, "combos_per_s": ncomb / t_e2e, "normalise_ms": 1e3 * t_norm,
                      "normalise_cells": pairs if whole else nblk * nblk,
                      "normalise_GBs": (16.0 * pairs / t_norm / 1e9) if whole else None,
                      "includes": "fsk_load_sequences (alphabet scan, bit-packing, H2D of the packed sequences, zeroing K) + "
                                  "all %d combos + fsk_finalize + %s" % (
                                      ncomb, "the WHOLE normalised triangle, %d cells of float64 written on the device "
                                      "(fsk_get_triangle_device; 16 bytes of HBM per cell)" % pairs if whole
                                      else "a normalised %d x %d train block on the device (no room for a second triangle)" % (nblk, nblk)),
                      "value_excludes": "`value` is timed with the packed sequences resident in HBM (the bench contract: host "
                                        "buffers and PCIe are never part of `value`): it excludes fsk_load_sequences (pack + H2D, "
                                        "load_seconds_untimed) and nothing else; end_to_end.combos_per_s is the same step with it"}
        torch.cuda.empty_cache()

    out = None
    if rank == 0:
        share_rows = ((my_rows[1] * (my_rows[1] + 1) - my_rows[0] * (my_rows[0] + 1)) // 2) / pairs
        roofline, phases, dtype = roofline_of(args, s0, s1, dense, world, len(mine) * args.steps, share_rows, offsets, len(mine), prof=(p0, p1))
        out = {
            "metric": "gkm kernel build: mismatch-combos/s", "value": ncomb * args.steps / elapsed, "unit": "combos/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": dtype, "data": data,
            "config": {"workload": workload, "n_seq": N, "seq_len": L, "g": g, "m": m, "combos": int(ncomb),
                       "parallelism": describe(mode, world, args.replicate),
                       "path": "dense" if dense else "sparse"},
            "value_boundary": "a step = fsk_reset_counts + every combo (extract -> sort/group -> count -> K +=)%s + fsk_finalize + the "
                              "normalised result written on the device (%s), packed sequences resident in HBM when it starts; "
                              "the engine runs its product dataflow (fsk_config.profile = 2: HIP events recorded, never waited for)"
                              % (" + the all-reduce of the partial triangles" if use_dist else "",
                                 "the WHOLE triangle, %d float64 cells" % pairs if whole else "a %d x %d block: no room for a second triangle" % (nblk, nblk)),
            "roofline": roofline,
            "load_seconds_untimed": t_load,
            "phases_ms_per_step": phases,
            "k_digest": k_digest,
            "bit_identical_to_1gpu": identical,
        }
        if alt is not None:
            out["alt"] = alt
        if comm is not None:
            out["comm"] = comm
        if preflight is not None:
            out["preflight"] = preflight
        if use_dist:
            out["memory_plan_GB"] = {k: round(v / 1e9, 2) for k, v in plan.items()}
            out["fail_fast"] = {"watchdog": wd is not None, "init_timeout_s": args.init_timeout, "step_bound_s": step_bound,
                                "note": "every stage of a multi-GPU run is bounded; an overrun prints one JSON error line on rank 0 and exits 2"}
        if end_to_end is not None:
            out["end_to_end"] = end_to_end
            # SURVEY 8(d)'s boundary (host buffers in: pack + H2D once + every combo + finalize + the whole triangle) beside the
            # contract's (`value`: inputs resident in HBM when the clock starts — host buffers and PCIe are never part of it)
            out["end_to_end_combos_per_s"] = end_to_end["combos_per_s"]
            out["resident_combos_per_s"] = out["value"]
    stage("closing barrier", step_bound)
    if use_dist:
        dist.barrier()
    eng.close()
    del K, tri
    torch.cuda.empty_cache()
    # ---- the same job from ONE process (FastSK(devices=[...])'s engine), measured once the ranks have let go of
    # their GPUs: rank 0 starts a fresh child (it has touched the GPU itself: a child process, never an exec)
    # and the others wait on the rendezvous store, not on a collective that would spin on their GPUs.
    if world > 1 and backend == "nccl" and not args.no_inproc_leg:
        import datetime
        stage("in-process leg (child of rank 0)", args.inproc_timeout + 180.0)
        if wd and out is not None:
            wd.partial["line_so_far"] = {k: out[k] for k in ("value", "ms_per_step", "bit_identical_to_1gpu", "k_digest") if k in out}
        store = dist.distributed_c10d._get_default_store()
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            leg = {"cmd": "bench.py --gpus %d --inproc" % world}
            try:
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--inproc", "--steps", str(args.steps), "--warmup",
                       str(args.warmup), "--config", str(args.config), "--n-seq", str(args.n_seq), "--seq-len", str(args.seq_len),
                       "--no-also", "--no-cpu-baseline"] + (["--bands", str(args.bands)] if args.bands else [])
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                         "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.inproc_timeout, env=env)
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if lines:
                    leg.update(json.loads(lines[-1]))
                    for k in ("roofline", "phases_ms_per_step", "dtype", "data", "metric", "unit", "higher_is_better", "vs_baseline"):
                        leg.pop(k, None)
                if r.returncode != 0 or not lines:
                    leg["error"] = "exit code %d: %s" % (r.returncode, (r.stderr or r.stdout)[-600:])
                if leg.get("bit_identical_to_1gpu") is False:
                    out["bit_identical_to_1gpu"] = False
            except subprocess.TimeoutExpired:
                leg["error"] = "no result within %.0f s (child stopped)" % args.inproc_timeout
            except Exception as exc:
                leg["error"] = repr(exc)
            out["inproc"] = leg
            store.set("fsk_inproc_leg_done", "1")
        else:
            store.wait(["fsk_inproc_leg_done"], datetime.timedelta(seconds=args.inproc_timeout + 120))
    stage("destroy_process_group", 120.0)
    if wd and out is not None:
        wd.partial["line_so_far"] = {k: out[k] for k in ("value", "ms_per_step", "bit_identical_to_1gpu", "k_digest") if k in out}
    if use_dist:
        dist.destroy_process_group()
    if wd:
        wd.finish()
    if rank == 0:
        finish(out, args, N, L, g, m, _native)
    elif identical is False:
        sys.exit(1)


if __name__ == "__main__":
    main()

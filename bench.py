#!/usr/bin/env python3
"""bench.py — headline benchmark of the gapped-k-mer kernel build (BASELINE.json metric).

A "step" is one pass of the hot path over the whole workload: BASELINE config 5, synthetic
100,000 x 300 bp DNA, g=12, m=8, exact, all C(12,8)=495 mismatch combinations, with the packed
sequences already resident in HBM when the timed region starts. With N GPUs the job is the same
(STRONG scaling: total work fixed) and is sharded one of two ways (fastsk_amd/distributed.py):
  rows    (default at this size) every rank owns an equal-area band of rows of the triangle and runs
          all 495 combos over it; no cell is shared, the only exchange is the 0.8 MB diagonal, and
          the kernel matrix stays distributed over the GPUs;
  combos  the reference's decomposition: combos c = rank (mod N), every rank accumulates a private
          triangle, RCCL all-reduces it over xGMI in row bands under the next band's kernels.
The other of the two is timed for one step afterwards and reported as "alt".
value = combos/s of the whole job = 495 * steps / max-over-ranks seconds.

One JSON line on rank 0. Extra objects:
  roofline      the dominant kernel (k_dense_tile_dma) priced on SURVEY 8d's algorithmic bytes
                (16*U + sort + input bytes per combo) against the 8 TB/s HBM peak, timed with HIP
                events on the engine's own stream; plus the integer-VALU view of the same launch.
  cpu_baseline  the compiled reference (oracle/_ref, "reference") or our C restatement ("port")
                on the host cores, on a bounded sample (smaller N, fewer combos), rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the host driver only supports dmabuf IPC: RCCL between processes needs this before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
# v_dot8_u32_u4 / v_dot4_u32_u8 issue at HALF the v_fma_f32 rate on gfx950 (measured:
# profiles/r01_ubench_valu_rates.txt): 64 lanes/clk/CU. Peak = CUs * 64 lanes * 8 MACs * 2.4 GHz.
VALU_DOT8_PEAK_TMACS = 256 * 64 * 8 * 2.4e9 / 1e12  # = 314.6 T MAC/s


def synthetic(N, L, seed=20201214):
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.integers(1, 5, size=(N, L), dtype=np.int32)
    return X.reshape(-1), np.arange(N + 1, dtype=np.int64) * L, X


def cpu_baseline(X, g, m, n_sample, budget_s):
    """Reference (or port) on the host cores over a bounded sample of the same workload."""
    from oracle import loader
    cores = os.cpu_count() or 1
    Xs = np.ascontiguousarray(X[:n_sample])
    tokens = Xs.reshape(-1).astype(np.int32)
    offsets = np.arange(n_sample + 1, dtype=np.int64) * X.shape[1]
    kind = "reference" if loader.have_ref() else "port"
    run = (lambda c: loader.ref().raw_counts(tokens, offsets, g, m, c, threads=cores, want_counts=False)[1]) \
        if kind == "reference" else \
        (lambda c: loader.port().raw_counts(tokens, offsets, g, m, c, threads=cores, want_counts=False)[1])
    ncomb = int(loader.port().num_combos(g, m))
    probe = np.arange(min(cores, ncomb), dtype=np.int32)
    t_probe = run(probe)
    per_round = max(t_probe, 1e-3)  # `cores` combos in parallel
    rounds = int(max(1, min((budget_s - t_probe) / per_round, (ncomb - len(probe)) // max(1, cores))))
    combos = np.arange(len(probe), len(probe) + rounds * cores, dtype=np.int32) % ncomb
    t_main = run(combos)
    measured = len(combos) / t_main
    return kind, cores, measured, len(combos), t_main


def other_configs(_native):
    """The other BASELINE configs on the same GPU: the whole fsk_compute call, host buffers in, result
    resident on the device (best of 4). Config 2 = BASELINE configs[1] (EP300 DNA, 2000+2000 x 100 bp,
    g=10 m=6 exact); configs 1, 3, 4 from the golden descriptors (same modes and combo orders as the
    parity tests)."""
    out = {}
    path = os.path.join(ROOT, "tests", "golden", "tokens_EP300.npz")
    if os.path.exists(path):
        z = np.load(path)
        tokens, offsets = z["tokens"].astype(np.int32), z["offsets"].astype(np.int64)
        ntr, nte = int(z["n_train"]), int(z["n_test"])
        e = _native.Engine(10, 6)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            e.compute(tokens, offsets, ntr, nte)
            best = min(best, time.perf_counter() - t0)
        e.close()
        out["config2_ep300_exact"] = {"n_seq": ntr + nte, "seq_len": 100, "g": 10, "m": 6, "combos": 210, "seconds": best,
                                      "combos_per_s": 210 / best, "reference_cpu_seconds_8_threads": 29.9}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import load_golden, load_tokens, GOLD
        for key, name in (("config1_prot11_approx_t1", "f7_cfg1_prot11_approx_t1"), ("config3_ep47848_100combos", "f7_cfg3_ep47848_100combos"),
                          ("config4_prot219_exact", "f7_cfg4_prot219_exact")):
            if not os.path.exists(os.path.join(GOLD, name + ".npz")):
                continue
            d = load_golden(name)
            tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
            e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"],
                               skip_variance=bool(d["skip_variance"]))
            if d["approx"]:
                e.set_combo_order(d["order"])
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter()
                e.compute(tokens, offsets, ntr, nte)
                best = min(best, time.perf_counter() - t0)
            done = int(e.stats()["combos_done"])
            e.close()
            out[key] = {"n_seq": ntr + nte, "g": int(d["g"]), "m": int(d["m"]), "combos": done, "seconds": best,
                        "combos_per_s": done / best, "reference_cpu_seconds": float(d["ref_seconds"])}
    except Exception as exc:  # the headline line must not depend on the extras
        out["error"] = repr(exc)
    return out or None


def describe(mode, world, replicate):
    if world == 1:
        return "single GPU"
    if mode == "rows":
        return ("row-band sharded x%d: every rank runs all combos over its own equal-area band of rows of the triangle; "
                "exchange = the N-entry diagonal only%s" % (world, "; finished bands broadcast to every rank" if replicate
                                                            else "; the kernel matrix stays distributed"))
    return "combo-sharded x%d + RCCL all-reduce of the triangle in row bands under the next band's kernels" % world


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-seq", type=int, default=100000)
    ap.add_argument("--seq-len", type=int, default=300)
    ap.add_argument("-g", type=int, default=12)
    ap.add_argument("-m", type=int, default=8)
    ap.add_argument("--cpu-sample", type=int, default=2000, help="sequences in the CPU baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra config-2 measurement (profiling runs)")
    ap.add_argument("--bands", type=int, default=None, help="row bands of the overlapped all-reduce (default: auto)")
    ap.add_argument("--shard", choices=["auto", "rows", "combos"], default="auto", help="multi-GPU decomposition")
    ap.add_argument("--replicate", action="store_true", help="rows: broadcast finished bands so every rank holds all of K")
    ap.add_argument("--no-alt", action="store_true", help="multi-GPU: do not time the other decomposition")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fastsk_amd import _native, distributed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # Smoke tests of the multi-rank code on a 1-GPU box, not measurements: FSK_BENCH_FORCE_DIST=1 runs
    # the RCCL leg with world size 1; FSK_BENCH_SHARE_GPU=1 puts every rank on cuda:0 over gloo (RCCL
    # cannot run two ranks on one device).
    share = os.environ.get("FSK_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("FSK_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    N, L, g, m = args.n_seq, args.seq_len, args.g, args.m
    tokens, offsets, X = synthetic(N, L)
    eng = _native.Engine(g, m, device=local_rank, profile=True)
    ncomb = eng.lib.num_combos(g, m)
    pairs = N * (N + 1) // 2
    K = torch.zeros(pairs, dtype=torch.int64, device="cuda")  # the integer triangle RCCL reduces
    torch.cuda.synchronize()
    eng.bind_counts(K.data_ptr(), pairs, keepalive=K)
    t_load = time.perf_counter()
    eng.load_sequences(tokens, offsets, N, 0)  # host packing + H2D: outside the timed region
    eng.synchronize()
    t_load = time.perf_counter() - t_load
    every = np.arange(ncomb, dtype=np.int32)
    edges = distributed.owner_edges(N, world)
    dense = eng.stats()["path_used"] == 1
    mode = args.shard
    if mode == "auto":
        mode = "rows" if (world > 1 and dense and edges is not None) else "combos"
    if mode == "rows" and edges is None:
        raise SystemExit("--shard rows needs one 128-row band per rank")
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
            eng.synchronize()
        eng.finalize()

    def timed(how, warmup, steps):
        eng.reset_counts()  # whatever the other decomposition left in K (untimed)
        for _ in range(warmup):
            step(how)
        barrier()
        a = eng.stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(how)
        barrier()
        dt = time.perf_counter() - t0
        b = eng.stats()
        if use_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, a, b

    elapsed, s0, s1 = timed(mode, args.warmup, args.steps)
    alt = None
    if world > 1 and not args.no_alt and (mode == "combos" and dense and edges is not None or mode == "rows"):
        other = "combos" if mode == "rows" else "rows"
        dt, _, _ = timed(other, 1, 1)
        alt = {"parallelism": describe(other, world, args.replicate), "value": ncomb / dt, "unit": "combos/s",
               "ms_per_step": 1e3 * dt, "steps": 1, "warmup": 1}

    if rank == 0:
        combos_rank = len(mine) * args.steps
        d = lambda k: s1[k] - s0[k]
        value = ncomb * args.steps / elapsed
        # ---- roofline of the dominant kernel (tile accumulate), per launch
        launches = max(1, d("n_tile_launches"))
        tile_ms = d("ms_tile") / launches
        # exact, from the count panels (whole triangle); a row-band launch owns its share of the cells
        share = ((my_rows[1] * (my_rows[1] + 1) - my_rows[0] * (my_rows[0] + 1)) // 2) / pairs
        U = d("cell_updates") / launches * share
        nfeat = s1["n_feat"]
        combos_per_launch = combos_rank / launches
        b_in = (N * L * s1["bits_per_symbol"] + 7) // 8
        P = 1  # ceil(k*b/8) 8-bit passes for the packed k-mer (k=4, b=2)
        alg_bytes = 16.0 * U + combos_per_launch * (b_in + 16.0 * P * nfeat)
        achieved = alg_bytes / (tile_ms * 1e-3) / 1e9 if tile_ms > 0 else 0.0
        macs = d("dense_macs") / launches
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if world == 1 and os.path.exists(tpath):  # (measured on the single-GPU launch: does not describe a rank's band)
            try:
                tj = json.load(open(tpath))
                if tj.get("n_seq") == N and tj.get("combos_per_launch") == int(combos_per_launch):
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "gkm kernel build: mismatch-combos/s", "value": value, "unit": "combos/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u4 count planes (v_dot8_u32_u4), u32 accumulate, u64 atomics", "data": "synthetic",
            "config": {"workload": "config5: synthetic DNA %d x %d bp, g=%d m=%d exact, %d combos" % (N, L, g, m, ncomb),
                       "n_seq": N, "seq_len": L, "g": g, "m": m, "combos": int(ncomb),
                       "parallelism": describe(mode, world, args.replicate),
                       "path": "dense" if s1["path_used"] == 1 else "sparse"},
            "roofline": {"bound": "hbm", "kernel": "k_dense_tile_dma" if os.environ.get("FSK_TILE_DMA", "1") != "0" else "k_dense_tile", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": tile_ms, "launches": int(launches), "combos_per_launch": combos_per_launch,
                         "cell_updates_per_launch": U, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "algorithmic bytes = 16*U + sort + input (direct-atomic dataflow, SURVEY 8d); "
                                 "frac > 1 means the tile kernel sums on chip what that dataflow would do in HBM",
                         "valu": {"achieved": macs / (tile_ms * 1e-3) / 1e12 if tile_ms > 0 else 0.0,
                                  "peak": VALU_DOT8_PEAK_TMACS, "unit": "T count-MAC/s (v_dot8_u32_u4 at 64 lanes/clk/CU, 2.4 GHz)",
                                  "frac": (macs / (tile_ms * 1e-3) / 1e12) / VALU_DOT8_PEAK_TMACS if tile_ms > 0 else 0.0}},
            "load_seconds_untimed": t_load,
            "phases_ms_per_step": {"count": d("ms_count") / args.steps, "tile": d("ms_tile") / args.steps,
                                   "accumulate_total": d("ms_total") / args.steps},
        }
        if alt is not None:
            out["alt"] = alt
        if world == 1 and not args.no_also:
            out["also"] = other_configs(_native)
        if world == 1 and not args.no_cpu_baseline:
            ns = min(args.cpu_sample, N)
            kind, cores, measured, nc, secs = cpu_baseline(X, g, m, ns, args.cpu_seconds)
            scale = (ns / N) ** 2
            cpu_model = ""
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            out["cpu_baseline"] = {
                "value": measured * scale, "unit": "combos/s", "cores": cores, "kind": kind, "cpu_model": cpu_model,
                "measured_at_sample": measured,
                "sample": "first %d of %d sequences, %d of %d combos, %.1f s on %d threads: %.3f combos/s at N=%d; "
                          "value = that x (%d/%d)^2 (count time scales as N^2; the reference itself cannot index N > 46340)"
                          % (ns, N, nc, ncomb, secs, cores, measured, ns, ns, N)}
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so the JSON line is last
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

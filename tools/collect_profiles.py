#!/usr/bin/env python3
"""Turn gpurun_out/prof/ (written by tools/refresh_profiles.sh on the GPU box) into the tracked
files under profiles/: kernel stats CSV, the bench JSON lines, the per-dispatch HBM counters and
traffic.json (what bench.py reports as roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"


def newest(pattern):
    """gpurun merges each call's output into the local gpurun_out/: keep the latest run's file."""
    return max(glob.glob(pattern), key=os.path.getmtime)


def json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return line
    raise SystemExit("no JSON line in " + path)


def main():
    # (bench_final.json: the same command run again on the box after traffic.json was rebuilt there)
    final = os.path.join(SRC, "bench_final.json")
    have_final = os.path.exists(final) and any(l.startswith("{") for l in open(final))
    open(os.path.join(DST, TAG + "_bench100k.json"), "w").write(json_line(final if have_final else os.path.join(SRC, "bench.json")))
    open(os.path.join(DST, TAG + "_bench100k_under_rocprof.json"), "w").write(json_line(os.path.join(SRC, "bench_under_rocprof.json")))
    shutil.copy(newest(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv")), os.path.join(DST, TAG + "_kernel_stats_bench100k.csv"))
    rows = []
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(SRC, "pmc_" + counter)
        cc = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        kt = newest(os.path.join(d, "*", "*_kernel_trace.csv"))
        dur = {}
        for r in csv.DictReader(open(kt)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] != counter or "fsk::" not in r["Kernel_Name"]:
                continue
            rows.append({"counter": counter, "kernel": r["Kernel_Name"].split("(")[0], "value_KiB": float(r["Counter_Value"]),
                         "duration_ms": dur.get(r["Dispatch_Id"]), "grid": int(r["Grid_Size"]), "vgpr": int(r["VGPR_Count"]),
                         "lds": int(r["LDS_Block_Size"])})
    out = {"command": "cd /tmp; rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python3 bench.py "
                      "--steps 1 --warmup 0 --no-cpu-baseline --no-also (one pass per counter; tools/refresh_profiles.sh)",
           "note": "values are KiB; on gfx950 FETCH_SIZE reads 1/2 of 16-B/lane streaming reads (MI355X_MICROARCH.md HBM "
                   "section): double it before comparing with bytes", "rows": rows}
    json.dump(out, open(os.path.join(DST, TAG + "_pmc_bench100k.json"), "w"), indent=1)
    tile = lambda c: max(r["value_KiB"] for r in rows if r["counter"] == c and "k_dense_tile" in r["kernel"])
    bench = json.loads(json_line(os.path.join(SRC, "bench.json")))
    fetch, write = tile("FETCH_SIZE") * 1024 * 2, tile("WRITE_SIZE") * 1024
    sys.path.insert(0, ROOT)
    import subprocess
    from bench import kernel_hashes
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "fastsk_amd/csrc"], capture_output=True, text=True).stdout.strip()
    json.dump({"n_seq": bench["config"]["n_seq"], "combos_per_launch": int(bench["roofline"]["combos_per_launch"]),
               "kernel": "fsk::k_dense_tile_dma", "fetch_bytes_corrected": fetch, "write_bytes": write,
               "hbm_bytes_per_launch": fetch + write,
               # bench.py refuses these numbers once the kernel sources they were measured on change
               "kernel_files": kernel_hashes(), "commit": commit + ("+uncommitted csrc changes" if dirty else ""),
               "source": "profiles/%s_pmc_bench100k.json (the 495-combo launch): FETCH_SIZE KiB x1024 x2 (gfx950 correction) + "
                         "WRITE_SIZE KiB x1024" % TAG}, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
    for src, dst in (("configs.jsonl", TAG + "_configs1-4_gpu_timings.jsonl"), ("large_g.jsonl", TAG + "_large_g_regime.jsonl"),
                     ("n_series.jsonl", TAG + "_n_series.jsonl"), ("bench_config4.json", TAG + "_bench_config4.json"),
                     ("bench_rccl_world1.json", TAG + "_bench_rccl_world1.json"), ("bench_inproc1.json", TAG + "_bench_inproc_1gpu.json"),
                     ("bench_inproc3_shared.json", TAG + "_bench_inproc_3engines_one_gpu_p2p.json"),
                     ("bench_inproc1_config4.json", TAG + "_bench_inproc_1gpu_config4.json")):
        if not os.path.exists(os.path.join(SRC, src)):
            continue
        lines = [l for l in open(os.path.join(SRC, src)) if l.startswith("{")]
        if lines:
            open(os.path.join(DST, dst), "w").writelines(lines)
    try:
        shutil.copy(newest(os.path.join(SRC, "stats_configs", "*", "*_kernel_stats.csv")), os.path.join(DST, TAG + "_kernel_stats_configs1-4.csv"))
    except ValueError:
        pass
    for tag, dst in ((TAG + "_cfg4", TAG + "_pmc_sparse_config4.json"), (TAG + "_cfg1", TAG + "_pmc_sparse_config1.json")):
        src = os.path.join(ROOT, "gpurun_out", "pmc", tag, "summary.json")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(DST, dst))
    # measured HBM bytes of one config-4 pass (what `bench.py --config 4` reports as roofline.traffic)
    c4 = os.path.join(DST, TAG + "_pmc_sparse_config4.json")
    if os.path.exists(c4):
        from bench import SPARSE_FILES
        ks = json.load(open(c4))["kernels"]
        total = sum(v.get("hbm_bytes") or 0.0 for v in ks.values())
        json.dump({"workload": "config4: protein 2.19, g=14 m=10, exact", "combos": 1001, "hbm_bytes_per_step": total,
                   "kernel_files": kernel_hashes(SPARSE_FILES), "commit": commit + ("+uncommitted csrc changes" if dirty else ""),
                   "source": "profiles/%s_pmc_sparse_config4.json: sum over the pipeline's kernels of FETCH_SIZE KiB x1024 x2 (gfx950 "
                             "correction) + WRITE_SIZE KiB x1024, one fsk_compute of all 1001 combos" % TAG},
                  open(os.path.join(DST, "traffic_config4.json"), "w"), indent=1)
    for src, dst in (("sparse_large_n.jsonl", TAG + "_sparse_large_n.jsonl"), ("sparse_mid_n.jsonl", TAG + "_sparse_mid_n_bands_vs_blocks.jsonl"),
                     ("sparse_large_n_no_desc.jsonl", TAG + "_sparse_large_n_word_streams_only.jsonl"), ("sparse_mid_n_dna_k8.jsonl", TAG + "_sparse_mid_n_dna_k8_bands_vs_blocks.jsonl"),
                     ("desc_ab.txt", TAG + "_descriptors_ab.txt"), ("kernel_times_large_g.txt", TAG + "_kernel_times_large_g.txt"),
                     ("dropin_wall.json", TAG + "_dropin_wall.json"), ("ubench_sparse_ops.txt", TAG + "_ubench_sparse_ops.txt"),
                     ("ubench_rmw.txt", TAG + "_ubench_rmw_shapes.txt")):
        if os.path.exists(os.path.join(SRC, src)) and os.path.getsize(os.path.join(SRC, src)) > 0:
            shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
    lg = os.path.join(ROOT, "gpurun_out", "pmc", TAG + "_large_g16", "summary.json")
    if os.path.exists(lg):
        shutil.copy(lg, os.path.join(DST, TAG + "_pmc_sparse_large_g16.json"))
    b64 = os.path.join(ROOT, "gpurun_out", "pmc", TAG + "_blocks64k", "summary.json")
    if os.path.exists(b64):
        shutil.copy(b64, os.path.join(DST, TAG + "_pmc_sparse_blocks_64k.json"))
    for src, dst in (("config1_timeline.txt", TAG + "_config1_timeline.txt"), ("kprof_cfg4/kstats.txt", TAG + "_kernel_times_config4.txt"),
                     ("kprof_cfg1/kstats.txt", TAG + "_kernel_times_config1.txt"), ("ep300_approx.txt", TAG + "_ep300_approx_ms.txt")):
        if os.path.exists(os.path.join(SRC, src)):
            shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
    if os.path.exists(os.path.join(SRC, "k_digests.json")):
        shutil.copy(os.path.join(SRC, "k_digests.json"), os.path.join(DST, "k_digests.json"))
    if os.path.exists(os.path.join(SRC, "ubench_mfma_i8.txt")):
        shutil.copy(os.path.join(SRC, "ubench_mfma_i8.txt"), os.path.join(DST, TAG + "_ubench_mfma_i8_vs_dot8.txt"))
    print("profiles/ refreshed from", SRC)


if __name__ == "__main__":
    main()

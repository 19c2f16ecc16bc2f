#!/usr/bin/env python3
"""gpurun_out/pmc/<tag>/ (tools/pmc_passes.sh) -> gpurun_out/pmc/<tag>/summary.json: per kernel, the sum
over its dispatches of every collected counter, the dispatch count and the total duration. FETCH_SIZE /
WRITE_SIZE stay in KiB as rocprofv3 reports them; `hbm_bytes` applies the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE x 2 for wide streaming reads — an upper bound for narrow ones)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", "pmc", tag)
    kernels = {}
    first_set = {}  # a counter listed in several passes (SQ_INSTS_VALU rides along as the yardstick) counts from the first one only
    for cc in sorted(glob.glob(os.path.join(base, "set*", "*", "*_counter_collection.csv"))):
        which = os.path.relpath(cc, base).split(os.sep)[0]
        for r in csv.DictReader(open(cc)):
            if first_set.setdefault(r["Counter_Name"], which) != which:
                continue
            k = kernels.setdefault(short(r["Kernel_Name"]), {"counters": {}})
            c = k["counters"]
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            k["vgpr"], k["lds"], k["wg"] = int(r["VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Workgroup_Size"])
    for kt in sorted(glob.glob(os.path.join(base, "stats", "*", "*_kernel_trace.csv"))):
        for r in csv.DictReader(open(kt)):
            k = kernels.setdefault(short(r["Kernel_Name"]), {"counters": {}})
            k["dispatches"] = k.get("dispatches", 0) + 1
            k["total_ms"] = k.get("total_ms", 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k in kernels.values():
        c = k["counters"]
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            k["hbm_bytes"] = c.get("FETCH_SIZE", 0.0) * 1024 * 2 + c.get("WRITE_SIZE", 0.0) * 1024
            if k.get("total_ms"):
                k["hbm_GBs"] = k["hbm_bytes"] / (k["total_ms"] * 1e-3) / 1e9
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
            k["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if c.get("SQ_WAVE_CYCLES"):
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
                if name in c:
                    k[name.lower() + "_frac_of_wave_cycles"] = c[name] / c["SQ_WAVE_CYCLES"]
    for k in kernels.values():
        c = k["counters"]
        # measured price of a VALU instruction: SQ_INST_CYCLES_VALU (cycles the VALU spent on them) per SQ_INSTS_VALU, and the
        # share of the kernel's SIMD time that is (total_ms x 1024 SIMDs x the 2.4 GHz peak clock: a lower bound of the share)
        if c.get("SQ_INST_CYCLES_VALU") and c.get("SQ_INSTS_VALU"):
            k["valu_cycles_per_inst"] = c["SQ_INST_CYCLES_VALU"] / c["SQ_INSTS_VALU"]
            if k.get("total_ms"):
                k["valu_issue_occupancy"] = c["SQ_INST_CYCLES_VALU"] / (k["total_ms"] * 1e-3 * 2.4e9 * 1024)
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_INSTS_VALU"):
            k["active_inst_valu_quadcycles_per_inst"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            k["lds_conflict_frac_of_idx_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    fsk = {n: v for n, v in kernels.items() if "fsk::" in n}
    out = {"tag": tag, "command": "tools/pmc_passes.sh %s ... (one rocprofv3 --pmc pass per counter set + one --kernel-trace --stats pass)" % tag,
           "kernels": dict(sorted(fsk.items(), key=lambda kv: -kv[1].get("total_ms", 0.0)))}
    path = os.path.join(base, "summary.json")
    json.dump(out, open(path, "w"), indent=1)
    for n, v in out["kernels"].items():
        print("%-48s %4d disp %9.3f ms  %s" % (n[:48], v.get("dispatches", 0), v.get("total_ms", 0.0),
                                                 {k: round(x, 3) for k, x in v.items() if isinstance(x, float) and k != "total_ms"}))


if __name__ == "__main__":
    main()

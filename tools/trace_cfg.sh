#!/bin/bash
# tools/trace_cfg.sh <golden name>: kernel timeline (with the gaps) of the last of three fsk_compute calls of a BASELINE config
NAME=${1:-f7_cfg4_prot219_exact}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace4
rm -rf "$O"; mkdir -p "$O"
RUN=$(mktemp /tmp/fsk_trace_XXXXXX.py)
cat > "$RUN" <<PY
import os, sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from conftest import load_golden, load_tokens
from fastsk_amd import _native
d = load_golden("$NAME")
tokens, offsets, ntr, nte, _, _ = load_tokens(d["data"])
e = _native.Engine(d["g"], d["m"], t=d["t"], approx=bool(d["approx"]), delta=d["delta"], max_iters=d["max_iters"], skip_variance=bool(d["skip_variance"]))
if d["approx"]: e.set_combo_order(d["order"])
for _ in range(3):
    t0=time.perf_counter(); e.compute(tokens, offsets, ntr, nte); print("run %.3f ms" % ((time.perf_counter()-t0)*1e3))
    time.sleep(0.01)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/t" -- python3 "$RUN" > "$O/out.txt" 2> "$O/err.txt"
cd "$R" && python3 - <<'PY'
import csv, glob, os
f = max(glob.glob("gpurun_out/trace4/t/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]) for r in csv.DictReader(open(f))]
rows.sort()
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 8):i]) > 1000000: cut = i
run = rows[cut:]
t0 = run[0][0]
print("kernels", len(run), "span %.3f ms" % ((max(r[1] for r in run) - t0) / 1e6))
ce = run[0][1]
for a, b, n in run:
    gap = (a - ce) / 1e3
    if gap > 8: print("   --- gap %.1f us" % gap)
    print("%9.1f %9.1f %7.1f  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, n))
    ce = max(ce, b)
PY

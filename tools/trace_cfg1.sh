#!/bin/bash
# Run on the GPU box from the repo root: kernel timeline of config 1 (variance mode) -> gpurun_out/trace1/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace1
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/t" -- python3 "$R/tools/trace_cfg1.py" 3 > "$O/out.txt" 2> "$O/err.txt"
cd "$R" && python3 - <<'PY'
import csv, glob, os
f = max(glob.glob("gpurun_out/trace1/t/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# last run = after the last gap > 300 us
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 8):i]) > 300000: cut = i
run = rows[cut:]
t0 = run[0][0]
print("kernels", len(run), "span %.3f ms" % ((max(r[1] for r in run) - t0) / 1e6))
for a, b, n, q in run[:400]:
    print("%9.1f %9.1f %7.1f  q%-4s %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, n))
PY

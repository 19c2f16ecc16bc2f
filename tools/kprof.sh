#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: per-kernel times of one BASELINE config under rocprofv3
# --kernel-trace --stats.   tools/kprof.sh <golden name> [tag]
set -u
NAME=${1:-f7_cfg4_prot219_exact}; TAG=${2:-kprof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/tools/profile_one.py" "$NAME" > "$O/out.txt" 2> "$O/err.txt"
cd "$R" && python3 tools/kstats.py "$O/stats" | tee "$O/kstats.txt"

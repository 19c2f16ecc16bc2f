#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: wall of BASELINE configs 1-4 (tools/bench_configs.py) + per-kernel times
# of the configs given under rocprofv3 --kernel-trace --stats.   tools/kprof_cfg.sh <tag> <golden name>...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
python3 tools/bench_configs.py 2>&1 | grep "^{" > "$O/walls.jsonl"
for cfg in "$@"; do
  tools/kprof.sh $cfg ${TAG}/k_$cfg > /dev/null 2>&1
  echo "-- $cfg" >> "$O/kernels.txt"; cat "$O/k_$cfg/kstats.txt" >> "$O/kernels.txt"
done
python3 - <<PY
import json
for l in open("$O/walls.jsonl"):
    d = json.loads(l); print(d["case"], "%.3f ms" % (1e3 * d["gpu_seconds"]), "frac", d["frac_of_8TBps"], d["path"], d["ms"])
PY

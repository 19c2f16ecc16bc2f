#!/bin/bash
# SURVEY 8(d) scaling series on one GPU: the headline bench at N = 4k .. 100k (same generator,
# g=12 m=8, 495 combos). One JSON line per N into gpurun_out/n_series.jsonl.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"; : > "$R/gpurun_out/n_series.jsonl"
for n in 4000 8000 16000 32000 64000 100000; do
  python3 "$R/bench.py" --n-seq $n --steps 3 --warmup 1 --no-cpu-baseline --no-also 2>/dev/null | grep '^{' >> "$R/gpurun_out/n_series.jsonl"
done

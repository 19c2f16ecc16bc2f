#!/bin/bash
# Round 6: the two-level blocks form of the sparse update stage at large N (digests against the 64-bit-atomics run).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p "$O"
cd "$R"
timeout 900 python3 tools/bench_sparse_large_n.py --combos 20 "$@" > "$O/sparse_large_n_blocks.jsonl" 2> "$O/sparse_large_n_blocks.err"
cat "$O/sparse_large_n_blocks.jsonl"; tail -5 "$O/sparse_large_n_blocks.err"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_large_n_blocks" -- python3 "$R/tools/bench_sparse_large_n.py" --only protein_like_64k --combos 20 > "$O/stats_large_n_blocks.out" 2> "$O/stats_large_n_blocks.err")
find "$O/stats_large_n_blocks" -name "*kernel_stats.csv" | head -1 | xargs -r head -22 | cut -c1-60,300-420

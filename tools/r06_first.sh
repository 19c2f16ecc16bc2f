#!/bin/bash
# Round 6, first GPU call: counter list, the sparse kernels' instruction prices, the sparse dataflow at large N, and the
# new counter passes over config 4.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p "$O"
cd "$R"
rocprofv3 -L > "$O/counters.txt" 2>&1
tools/ubench_sparse_ops > "$O/ubench_sparse_ops.txt" 2>&1
timeout 900 python3 tools/bench_sparse_large_n.py --combos 20 > "$O/sparse_large_n.jsonl" 2> "$O/sparse_large_n.err"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_large_n" -- python3 "$R/tools/bench_sparse_large_n.py" --only protein_like_64k --combos 10 > "$O/stats_large_n.out" 2> "$O/stats_large_n.err")
tools/pmc_passes.sh r06_cfg4 "$R/tools/profile_one.py" f7_cfg4_prot219_exact > "$O/pmc_cfg4.log" 2>&1
tail -5 "$O/sparse_large_n.jsonl"; tail -3 "$O/sparse_large_n.err"
find "$O/stats_large_n" -name "*kernel_stats.csv" | head -1 | xargs -r head -25

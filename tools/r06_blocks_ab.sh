#!/bin/bash
# A/B of the blocks form at large N: tools/r06_blocks_ab.sh <workload> <combos> "tuning" "tuning" ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
W=$1; C=$2; shift 2
python3 tools/bench_sparse_large_n.py --only $W --combos $C | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('(default)', d['ms_per_combo'], 'ms/combo', d['frac_of_hbm_peak'], d['sparse_form'], 'passes', d['passes'], d['digest'], d['ms'])"
for t in "$@"; do
  python3 tools/bench_sparse_large_n.py --only $W --combos $C --tuning "$t" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_combo'], 'ms/combo', d['frac_of_hbm_peak'], d['sparse_form'], 'passes', d['passes'], d['digest'], d['ms'])"
done

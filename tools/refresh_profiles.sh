#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: the headline bench un-profiled, under
# rocprofv3 --kernel-trace --stats, and one --pmc pass per HBM counter. Raw output lands in
# gpurun_out/prof/; tools/collect_profiles.py turns it into the files under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
cd "$R" && python3 bench.py --steps 2 --warmup 1 > "$O/bench.json" 2> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-also > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-also > /dev/null 2> "$O/pmc_$c.err"
done
cd "$R" && python3 tools/bench_configs.py > "$O/configs.jsonl" 2> "$O/configs.err"
python3 tools/bench_large_g.py > "$O/large_g.jsonl" 2> "$O/large_g.err"
ls -R "$O" | head -50

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python3 tools/bench_large_g.py > $O/large_g.jsonl 2> $O/large_g.err; cat $O/large_g.jsonl; tail -3 $O/large_g.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_large_g -- python3 $R/tools/bench_large_g.py 20,14 > $O/stats_large_g.out 2>&1
python3 $R/tools/kstats.py $O/stats_large_g | head -14

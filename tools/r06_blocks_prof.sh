#!/bin/bash
# kernel times of one large-N workload of the blocks form:  tools/r06_blocks_prof.sh WORKLOAD "tuning"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/stats_blk
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_blk -- python3 $R/tools/bench_sparse_large_n.py --only $1 --combos 20 --tuning $2 > $O/stats_blk.out 2>&1
tail -1 $O/stats_blk.out | cut -c1-330
python3 $R/tools/kstats.py $O/stats_blk | head -14

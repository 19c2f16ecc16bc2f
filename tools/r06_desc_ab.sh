#!/bin/bash
# A/B of the descriptor form (tuning sparse_desc) on the GPU box:  tools/r06_desc_ab.sh CASES "setting" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
CASES=$1; shift
python3 tools/ab_env.py $CASES "$@" 2>&1 | tee $O/desc_ab.txt

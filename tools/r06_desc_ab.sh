#!/bin/bash
# A/B of the descriptor form (tuning sparse_desc) on the GPU box: the large-g regime, config 4, config 1
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
python3 tools/ab_env.py large_g,f7_cfg4_prot219_exact,f7_cfg1_prot11_approx_t1 "sparse_desc=-1" "sparse_desc=1" "sparse_desc=1,sparse_desc_min=32" "sparse_desc=1,sparse_desc_min=16" "sparse_desc=1,sparse_desc_min=8" 2>&1 | tee $O/desc_ab.txt

#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  tools/pmc_passes.sh <tag> <python script + args>
# One rocprofv3 pass per counter set (SQ: 8 slots; TCC: FETCH_SIZE and WRITE_SIZE never together) plus a
# --kernel-trace --stats pass, all on the same command. Raw CSVs land in gpurun_out/pmc/<tag>/<set>/;
# tools/collect_pmc.py turns them into one per-kernel JSON summary for profiles/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$@" > "$O/stats.out" 2> "$O/stats.err"
i=0
# round 6: what bounds the VALU-heavy kernels — issue occupancy (ACTIVE_INST_VALU / SCA in quad-cycles of a wave, INST_CYCLES_VALU,
# BUSY_CU_CYCLES) and the LDS queue depth. A counter name this ROCm does not know fails its own pass only, hence one risky name a pass.
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INST_CYCLES_VALU SQ_INSTS_VALU" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/set$i" -- python3 "$@" > "$O/set$i.out" 2> "$O/set$i.err" || echo "set $i ($set) failed: $(tail -2 $O/set$i.err)"
done
cd "$R" && python3 tools/collect_pmc.py "$TAG"

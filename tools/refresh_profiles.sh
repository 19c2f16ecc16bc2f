#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: the headline bench un-profiled, under
# rocprofv3 --kernel-trace --stats, one --pmc pass per HBM counter on the same command, the sparse
# pipeline's counter passes (configs 4 and 1), configs 1-4 timings, bench.py --config 4, the N series
# and the micro-benchmarks. Raw output lands in gpurun_out/prof/ and gpurun_out/pmc/;
# tools/collect_profiles.py turns it into the files under profiles/.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
cd "$R" && python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic > "$O/bench.json" 2> "$O/bench.err"
FSK_BENCH_FORCE_DIST=1 MASTER_PORT=29777 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-live-traffic > "$O/bench_rccl_world1.json" 2> "$O/bench_rccl_world1.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-live-traffic > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-live-traffic > /dev/null 2> "$O/pmc_$c.err"
done
# the sparse pipeline's counter passes at config 4 (traffic_config4.json is built from them)
(cd "$R" && tools/pmc_passes.sh ${TAG}_cfg4 "$R/tools/profile_one.py" f7_cfg4_prot219_exact > "$O/pmc_cfg4.log" 2>&1)
# traffic.json / traffic_config4.json from the passes above (needs bench.json, stats, pmc_*), then the headline
# line once more and the config-4 line: they now carry the measured HBM traffic of exactly these sources
(cd "$R" && python3 tools/collect_profiles.py $TAG > "$O/collect_on_box.log" 2>&1 && python3 bench.py --steps 2 --warmup 1 > "$O/bench_final.json" 2> "$O/bench_final.err";
 python3 bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-also > "$O/bench_config4.json" 2> "$O/bench_config4.err")
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_configs" -- python3 "$R/tools/bench_configs.py" > /dev/null 2> "$O/stats_configs.err"
cd "$R" && python3 tools/bench_configs.py > "$O/configs.jsonl" 2> "$O/configs.err"
python3 tools/bench_large_g.py > "$O/large_g.jsonl" 2> "$O/large_g.err"
# round 6: the descriptors — large-g kernel times and counters, the A/B against the word streams (large-g, configs 1 and 4), large N without them
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_large_g" -- python3 "$R/tools/bench_large_g.py" 20,14 > "$O/stats_large_g.out" 2>&1)
python3 tools/kstats.py "$O/stats_large_g" > "$O/kernel_times_large_g.txt" 2>&1
tools/pmc_passes.sh ${TAG}_large_g16 "$R/tools/bench_large_g.py" 16,10 > "$O/pmc_large_g16.log" 2>&1
python3 tools/ab_env.py large_g,f7_cfg1_prot11_approx_t1,f7_cfg4_prot219_exact "sparse_desc=-1" "sparse_desc=1" "sparse_desc_min=48" "sparse_desc_cols=3" > "$O/desc_ab.txt" 2>&1
python3 tools/bench_sparse_large_n.py --combos 20 --tuning sparse_desc=-1 > "$O/sparse_large_n_no_desc.jsonl" 2>> "$O/sparse_large_n.err"
for f in 1 2; do python3 tools/bench_sparse_large_n.py --only dna_k8_6k,dna_k8_8k,dna_k8_12k,dna_k8_16k --combos 20 --tuning sparse_form=$f,sparse_desc=1; done > "$O/sparse_mid_n_dna_k8.jsonl" 2>> "$O/sparse_large_n.err"
tools/pmc_passes.sh ${TAG}_cfg1 "$R/tools/profile_one.py" f7_cfg1_prot11_approx_t1 > "$O/pmc_cfg1.log" 2>&1
tools/bench_n_series.sh > /dev/null 2>&1; cp "$R/gpurun_out/n_series.jsonl" "$O/n_series.jsonl"
[ -x tools/ubench_mfma_i8 ] && tools/ubench_mfma_i8 > "$O/ubench_mfma_i8.txt" 2>&1
# the in-process multi-GPU engine (FastSK(devices=[...])) on this box: RCCL from the host C++ over a world of one at full
# size, and three engines sharing the GPU over the peer-to-peer kernels at N = 32000 (a smoke test of the group, not a measurement)
python3 bench.py --gpus 1 --inproc --steps 3 --warmup 1 --no-cpu-baseline --no-also > "$O/bench_inproc1.json" 2> "$O/bench_inproc1.err"
FSK_BENCH_SHARE_GPU=1 python3 bench.py --gpus 3 --inproc --n-seq 32000 --steps 3 --warmup 1 --no-cpu-baseline --no-also > "$O/bench_inproc3_shared.json" 2> "$O/bench_inproc3_shared.err"
python3 bench.py --gpus 1 --inproc --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-also > "$O/bench_inproc1_config4.json" 2> "$O/bench_inproc1_config4.err"
python3 tools/make_digests.py > "$O/digests.log" 2>&1; cp profiles/k_digests.json "$O/k_digests.json"
tools/trace_cfg1.sh > "$O/config1_timeline.txt" 2>&1
tools/kprof.sh f7_cfg4_prot219_exact prof/kprof_cfg4 > /dev/null 2>&1
tools/kprof.sh f7_cfg1_prot11_approx_t1 prof/kprof_cfg1 > /dev/null 2>&1
python3 tools/time_ep300_approx.py > "$O/ep300_approx.txt" 2>&1
# round 6: the sparse dataflow beyond the owner bands (the two-level blocks), its counters at N = 64k, the drop-in wall clock
python3 tools/bench_sparse_large_n.py --combos 20 > "$O/sparse_large_n.jsonl" 2> "$O/sparse_large_n.err"
python3 tools/bench_sparse_large_n.py --only protein_like_8k,protein_like_12k,protein_like_16k --combos 40 > "$O/sparse_mid_n.jsonl" 2>> "$O/sparse_large_n.err"
tools/pmc_passes.sh ${TAG}_blocks64k "$R/tools/bench_sparse_large_n.py" --only protein_like_64k --combos 20 > "$O/pmc_blocks64k.log" 2>&1
python3 tools/time_dropin.py > "$O/dropin_wall.json" 2> "$O/dropin_wall.err"
[ -x tools/ubench_sparse_ops ] && tools/ubench_sparse_ops > "$O/ubench_sparse_ops.txt" 2>&1
[ -x tools/ubench_rmw ] && tools/ubench_rmw > "$O/ubench_rmw.txt" 2>&1
ls -R "$O" | head -60

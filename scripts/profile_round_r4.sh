#!/bin/bash
# Round 4: one GPU session that regenerates the measurements DESIGN.md / profiles/r04_* cite.
set -o pipefail
TAG=r04
O=gpurun_out/$TAG; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
step() { echo "== $1 ($(date +%T))"; }
step bench_cfg4; timeout -k 10 600 python bench.py > $O/bench_cfg4_n1.json 2> $O/bench_cfg4.err; echo "bench cfg4 exit=$?"
step bench_cfg3; timeout -k 10 300 python bench.py --workload cfg3 --steps 400 --warmup 40 --no-extra-configs > $O/bench_cfg3_n1.json 2> $O/bench_cfg3.err; echo "bench cfg3 exit=$?"
step bench_cfg5; timeout -k 10 600 python bench.py --workload cfg5 --steps 40 --warmup 5 --cpu-iters 2 --no-extra-configs > $O/bench_cfg5_n1.json 2> $O/bench_cfg5.err; echo "bench cfg5 exit=$?"
step bench_forced_dist; GS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu --no-extra-configs --steps 200 --warmup 20 > $O/bench_cfg4_library_rccl_group_of_one_forced_shared_top.json 2> $O/bench_fd.err; echo "forced-dist exit=$?"
step gloo2; GS_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2rank_gloo_rehearsal_one_gpu.json 2> $O/bench2.err; echo "gloo2 exit=$?"
step levels; GS_LIB=$B/var_ts/libgraphslam_hip.so timeout -k 10 200 python scripts/level_times.py cfg4 > $O/level_completion_times_cfg4.txt 2>&1
GS_LIB=$B/var_ts/libgraphslam_hip.so timeout -k 10 300 python scripts/level_times.py cfg5 > $O/level_completion_times_cfg5.txt 2>&1
GS_SUBTREE=1 GS_LIB=$B/var_ts/libgraphslam_hip.so timeout -k 10 200 python scripts/level_times.py cfg4 > $O/level_completion_times_cfg4_bottom_subtrees_in_one_workgroup.txt 2>&1
GS_SUBTREE=1 GS_LIB=$B/var_ts/libgraphslam_hip.so timeout -k 10 300 python scripts/level_times.py cfg5 > $O/level_completion_times_cfg5_bottom_subtrees_in_one_workgroup.txt 2>&1
step subtree_ab; for cfg in cfg4 cfg5; do timeout -k 10 400 python scripts/ab_iter.py $cfg "" "GS_SUBTREE=1" "GS_LEAF_POSES=8" "GS_LEAF_POSES=8 GS_SUBTREE=1" 2>&1 | tail -8; done > $O/bottom_subtrees_and_leaf_size_ab.txt; cat $O/bottom_subtrees_and_leaf_size_ab.txt
step call_latency; timeout -k 10 300 python scripts/call_latency.py > $O/call_latency_reference_sizes.txt 2>&1; cat $O/call_latency_reference_sizes.txt
step assoc; for c in cfg3 cfg4 cfg5; do timeout -k 10 200 python scripts/assoc_time.py $c; done > $O/association_resident.txt 2>&1; cat $O/association_resident.txt
step shard_footprint; timeout -k 10 600 python scripts/shard_footprint.py 8 cfg4 > $O/shard_footprint_8xcfg4.txt 2>&1; cat $O/shard_footprint_8xcfg4.txt
step pmc_iter; bash scripts/pmc_iter.sh $TAG cfg4 5 > /dev/null; cp gpurun_out/pmc_$TAG.txt $O/iteration_hbm_traffic_cfg4.txt
bash scripts/pmc_iter.sh ${TAG}c5 cfg5 3 > /dev/null; cp gpurun_out/pmc_${TAG}c5.txt $O/iteration_hbm_traffic_cfg5.txt
step lin_ablation; bash scripts/r4_b.sh cfg4 > /dev/null 2>&1; cp gpurun_out/r4b/summary.txt $O/linearize_traffic_by_part_cfg4.txt; cat $O/linearize_traffic_by_part_cfg4.txt
step pmc_groups; bash scripts/pmc_groups.sh ${TAG}mix cfg4 3 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" > $O/solver_pmc_instruction_mix_cfg4.txt 2>&1
step rocprof; cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $GRAFT_REPO_ROOT/$O/rocprof_bench.log 2>&1; echo "rocprof exit=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_cfg4_kernel_stats.csv && head -16 "$f"
f=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/lin_duration_check.py "$f" | tee $O/linearize_duration_trace.txt
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
step done
for c in cfg3 cfg4 cfg5; do python - <<PY
import json
try:
    d = json.load(open("$O/bench_${c}_n1.json"))
    print("$c", round(d["value"]), "it/s", "lin frac", round(d["roofline"]["frac"], 3), "b2b", round(d["roofline"].get("achieved_back_to_back", 0) / 8000, 3), d.get("phases_ms"), "cpu", d.get("cpu_baseline", {}) and round(d["cpu_baseline"]["value"], 2))
except Exception as e:
    print("$c", "no json:", e)
PY
done

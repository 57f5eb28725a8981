#!/bin/bash
# One GPU session that regenerates every measurement the round's DESIGN.md / profiles/ cite.  usage: profile_round.sh TAG
TAG=${1:-r02}
set -o pipefail
mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
python bench.py > $O/bench_cfg4_n1.json 2> $O/bench_cfg4.err; echo "bench cfg4 exit=$?"
python bench.py --workload cfg3 --steps 400 --warmup 40 > $O/bench_cfg3_n1.json 2> $O/bench_cfg3.err; echo "bench cfg3 exit=$?"
python bench.py --workload cfg5 --steps 40 --warmup 5 --cpu-iters 2 > $O/bench_cfg5_n1.json 2> $O/bench_cfg5.err; echo "bench cfg5 exit=$?"
GS_PLAN_TIMING=1 python scripts/iter_loop.py cfg4 1 2>&1 | grep -E "upload|plan phase" > $O/structure_phase_breakdown_cfg4.txt
GS_LIB=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build/var_ts/libgraphslam_hip.so python scripts/level_times.py cfg4 > $O/level_completion_times_cfg4.txt 2>&1
bash scripts/pmc_iter.sh $TAG cfg4 5 > /dev/null; cp gpurun_out/pmc_$TAG.txt $O/iteration_hbm_traffic_cfg4.txt
bash scripts/pmc_iter.sh ${TAG}c5 cfg5 3 > /dev/null; cp gpurun_out/pmc_${TAG}c5.txt $O/iteration_hbm_traffic_cfg5.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $GRAFT_REPO_ROOT/$O/rocprof_bench.log 2>&1; echo "rocprof exit=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_cfg4_kernel_stats.csv && head -12 "$f"
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
for c in cfg3 cfg4 cfg5; do python - <<PY
import json
try:
    d = json.load(open("$O/bench_${c}_n1.json"))
    print("$c", round(d["value"]), "it/s", "lin frac", round(d["roofline"]["frac"], 3), "b2b", round(d["roofline"].get("achieved_back_to_back", 0) / 8000, 3), d.get("phases_ms"))
except Exception as e:
    print("$c", "no json:", e)
PY
done
cat $O/structure_phase_breakdown_cfg4.txt

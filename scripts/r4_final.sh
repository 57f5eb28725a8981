#!/bin/bash
# round 4, final measurement pass (kernels unchanged since scripts/profile_round_r4.sh ran; the structure phase changed): bench lines, call latency,
# keyframe growth, kernel stats of the default command
O=gpurun_out/r04f; mkdir -p $O
step() { echo "== $1 ($(date +%T))"; }
step smoke; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
step bench_cfg4; timeout -k 10 600 python bench.py > $O/bench_cfg4_n1.json 2> $O/bench_cfg4.err; echo "bench cfg4 exit=$?"
step bench_cfg3; timeout -k 10 300 python bench.py --workload cfg3 --steps 400 --warmup 40 --no-extra-configs > $O/bench_cfg3_n1.json 2> $O/bench_cfg3.err; echo "bench cfg3 exit=$?"
step bench_cfg5; timeout -k 10 600 python bench.py --workload cfg5 --steps 40 --warmup 5 --cpu-iters 2 --no-extra-configs > $O/bench_cfg5_n1.json 2> $O/bench_cfg5.err; echo "bench cfg5 exit=$?"
step bench_forced_dist; GS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu --no-extra-configs --steps 200 --warmup 20 > $O/bench_cfg4_library_rccl_group_of_one_forced_shared_top.json 2> $O/bench_fd.err; echo "forced-dist exit=$?"
step gloo2; GS_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > $O/bench_2rank_gloo_rehearsal_one_gpu.json 2> $O/bench2.err; echo "gloo2 exit=$?"
step call_latency; timeout -k 10 300 python scripts/call_latency.py > $O/call_latency_reference_sizes.txt 2>&1; cat $O/call_latency_reference_sizes.txt
step structure; GS_PLAN_TIMING=1 timeout -k 10 300 python scripts/structure_probe.py 100000 10000 > $O/structure_probe_cfg4.txt 2>&1; tail -30 $O/structure_probe_cfg4.txt
step rocprof; cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu --no-extra-configs > $GRAFT_REPO_ROOT/$O/rocprof_bench.log 2>&1; echo "rocprof exit=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r04f -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_cfg4_kernel_stats.csv && head -10 "$f" | cut -c1-160
f=$(find gpurun_out/prof_r04f -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/lin_duration_check.py "$f" | tee $O/linearize_duration_trace.txt
find gpurun_out/prof_r04f -name "*kernel_trace.csv" -size +20M -delete
for c in cfg3 cfg4 cfg5; do python - <<PY
import json
try:
    d = json.load(open("$O/bench_${c}_n1.json"))
    print("$c", round(d["value"]), "it/s", "lin frac", round(d["roofline"]["frac"], 3), "e2e", round(d["roofline"].get("frac_event_to_event", 0), 3), "b2b", round(d["roofline"].get("achieved_back_to_back", 0) / 8000, 3), d.get("phases_ms"), "cpu", d.get("cpu_baseline", {}) and round(d["cpu_baseline"]["value"], 2), "e2e ms", d.get("optimize10_end_to_end_ms", {}).get("gpu"))
except Exception as e:
    print("$c", "no json:", e)
PY
done

#!/bin/bash
# round 4: rocprofv3 kernel stats of the default bench command's timed workload (cfg4 only) + the linearisation kernel's durations by context
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r04b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu --no-extra-configs > $O/rocprof_bench.log 2>&1; echo "rocprof exit=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r04b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_cfg4_kernel_stats.csv && head -12 "$f"
f=$(find gpurun_out/prof_r04b -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/lin_duration_check.py "$f" | tee $O/linearize_duration_trace.txt
find gpurun_out/prof_r04b -name "*kernel_trace.csv" -size +20M -delete
tail -2 $O/rocprof_bench.log | cut -c1-600

#!/bin/bash
# rocprofv3 kernel trace of a linearise-heavy run; prints per-kernel stats
TAG=${1:-lin}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/scripts/lin_loop.py > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +5M -delete
tail -3 gpurun_out/prof_$TAG.log

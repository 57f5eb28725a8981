#!/bin/bash
# One GPU session: parity tests, smoke, bench (N=1), 2-rank rehearsal on the one GPU, rocprofv3 kernel trace.
set -o pipefail
TAG=${1:-r1}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest exit=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -3 gpurun_out/pytest_gpu_$TAG.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke exit=$?" >> gpurun_out/smoke_$TAG.log
tail -2 gpurun_out/smoke_$TAG.log
timeout -k 10 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench exit=$?"
cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
GS_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/bench2_$TAG.json 2> gpurun_out/bench2_$TAG.err; echo "bench2(gloo rehearsal) exit=$?"
cat gpurun_out/bench2_$TAG.json; tail -3 gpurun_out/bench2_$TAG.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; echo "rocprof exit=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete

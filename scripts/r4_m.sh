#!/bin/bash
# round 4, batch M: A9 inside the backward solve (fused_update) — GPU suite, then on / off at cfg4 / cfg3 / cfg5 and at the reference's graph sizes
O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt
[ $rc -ne 0 ] && exit $rc
for cfg in cfg4 cfg3 cfg5; do timeout -k 10 300 python scripts/ab_iter.py $cfg "GS_FUSED_UPDATE=0" "GS_FUSED_UPDATE=1" 2>&1 | tail -4; done | tee $O/ab.txt
for v in 0 1; do echo "== GS_FUSED_UPDATE=$v"; GS_FUSED_UPDATE=$v timeout -k 10 300 python scripts/call_latency.py 2>&1 | tail -5; done | tee $O/call_latency.txt

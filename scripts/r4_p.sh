#!/bin/bash
# round 4, batch P: ticketed workgroup numbers in the whole-tree launches — GPU suite, then on / off at cfg4 / cfg3 / cfg5 / K = 16 and at lap sizes
O=gpurun_out/r4p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt
[ $rc -ne 0 ] && exit $rc
for cfg in cfg4 cfg3 cfg5; do timeout -k 10 300 python scripts/ab_iter.py $cfg "GS_TICKETS=0" "GS_TICKETS=1" 2>&1 | tail -4; done | tee $O/ab.txt
for v in 0 1; do echo "== GS_TICKETS=$v"; GS_TICKETS=$v timeout -k 10 300 python scripts/call_latency.py 2>&1 | tail -4; done | tee $O/call_latency.txt
for v in 0 1; do echo "== GS_TICKETS=$v wide view"; GS_TICKETS=$v timeout -k 10 300 python scripts/wide_view.py 2>&1 | tail -3; done | tee $O/wide.txt

#!/bin/bash
# round 3: full GPU suite on the class-kernel build, then the per-call latency at the reference's own graph sizes
O=gpurun_out/r3p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit=$rc" | tee -a $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python scripts/call_latency.py 2>&1 | tee $O/call_latency.txt

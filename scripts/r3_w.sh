#!/bin/bash
# round 3: the tail kernel without atomics (one workgroup, fixed-order sums) — growth tests, timing with 1 / 4 / 5 tail poses
O=gpurun_out/r3w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "growth or appended or fallback or falls_back" > $O/pytest_growth.log 2>&1; rc=$?; echo "pytest growth exit=$rc"; tail -5 $O/pytest_growth.log
[ $rc -eq 0 ] || exit $rc
for a in "cfg4 1 1" "cfg4 4 4" "cfg4 5 5 60007" "cfg3 2 2"; do timeout -k 10 300 python scripts/growth_time.py $a; done 2>&1 | tee $O/growth_time.txt

#!/bin/bash
# round 4, batch X: a shard's helper thread sends only the odometry records of its own window — GPU suite, random sharded graphs, phase table of a rank
O=gpurun_out/r4x; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_suite.txt 2>&1; rc=$?; tail -3 $O/gpu_suite.txt
[ $rc -ne 0 ] && exit $rc
GS_PLAN_TIMING=1 timeout -k 10 400 python scripts/stress_gpu.py 500 60 > $O/stress.txt 2> $O/stress_err.txt; tail -2 $O/stress.txt; grep -c "sent after the plan" $O/stress_err.txt
GS_PLAN_TIMING=1 timeout -k 10 600 python scripts/shard_footprint.py 8 cfg4 > $O/shard_footprint.txt 2> $O/phases_raw.txt; cat $O/shard_footprint.txt
python scripts/plan_phase_table.py $O/phases_raw.txt | tail -4 | tee $O/phases.txt
grep "wait for the raw\|sent after" $O/phases_raw.txt | tail -8

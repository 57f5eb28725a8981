#!/bin/bash
# round 4, batch W: phase table of a rank's structure phase against a single handle's (8 x cfg4), last code
O=gpurun_out/r4w; mkdir -p $O
GS_PLAN_TIMING=1 timeout -k 10 600 python scripts/shard_footprint.py 8 cfg4 > $O/shard_footprint.txt 2> $O/phases_raw.txt; cat $O/shard_footprint.txt
python scripts/plan_phase_table.py $O/phases_raw.txt | tee $O/phases.txt

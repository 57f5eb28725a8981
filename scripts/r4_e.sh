#!/bin/bash
# round 4, experiment E: leaf size x cluster fan-out at cfg3 / cfg4 / cfg2-size graphs
O=gpurun_out/r4e; mkdir -p $O
for cfg in cfg3 cfg4; do
  timeout -k 10 500 python scripts/ab_iter.py $cfg "" "GS_LEAF_POSES=6" "GS_LEAF_POSES=5" "GS_LEAF_POSES=6 GS_CLUSTER_WAYS=6" "GS_LEAF_POSES=6 GS_GROW_HEADROOM=0" "GS_LEAF_POSES=7 GS_GROW_HEADROOM=0" "GS_GROW_HEADROOM=0" 2>&1 | tail -14
done | tee $O/leaf_ab.txt

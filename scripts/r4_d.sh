#!/bin/bash
# round 4, experiment D: leaf size (GS_LEAF_POSES 8 = default, 7, 6: leaves of <= 47 scalars take the three-tile-row leaf instance)
O=gpurun_out/r4d; mkdir -p $O
for cfg in cfg4 cfg5; do
  timeout -k 10 400 python scripts/ab_iter.py $cfg "" "GS_LEAF_POSES=7" "GS_LEAF_POSES=6" "GS_LEAF_POSES=7 GS_SUBTREE=1" 2>&1 | tail -8
done | tee $O/leaf_ab.txt

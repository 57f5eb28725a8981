#!/bin/bash
# round 3: leaf size x cluster fan-out on the current kernels (cfg4, 30 timed iterations each)
O=gpurun_out/r3s; mkdir -p $O
run() { echo "== $*" | tee -a $O/sweep2.txt; env "$@" timeout -k 10 120 python scripts/iter_time.py ${CFG:-cfg4} 2>&1 | tail -1 | tee -a $O/sweep2.txt; }
for L in 4 5 6; do for W in 8 10 12 16; do run GS_LEAF_POSES=$L GS_CLUSTER_WAYS=$W; done; done
run GS_LEAF_POSES=8 GS_CLUSTER_WAYS=8
CFG=cfg5 run GS_LEAF_POSES=8 GS_CLUSTER_WAYS=8
CFG=cfg5 run GS_LEAF_POSES=6 GS_CLUSTER_WAYS=8
CFG=cfg3 run GS_LEAF_POSES=8 GS_CLUSTER_WAYS=8
CFG=cfg3 run GS_LEAF_POSES=6 GS_CLUSTER_WAYS=8

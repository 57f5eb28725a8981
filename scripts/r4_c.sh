#!/bin/bash
# round 4, experiment C: the bottom subtrees (k_factor3_sub) on / off, cfg3 (forced leaf launches) cfg4 cfg5, + per-level completion times
O=gpurun_out/r4c; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
for cfg in cfg4 cfg5; do for rep in 1 2; do for v in "GS_SUBTREE=1" "GS_SUBTREE=0"; do
  echo -n "$v $cfg: "; env $v timeout -k 10 200 python scripts/iter_time.py $cfg 2>&1 | tail -1
done; done; done 2>&1 | tee $O/sub_ab.txt
for v in "GS_SUBTREE=1" "GS_SUBTREE=0"; do
  echo "== $v cfg4"; env $v GS_LIB=$B/var_ts/libgraphslam_hip.so timeout -k 10 200 python scripts/level_times.py cfg4 2>&1 | grep -A8 "^factor"
  echo "== $v cfg5"; env $v GS_LIB=$B/var_ts/libgraphslam_hip.so timeout -k 10 300 python scripts/level_times.py cfg5 2>&1 | grep -A8 "^factor"
done > $O/levels.txt 2>&1
cat $O/levels.txt

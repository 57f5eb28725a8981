#!/bin/bash
# round 4, experiment I: level 1 behind the leaves in the leaf launch (k_factor3_leaf1) on / off; launch-mode parity tests first
O=gpurun_out/r4i; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_modes or cfg4_properties or growth or grown or keyframe" > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for cfg in cfg4 cfg5; do
  timeout -k 10 300 python scripts/ab_iter.py $cfg "GS_LEAF_LEVEL1=0" "GS_LEAF_LEVEL1=1" 2>&1 | tail -4
done | tee $O/ab.txt
for v in 1 0; do
  echo "== GS_LEAF_LEVEL1=$v cfg4"; GS_LEAF_LEVEL1=$v GS_LIB=$B/var_ts/libgraphslam_hip.so timeout -k 10 200 python scripts/level_times.py cfg4 2>&1 | grep -A8 "^factor"
done | tee $O/levels.txt

#!/bin/bash
# round 4, batch K: fallback-retry tests; every level above the leaves on four waves per front in a block-only flagged launch (5 workgroups / CU)
O=gpurun_out/r4k; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fall or timeout or launch_modes or healthy" > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/ab_iter.py cfg4 "" "GS_BLOCK_FRONTS=1024" "GS_BLOCK_FRONTS=4096" 2>&1 | tail -6 | tee $O/ab_cfg4.txt
timeout -k 10 300 python scripts/ab_iter.py cfg5 "" "GS_BLOCK_FRONTS=20000" 2>&1 | tail -4 | tee $O/ab_cfg5.txt
timeout -k 10 300 python scripts/ab_iter.py cfg3 "" "GS_LEAF_KERNEL=2 GS_BLOCK_FRONTS=4096" 2>&1 | tail -4 | tee $O/ab_cfg3.txt
echo "== GS_BLOCK_FRONTS=4096 cfg4"; GS_BLOCK_FRONTS=4096 GS_LIB=$B/var_ts/libgraphslam_hip.so timeout -k 10 200 python scripts/level_times.py cfg4 2>&1 | grep -A8 "^factor" | tee $O/levels.txt

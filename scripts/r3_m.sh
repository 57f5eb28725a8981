#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
TS=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build/var_ts/libgraphslam_hip.so
for kb in 0 40 52 78 100; do for c in cfg4 cfg5; do echo -n "GS_LEAF_LDS_KB=$kb $c: "; GS_LEAF_LDS_KB=$kb GS_LIB=$TS timeout -k 10 200 python scripts/level_times.py $c 2>&1 | grep "level  0" | head -1; done; done | tee $O/leaf_occupancy.txt

#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
GS_LIB=$B/var_div/libgraphslam_hip.so timeout -k 10 400 python scripts/parity_spread.py cfg4 exact_div 1 > $O/spread_cfg4_div.log 2>&1; grep -h "gpu vs" $O/spread_cfg4_div.log | cut -c1-200
GS_LIB=$B/var_div/libgraphslam_hip.so timeout -k 10 100 python scripts/iter_time.py cfg4 | tail -1

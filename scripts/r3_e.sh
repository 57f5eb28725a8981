#!/bin/bash
set -o pipefail
O=gpurun_out/r3e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "beyond_a_wave or wide_view or irregular or single_iteration_random or falls_back" > $O/pytest_big.log 2>&1; echo "pytest big exit=$?" | tee -a $O/pytest_big.log
tail -5 $O/pytest_big.log
timeout -k 10 300 python scripts/wide_view.py cfg4 8 16 24 2>&1 | tee $O/wide_view.txt
TS=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build/var_ts/libgraphslam_hip.so
GS_LIB=$TS timeout -k 10 200 python scripts/level_times.py cfg4 24 > $O/level_times_cfg4_K24.txt 2>&1; cat $O/level_times_cfg4_K24.txt
GS_LIB=$TS timeout -k 10 200 python scripts/level_times.py cfg4 16 > $O/level_times_cfg4_K16.txt 2>&1; cat $O/level_times_cfg4_K16.txt

#!/bin/bash
# round 4, experiment A: the leaf launch's stores written through the L2 (sc1) instead of staying dirty until the end-of-kernel write-back
O=gpurun_out/r4a; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
for cfg in cfg4 cfg5; do
for rep in 1 2; do
for v in default wtL wtU wtLU; do
  if [ $v = default ]; then L=""; else L=$B/var_$v/libgraphslam_hip.so; fi
  echo -n "$v $cfg: "; GS_LIB=$L timeout -k 10 200 python scripts/iter_time.py $cfg 2>&1 | tail -1
done; done; done 2>&1 | tee $O/wt_ab.txt
for v in ts tswtLU; do
  echo "== $v cfg4"; GS_LIB=$B/var_$v/libgraphslam_hip.so timeout -k 10 200 python scripts/level_times.py cfg4 2>&1 | tail -40
done > $O/levels_cfg4.txt 2>&1
for v in ts tswtLU; do
  echo "== $v cfg5"; GS_LIB=$B/var_$v/libgraphslam_hip.so timeout -k 10 300 python scripts/level_times.py cfg5 2>&1 | tail -40
done > $O/levels_cfg5.txt 2>&1

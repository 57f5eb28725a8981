#!/bin/bash
# round 4, experiment H: odometry operands of the linearisation pass requested with the streams of slots 2-3 (LIN_PP_EARLY) — A/B of tuning builds
# usage: scripts/r4_h.sh "var1 var2 ..." "cfg4 cfg5"   (variant "default" = the shipped library)
O=gpurun_out/r4h; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
for cfg in ${2:-cfg4}; do for rep in 1 2; do for v in ${1:-default}; do
  if [ $v = default ]; then L=""; else L=$B/var_$v/libgraphslam_hip.so; fi
  echo -n "$v $cfg: "; GS_LIB=$L timeout -k 10 200 python scripts/iter_time.py $cfg 2>&1 | tail -1
  echo -n "$v $cfg back to back: "; GS_LIB=$L timeout -k 10 200 python scripts/lin_loop.py $cfg 200 2>&1 | tail -1
done; done; done 2>&1 | tee $O/summary.txt

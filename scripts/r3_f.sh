#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
for rep in 1 2; do
for v in default nt0 nt3 nt5 w3 w5; do
  if [ $v = default ]; then L=""; else L=$B/var_$v/libgraphslam_hip.so; fi
  echo -n "$v cfg4: "; GS_LIB=$L timeout -k 10 120 python scripts/iter_time.py cfg4 2>&1 | tail -1
done; done 2>&1 | tee $O/lin_ab_cfg4.txt
for v in default nt0 nt5 w3 w5; do
  if [ $v = default ]; then L=""; else L=$B/var_$v/libgraphslam_hip.so; fi
  echo -n "$v cfg5: "; GS_LIB=$L timeout -k 10 200 python scripts/iter_time.py cfg5 2>&1 | tail -1
done 2>&1 | tee $O/lin_ab_cfg5.txt

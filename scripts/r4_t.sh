#!/bin/bash
# round 4, batch T: transparent-huge-page hints on the big plan arrays — structure phase of fresh handles (cfg4, cfg5) against the previous commit's library
O=gpurun_out/r4t; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
for rep in 1 2; do for v in nohuge default; do
  if [ $v = default ]; then L=""; else L=$B/var_$v/libgraphslam_hip.so; fi
  echo "== $v"; GS_LIB=$L timeout -k 10 300 python scripts/structure_probe.py 100000 10000 2>&1 | grep "^---" | grep -v "fresh handle [01]$\|same handle again$"
  GS_LIB=$L timeout -k 10 300 python scripts/structure_probe.py 1000000 50000 2>&1 | grep "^---" | grep -v "fresh handle [01]$\|same handle again$"
done; done | tee $O/structure.txt

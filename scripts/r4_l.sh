#!/bin/bash
# round 4, batch L: packed L panels in the light backward-solve launches (default) against the rectangular staging (tuning build bsrect)
O=gpurun_out/r4l; mkdir -p $O
B=$PWD/opendlv-logic-cfsd18-sensation-slam_amd/csrc/build
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_modes or cfg4_properties or growth or grown or ten_iterations or random or irregular or shard or rank" > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc -ne 0 ] && exit $rc
for cfg in cfg4 cfg5 cfg3; do for rep in 1 2; do for v in bsrect default; do
  if [ $v = default ]; then L=""; else L=$B/var_$v/libgraphslam_hip.so; fi
  echo -n "$v $cfg: "; GS_LIB=$L timeout -k 10 200 python scripts/iter_time.py $cfg 2>&1 | tail -1
done; done; done 2>&1 | tee $O/ab.txt

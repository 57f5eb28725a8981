#!/bin/bash
# round 3: the handle keeps its device memory across structure phases — GPU suite as it is and with poisoned pool memory, keyframe streams again
O=gpurun_out/r3y; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
GS_POOL_POISON=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_poison.log 2>&1; rc=$?; echo "pytest (poisoned pool) exit=$rc"; tail -12 $O/pytest_gpu_poison.log
timeout -k 10 400 python scripts/keyframe_stream.py cfg3 6000 24 2>&1 | cut -c1-330 | tee $O/keyframe_stream_cfg3.txt
timeout -k 10 400 python scripts/keyframe_stream.py cfg4 60000 24 2>&1 | cut -c1-330 | tee $O/keyframe_stream_cfg4.txt
timeout -k 10 200 python scripts/call_latency.py | tee $O/call_latency.txt

#!/bin/bash
# round 3: append-only growth — the GPU suite (growth parity tests included; the cluster-front headroom changes plans), timing at cfg4 / cfg5
O=gpurun_out/r3u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit=$rc"; tail -5 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/growth_time.py cfg4 1 1 2>&1 | tee $O/growth_time.txt
timeout -k 10 300 python scripts/growth_time.py cfg4 4 4 2>&1 | tee -a $O/growth_time.txt
timeout -k 10 300 python scripts/growth_time.py cfg3 2 2 2>&1 | tee -a $O/growth_time.txt
timeout -k 10 300 python scripts/growth_time.py cfg4 5 5 60007 2>&1 | tee -a $O/growth_time.txt
timeout -k 10 600 python scripts/growth_time.py cfg5 1 1 2>&1 | tee -a $O/growth_time.txt
timeout -k 10 300 python scripts/iter_time.py cfg4 | tee $O/iter_cfg4.txt

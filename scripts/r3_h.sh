#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit=$?" | tee -a $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
for c in cfg3 cfg4 cfg5; do timeout -k 10 200 python scripts/iter_time.py $c 2>&1 | tail -1; done | tee $O/iter_time.txt

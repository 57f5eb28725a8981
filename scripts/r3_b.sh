#!/bin/bash
# round 3, second GPU session: big fronts
set -o pipefail
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "beyond_a_wave or wide_view or irregular or single_iteration_random or falls_back or healthy" > $O/pytest_big.log 2>&1; echo "pytest big exit=$?" | tee -a $O/pytest_big.log
tail -15 $O/pytest_big.log

#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "beyond_a_wave or wide_view or irregular or single_iteration_random or falls_back" > $O/pytest_big.log 2>&1; echo "pytest big exit=$?" | tee -a $O/pytest_big.log
tail -3 $O/pytest_big.log
timeout -k 10 300 python scripts/wide_view.py cfg4 8 16 24 2>&1 | tee $O/wide_view.txt
for lev in 0 5; do timeout -k 10 200 python scripts/big_probe.py cfg4 24 $lev 3 2>&1 | tail -3; done | tee $O/big_probe_K24.txt

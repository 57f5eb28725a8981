#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
for lev in 0 1 5 8 11; do timeout -k 10 200 python scripts/big_probe.py cfg4 24 $lev 3 2>&1 | tail -2; done | tee $O/big_probe_K24.txt

#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
GS_HOST_TRIG=1 timeout -k 10 300 python scripts/system_diff.py cfg4 2>&1 | tee $O/system_diff_cfg4.txt

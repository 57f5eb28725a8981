#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 400 python scripts/parity_spread.py cfg4 g2o_jacobian 1 > $O/spread_cfg4.log 2>&1; grep -h "gpu vs\|max |gpu" $O/spread_cfg4.log | cut -c1-200
timeout -k 10 100 python scripts/iter_time.py cfg4 | tail -1

#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
GS_HOST_TRIG=1 timeout -k 10 400 python scripts/parity_spread.py cfg4 host_trig 1 > $O/spread_cfg4_host_trig.log 2>&1; grep -h " vs \|b_pose\|b_lm\|Hpl " $O/spread_cfg4_host_trig.log | cut -c1-200
timeout -k 10 400 python scripts/parity_spread.py cfg3 r03 > $O/spread_cfg3.log 2>&1; grep -h "gpu vs\|b_pose" $O/spread_cfg3.log | cut -c1-200
for lp in 3 4 5 6 8; do echo "GS_LEAF_POSES=$lp"; GS_LEAF_POSES=$lp timeout -k 10 200 python scripts/wide_view.py cfg4 24 2>&1 | tail -1; done | tee $O/k24_leaf_sweep.txt
for cw in 4 6 8; do echo "GS_CLUSTER_WAYS=$cw"; GS_CLUSTER_WAYS=$cw timeout -k 10 200 python scripts/wide_view.py cfg4 24 2>&1 | tail -1; done | tee -a $O/k24_leaf_sweep.txt
